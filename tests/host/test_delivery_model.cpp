// Host-only model test of bulk delivery's state machine (sdrainer_amd/csrc/host/delivery.h - the code the library runs,
// not a copy): a fake device thread completes fake events in stream order and stamps each set's block the way the pack
// kernels do, a producer enqueues batches over a ring of sets (parking before it reuses one, as capi_process.hip and
// capi_graph.hip do), consumers poll.  Checked: every batch is delivered exactly once, in order, with ITS stamp (a block
// overwritten before it was delivered or parked shows up as a wrong stamp), whatever the interleaving.  Built with
// -fsanitize=thread by tests/test_delivery_model.py.  No GPU, no HIP.
//
// Scenarios (the interleavings tests/test_gpu_parity.py covers on hardware, plus the one round 3's review found missing):
//   erratic    a consumer that polls in bursts and sleeps at random (test_many_batches_with_an_erratic_consumer)
//   nosync     short batches, nobody polls until the end: everything older than the ring is parked
//              (test_short_batches_without_sync_keep_the_ring_sets_safe)
//   blocking   a consumer thread parked in poll(wait) beside a producer that runs ahead (bench.py's consumer thread)
//   two        two consumers in poll(wait)
//   graph      graph replays (six batches each, four phases of sets), nothing polled for more than
//              GRAPH_PHASES * RING batches, then the release and a poll of everything, then eager batches again
#include <atomic>
#include <cassert>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../../sdrainer_amd/csrc/host/delivery.h"

namespace {

constexpr int RING = 6, PHASES = 4, SPAN = RING * PHASES;
constexpr size_t BLOCK = 64;

struct FakeEvent {
    std::atomic<int64_t> done{-1};  // generation (= batch) whose record has completed
    std::atomic<int64_t> want{-1};  // generation of the last record enqueued
};

thread_local std::string t_err;

struct Out {  // the caller's buffers
    int64_t batch = -1, stamp = -1;
    int slots = 0;
};

struct FakeBackend final : host::DeliveryBackend {
    std::atomic<long> copies{0};
    int wait(void *ev) override
    {
        FakeEvent *e = static_cast<FakeEvent *>(ev);
        const int64_t w = e->want.load(std::memory_order_acquire);  // hipEventSynchronize waits for the LAST record
        while (e->done.load(std::memory_order_acquire) < w)
            std::this_thread::yield();
        return SDR_OK;
    }
    int query(void *ev) override
    {
        FakeEvent *e = static_cast<FakeEvent *>(ev);
        return e->done.load(std::memory_order_acquire) >= e->want.load(std::memory_order_acquire) ? SDR_OK : SDR_ERR_WOULD_BLOCK;
    }
    std::unique_ptr<unsigned char[]> copy_used(const unsigned char *block, const host::BatchMeta &) override
    {
        copies++;
        std::unique_ptr<unsigned char[]> p(new unsigned char[BLOCK]);
        memcpy(p.get(), block, BLOCK);
        return p;
    }
    int deliver(const unsigned char *block, const host::BatchMeta &m, void *out) override
    {
        Out *o = static_cast<Out *>(out);
        o->batch = m.batch;
        o->slots = m.slots;
        memcpy(&o->stamp, block, sizeof(int64_t));
        return SDR_OK;
    }
    int report(int code, const char *msg) override
    {
        t_err = msg;
        return code;
    }
};

// the "device": executes enqueued batches in order, each after a random delay; a batch's execution writes its stamp
// into its set's block and then completes the set's two events
struct FakeDevice {
    struct Job {
        int64_t batch;
        unsigned char *block;
        FakeEvent *listen, *peaks;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    bool stop = false;
    std::thread th;
    int max_delay_us;
    explicit FakeDevice(int delay) : max_delay_us(delay) { th = std::thread([this] { run(); }); }
    ~FakeDevice()
    {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void enqueue(Job j)
    {
        j.listen->want.store(j.batch, std::memory_order_release);
        j.peaks->want.store(j.batch, std::memory_order_release);
        {
            std::lock_guard<std::mutex> g(mu);
            q.push_back(j);
        }
        cv.notify_one();
    }
    void run()
    {
        std::mt19937 rng(12345);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return stop || !q.empty(); });
                if (q.empty())
                    return;
                j = q.front();
                q.pop_front();
            }
            if (max_delay_us)
                std::this_thread::sleep_for(std::chrono::microseconds(rng() % (unsigned)max_delay_us));
            memcpy(j.block, &j.batch, sizeof(int64_t));  // the pack kernels' writes (the host reads them behind the events)
            j.peaks->done.store(j.batch, std::memory_order_release);
            j.listen->done.store(j.batch, std::memory_order_release);
        }
    }
};

struct Rig {
    FakeBackend be;
    host::Delivery d{&be, RING, SPAN};
    std::vector<std::unique_ptr<unsigned char[]>> blocks;
    std::vector<std::unique_ptr<FakeEvent>> events;
    FakeDevice dev;
    int64_t batch_index = 0;
    bool graph = false;
    int64_t graph_base = 0, replays = 0;
    explicit Rig(int delay) : dev(delay)
    {
        attach(RING);
        d.reset(true, 0);
    }
    void attach(int n)
    {
        d.grow(n);
        while ((int)blocks.size() < n) {
            blocks.emplace_back(new unsigned char[BLOCK]());
            events.emplace_back(new FakeEvent());
            events.emplace_back(new FakeEvent());
            host::ResultSet &rs = d.set((int)blocks.size() - 1);
            rs.block = blocks.back().get();
            rs.ev_listen = events[events.size() - 2].get();
            rs.ev_peaks = events[events.size() - 1].get();
        }
    }
    // one eager batch (capi_process.hip: park the set, enqueue, publish)
    void process()
    {
        const int si = (int)(batch_index % RING);
        const int rc = d.park(si);
        assert(rc == SDR_OK);
        (void)rc;
        host::ResultSet &rs = d.set(si);
        dev.enqueue({batch_index, rs.block, static_cast<FakeEvent *>(rs.ev_listen), static_cast<FakeEvent *>(rs.ev_peaks)});
        host::BatchMeta m;
        m.batch = batch_index;
        m.slots = 3;
        d.publish(si, std::move(m), true);
        batch_index++;
    }
    // capi_graph.hip
    void graph_capture()
    {
        assert(batch_index % RING == 0);
        graph_release();
        for (int i = 0; i < RING; i++) {
            const int rc = d.park((int)((std::max<int64_t>(batch_index - RING, 0) + i) % RING));
            assert(rc == SDR_OK);
            (void)rc;
        }
        attach(RING + SPAN);
        graph = true;
        graph_base = batch_index;
        replays = 0;
        d.graph_begin(batch_index);
    }
    void graph_launch()
    {
        const int set0 = RING + (int)(replays % PHASES) * RING;
        for (int k = 0; k < RING; k++) {
            const int rc = d.park(set0 + k);
            assert(rc == SDR_OK);
            (void)rc;
        }
        for (int k = 0; k < RING; k++) {
            host::ResultSet &rs = d.set(set0 + k);
            dev.enqueue({batch_index + k, rs.block, static_cast<FakeEvent *>(rs.ev_listen), static_cast<FakeEvent *>(rs.ev_peaks)});
        }
        for (int k = 0; k < RING; k++) {
            host::BatchMeta m;
            m.batch = batch_index;
            m.slots = 3;
            d.publish(set0 + k, std::move(m), true);
            batch_index++;
        }
        replays++;
    }
    void graph_release()
    {
        const int rc = d.graph_end(batch_index);
        if (rc != SDR_OK)
            fprintf(stderr, "graph_end: %s\n", t_err.c_str());
        assert(rc == SDR_OK);
        graph = false;
    }
};

struct Tally {
    std::mutex mu;
    std::vector<int64_t> order;
    bool bad = false;
    void take(const Out &o)
    {
        std::lock_guard<std::mutex> g(mu);
        if (o.stamp != o.batch) {
            fprintf(stderr, "batch %lld delivered with the stamp of batch %lld\n", (long long)o.batch, (long long)o.stamp);
            bad = true;
        }
        order.push_back(o.batch);
    }
    bool complete(int64_t n, const char *what, int64_t first = 0)
    {
        std::lock_guard<std::mutex> g(mu);
        bool ok = !bad && (int64_t)order.size() == n;
        for (int64_t i = 0; ok && i < n; i++)
            ok = order[(size_t)i] == first + i;
        if (!ok)
            fprintf(stderr, "%s: delivered %zu of %lld batches, in order: %s\n", what, order.size(), (long long)n, ok ? "yes" : "NO");
        return ok;
    }
};

// polls until `total` batches have been taken by anybody
void consume(Rig &r, Tally &t, int64_t total, bool wait, unsigned seed, int max_sleep_us)
{
    std::mt19937 rng(seed);
    for (;;) {
        {
            std::lock_guard<std::mutex> g(t.mu);
            if ((int64_t)t.order.size() >= total)
                return;
        }
        Out o;
        const int rc = r.d.poll(&o, wait);
        if (rc == SDR_OK)
            t.take(o);
        else if (rc != SDR_ERR_WOULD_BLOCK) {
            fprintf(stderr, "poll: %d %s\n", rc, t_err.c_str());
            std::lock_guard<std::mutex> g(t.mu);
            t.bad = true;
            return;
        } else {
            std::this_thread::yield();
        }
        if (max_sleep_us && rng() % 4 == 0)
            std::this_thread::sleep_for(std::chrono::microseconds(rng() % (unsigned)max_sleep_us));
    }
}

bool erratic()
{
    Rig r(120);
    Tally t;
    const int64_t n = 600;
    std::thread c([&] { consume(r, t, n, false, 7, 400); });
    std::mt19937 rng(3);
    for (int64_t i = 0; i < n; i++) {
        r.process();
        if (rng() % 16 == 0)
            std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));
    }
    c.join();
    return t.complete(n, "erratic");
}

bool nosync()
{
    Rig r(20);
    Tally t;
    const int64_t n = 200;
    for (int64_t i = 0; i < n; i++)
        r.process();
    bool ok = r.d.parked_count() == (size_t)(n - RING) && r.d.pending() == (int)n;
    if (!ok)
        fprintf(stderr, "nosync: %zu parked, %d pending\n", r.d.parked_count(), r.d.pending());
    consume(r, t, n, true, 1, 0);
    return ok && t.complete(n, "nosync") && r.d.pending() == 0;
}

bool blocking(int consumers)
{
    Rig r(60);
    Tally t;
    const int64_t n = 800;
    std::vector<std::thread> cs;
    for (int i = 0; i < consumers; i++)
        cs.emplace_back([&, i] { consume(r, t, n, true, 11 + (unsigned)i, i ? 150 : 0); });
    for (int64_t i = 0; i < n; i++)
        r.process();
    for (auto &c : cs)
        c.join();
    if (consumers > 1) {  // two consumers each deliver in order, but record in any order: sort before the check
        std::lock_guard<std::mutex> g(t.mu);
        std::sort(t.order.begin(), t.order.end());
    }
    const bool ok = t.complete(n, consumers > 1 ? "two consumers" : "blocking");
    if (consumers == 1 && r.be.copies.load() > n / 4)
        fprintf(stderr, "blocking: note - %ld of %lld batches were parked although a consumer was polling\n", r.be.copies.load(), (long long)n);
    return ok;
}

bool graph_mode()
{
    Rig r(10);
    Tally t;
    for (int i = 0; i < RING; i++)
        r.process();  // six eager batches, unpolled, before the capture
    r.graph_capture();
    for (int k = 0; k < 7; k++)
        r.graph_launch();  // 42 more: beyond GRAPH_PHASES * RING = 24 unpolled ones, launches park the oldest phases
    r.graph_release();     // round 3's defect: parked the youngest batches first once more than 24 were unpolled
    const int64_t after_release = r.batch_index;
    consume(r, t, after_release, false, 5, 0);
    bool ok = t.complete(after_release, "graph: poll after release");
    // eager again, then a second capture with a consumer running beside the replays
    for (int i = 0; i < RING; i++)
        r.process();
    r.graph_capture();
    std::thread c([&] { consume(r, t, after_release + RING + 20 * RING, true, 9, 200); });
    for (int k = 0; k < 20; k++)
        r.graph_launch();
    c.join();
    r.graph_release();
    ok = ok && t.complete(after_release + RING + 20 * RING, "graph: consumer beside replays");
    return ok;
}

// sdr_enable_results on a bank that already delivers (round 4's advice): what was not delivered is discarded - the
// parked entries AND the batches still sitting in their sets - and delivery starts again at the next batch.  With the
// stale entries kept, a real batch parked behind them was never found and poll failed for good.
bool re_enable()
{
    Rig r(15);
    Tally t;
    for (int i = 0; i < 3 * RING + 2; i++)
        r.process();  // nobody polls: the oldest are parked, the youngest six sit in their sets
    bool ok = r.d.parked_count() > 0;
    // sync_bank: the pipeline is drained before the mode is set
    for (int i = 0; i < RING; i++) {
        host::ResultSet &rs = r.d.set(i);
        while (r.be.query(rs.ev_listen) != SDR_OK || r.be.query(rs.ev_peaks) != SDR_OK)
            std::this_thread::yield();
    }
    const int64_t first = r.batch_index;
    r.d.reset(true, first);
    ok = ok && r.d.parked_count() == 0 && r.d.pending() == 0;
    if (!ok)
        fprintf(stderr, "re_enable: %zu parked, %d pending after the reset\n", r.d.parked_count(), r.d.pending());
    // the consumer lags past the ring: batches are parked again, behind nothing stale
    const int64_t n = 4 * RING + 3;
    for (int64_t i = 0; i < n; i++)
        r.process();
    Out o;
    if (r.d.poll(&o, false) != SDR_OK || o.batch != first) {
        fprintf(stderr, "re_enable: the first poll after the reset gave batch %lld, not %lld\n", (long long)o.batch, (long long)first);
        return false;
    }
    t.take(o);
    consume(r, t, n, true, 21, 0);
    return ok && t.complete(n, "re_enable", first) && r.d.pending() == 0;
}

}  // namespace

int main()
{
    struct {
        const char *name;
        bool (*fn)();
    } tests[] = {{"erratic", erratic}, {"nosync", nosync}, {"blocking", [] { return blocking(1); }}, {"two", [] { return blocking(2); }}, {"graph", graph_mode}, {"re_enable", re_enable}};
    int failed = 0;
    for (auto &tc : tests) {
        const bool ok = tc.fn();
        printf("%-9s %s\n", tc.name, ok ? "ok" : "FAILED");
        failed += ok ? 0 : 1;
    }
    return failed ? 1 : 0;
}
