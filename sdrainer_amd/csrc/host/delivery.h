// host/delivery.h — bulk delivery's bookkeeping (the consumer side of rx.Reporter / the listeners' io.Writer,
// rx/rx.go:11-17, rx/receiver.go:123,508-539): where each finished batch's results sit - in the pinned block of the
// buffer set it ran in, or in the host-side queue of "parked" batches - who may take them, and in what order.
//
// Pure C++: device events are opaque handles behind DeliveryBackend, so the same code that libsdrainer_hip.so runs is
// driven by tests/host/test_delivery_model.cpp with fake events and a fake device thread, under ThreadSanitizer, through
// the interleavings the GPU tests cover on hardware (an erratic consumer, short batches without a sync, a caller that
// polls nothing for dozens of batches, graph replays and their release).
//
// Rules:
//  * one producer thread (the caller of sdr_process_* / sdr_graph_*), any number of consumer threads in poll();
//  * batches are delivered in batch order, exactly once; nothing is dropped: a set that is wanted back while its batch
//    is undelivered is waited for and then either taken by a consumer that is polling or copied (used entries only)
//    into the parked queue - parked batches are always older than anything still in a set;
//  * everything below `mu` is guarded by it; the mutex is never held across a wait for the device.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

#include "../../../include/sdrainer_hip.h"

namespace host {

// what the host knows about the batch whose results sit in a block
struct BatchMeta {
    int64_t batch = -1;  // -1: no undelivered batch here
    int64_t first_frame = 0;
    int frames = 0, chunks = 0, count0 = 0, slots = 0;
    std::vector<int64_t> center;  // the bands' centre frequencies when the batch was enqueued (peak frequencies)
};

// the delivery side of one buffer set
struct ResultSet {
    BatchMeta meta;                 // guarded by Delivery::mu
    unsigned char *block = nullptr;  // the set's pinned block (device kernels write it, the host reads it behind the events)
    void *ev_listen = nullptr, *ev_peaks = nullptr;  // recorded behind the kernels that fill the block
};

struct DeliveryBackend {
    virtual ~DeliveryBackend() = default;
    virtual int wait(void *event) = 0;   // SDR_OK once the event has happened, else an error code (message set)
    virtual int query(void *event) = 0;  // SDR_OK: happened; SDR_ERR_WOULD_BLOCK: not yet; else an error code
    // the used entries of a block, copied aside (laid out like the block)
    virtual std::unique_ptr<unsigned char[]> copy_used(const unsigned char *block, const BatchMeta &m) = 0;
    // a block's content into the caller's buffers; anything but SDR_OK leaves the batch undelivered
    virtual int deliver(const unsigned char *block, const BatchMeta &m, void *out) = 0;
    virtual int report(int code, const char *msg) = 0;  // records the message for sdr_last_error(), returns code
};

class Delivery {
public:
    struct Parked {
        BatchMeta meta;
        std::unique_ptr<unsigned char[]> block;
    };

    // ring: the eager pipeline's sets (batch i -> set i % ring); graph_span: the sets graph mode adds behind them
    // (batch i of a capture -> set ring + (i - graph_base) % graph_span)
    Delivery(DeliveryBackend *backend, int ring, int graph_span) : be_(backend), ring_(ring), graph_span_(graph_span) { sets_.resize((size_t)ring); }

    // --- set-up (producer thread, pipeline drained) ---
    ResultSet &set(int i) { return sets_[(size_t)i]; }
    int n_sets() const { return (int)sets_.size(); }
    void grow(int n)  // (a deque: references to the existing sets stay valid while a consumer holds one)
    {
        std::lock_guard<std::mutex> g(mu_);
        if ((int)sets_.size() < n)
            sets_.resize((size_t)n);
    }
    void reset(bool on, int64_t batch_index)  // sdr_enable_results: undelivered batches go with the mode
    {
        std::lock_guard<std::mutex> g(mu_);
        // whatever the mode was: what was not delivered is discarded (the pipeline is drained, sdr_enable_results said so).
        // (Round 4's advice: enabling delivery on a bank that already had it on kept the parked entries and the sets'
        // batches - stale, never at deliver_next_ again - and a real batch parked behind them was never found.)
        for (auto &s : sets_)
            s.meta.batch = -1;
        parked_.clear();
        on_ = on;
        deliver_next_ = batches_enqueued_ = batch_index;
    }
    bool on() const { return on_.load(std::memory_order_relaxed); }

    // --- producer ---
    // the set of batch `batch` (call with the graph window as it is NOW: producer thread, or under the mutex)
    int set_index(int64_t batch) const { return graph_on_ ? ring_ + (int)((batch - graph_base_) % graph_span_) : (int)(batch % ring_); }

    // The set is about to be reused: if its batch was never polled, wait for it and then either let a consumer that is
    // polling take it (a thread inside poll(wait), or one that left poll a moment ago and keeps making progress) or keep
    // a copy of the used entries in the parked queue.
    int park(int set_idx)
    {
        std::unique_lock<std::mutex> guard(mu_);
        ResultSet &S = sets_[(size_t)set_idx];
        if (S.meta.batch < 0)
            return SDR_OK;
        const int64_t batch = S.meta.batch;
        // (not under the mutex: the consumer must be able to take older batches, and this one, meanwhile; nobody but
        // the producer - this thread - puts a new batch into the set)
        guard.unlock();
        int rc = be_->wait(S.ev_listen);
        if (rc == SDR_OK)
            rc = be_->wait(S.ev_peaks);
        guard.lock();
        if (rc != SDR_OK)
            return rc;
        while (S.meta.batch == batch) {
            const bool consumer = pollers_waiting_ > 0 || std::chrono::steady_clock::now() - last_poll_ < std::chrono::milliseconds(2);
            if (!consumer)
                break;
            const int64_t before = deliver_next_;
            // (system_clock: pthread_cond_timedwait, which every ThreadSanitizer intercepts; the steady-clock wait is
            // pthread_cond_clockwait, which older ones do not - they then miss the unlock inside the wait)
            cv_.wait_until(guard, std::chrono::system_clock::now() + std::chrono::microseconds(500));
            if (deliver_next_ == before && S.meta.batch == batch && pollers_waiting_ == 0)
                break;  // it went away
        }
        if (S.meta.batch != batch)
            return SDR_OK;  // delivered meanwhile
        Parked p;
        p.meta = S.meta;
        p.block = be_->copy_used(S.block, S.meta);
        // parked stays sorted by batch: whatever is parked is older than whatever still sits in a set, and the producer
        // parks in batch order - checked here because delivery silently stalls if it is ever violated
        if (!parked_.empty() && parked_.back().meta.batch >= p.meta.batch)
            return be_->report(SDR_ERR_STATE, "internal: results parked out of batch order");
        parked_.push_back(std::move(p));
        S.meta.batch = -1;
        return SDR_OK;
    }

    // a batch's kernels and events are enqueued: its results will appear in the set's block.  `complete` = the listen
    // half is in too (the batch may be handed out by poll()).
    void publish(int set_idx, BatchMeta m, bool complete)
    {
        std::lock_guard<std::mutex> g(mu_);
        const int64_t batch = m.batch;
        sets_[(size_t)set_idx].meta = std::move(m);
        if (complete)
            batches_enqueued_ = batch + 1;
    }
    // the deferred listen half of `batch` has been enqueued
    void complete(int set_idx, int slots, int64_t batch)
    {
        std::lock_guard<std::mutex> g(mu_);
        sets_[(size_t)set_idx].meta.slots = slots;
        batches_enqueued_ = batch + 1;
    }
    // no bulk delivery: only the count moves
    void note_enqueued(int64_t batch_index)
    {
        std::lock_guard<std::mutex> g(mu_);
        batches_enqueued_ = batch_index;
    }

    // graph mode: from `base` on, batches live in the graph sets
    void graph_begin(int64_t base)
    {
        std::lock_guard<std::mutex> g(mu_);
        graph_on_ = true;
        graph_base_ = base;
    }
    // ... until here: what was not polled moves to the parked queue, oldest first.  Only the last graph_span batches can
    // still sit in a set (sdr_graph_launch parks a phase's sets before it reuses them), and the modulo in set_index would
    // map an older batch onto the set of a younger one.
    int graph_end(int64_t batch_index)
    {
        int64_t from;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!graph_on_)
                return SDR_OK;
            from = std::max(std::max(deliver_next_, graph_base_), batch_index - (int64_t)graph_span_);
        }
        if (on_)
            for (int64_t i = from; i < batch_index; i++) {
                const int rc = park(set_index(i));
                if (rc != SDR_OK)
                    return rc;
            }
        std::lock_guard<std::mutex> g(mu_);
        graph_on_ = false;
        return SDR_OK;
    }

    // --- consumer ---
    int pending()
    {
        std::lock_guard<std::mutex> g(mu_);
        return on_ ? (int)(batches_enqueued_ - deliver_next_) : 0;
    }

    // the oldest undelivered batch into the caller's buffers (`out` is the backend's business)
    int poll(void *out, bool wait)
    {
        std::unique_lock<std::mutex> guard(mu_);
        // (a producer that needs a set back gives a polling consumer the chance to take its batch: park)
        struct Polling {
            Delivery *d;
            bool waiting;
            Polling(Delivery *d_, bool w) : d(d_), waiting(w) { d->pollers_waiting_ += waiting ? 1 : 0; }
            ~Polling()  // (the mutex is held again whenever poll returns)
            {
                d->pollers_waiting_ -= waiting ? 1 : 0;
                d->last_poll_ = std::chrono::steady_clock::now();
                d->cv_.notify_all();
            }
        } polling(this, wait);
        if (deliver_next_ >= batches_enqueued_)
            return be_->report(SDR_ERR_WOULD_BLOCK, "no batch waiting");
        // oldest first: parked batches are older than anything still in a set
        if (!parked_.empty() && parked_.front().meta.batch == deliver_next_)
            return poll_parked(out);
        const int64_t want = deliver_next_;  // the batch this call is about, whatever happens while it waits unlocked
        ResultSet &S = sets_[(size_t)set_index(want)];
        if (S.meta.batch != want)
            return be_->report(SDR_ERR_STATE, "results of the next batch are not where they should be");
        for (void *e : {S.ev_listen, S.ev_peaks}) {
            if (wait) {
                // (the producer must not be held up while this thread waits for the device; the set cannot be reused
                // meanwhile - its batch is the oldest undelivered one and parking waits for the same events)
                guard.unlock();
                const int rc = be_->wait(e);
                guard.lock();
                if (rc != SDR_OK)
                    return rc;
                // the producer parked it, or another consumer took it, in the meantime - and the set may even hold a LATER
                // batch by now (whose first event this call has not waited for): only batch `want`, still next, goes on
                if (S.meta.batch != want || deliver_next_ != want)
                    return (!parked_.empty() && parked_.front().meta.batch == deliver_next_) ? poll_parked(out)
                           : deliver_next_ >= batches_enqueued_ ? be_->report(SDR_ERR_WOULD_BLOCK, "no batch waiting")
                                                                : be_->report(SDR_ERR_WOULD_BLOCK, "the batch went to another consumer; poll again");
            } else {
                const int rc = be_->query(e);
                if (rc == SDR_ERR_WOULD_BLOCK)
                    return be_->report(SDR_ERR_WOULD_BLOCK, "the oldest undelivered batch has not finished");
                if (rc != SDR_OK)
                    return rc;
            }
        }
        const int rc = be_->deliver(S.block, S.meta, out);
        if (rc == SDR_OK) {
            S.meta.batch = -1;
            deliver_next_++;
        }
        return rc;
    }

    // the spectral half of the batch in `set_idx` (its listeners have not run: no slots; the batch stays undelivered)
    int poll_peaks(int set_idx, void *out, bool wait)
    {
        ResultSet &S = sets_[(size_t)set_idx];
        if (wait) {
            const int rc = be_->wait(S.ev_peaks);
            if (rc != SDR_OK)
                return rc;
        } else {
            const int rc = be_->query(S.ev_peaks);
            if (rc == SDR_ERR_WOULD_BLOCK)
                return be_->report(SDR_ERR_WOULD_BLOCK, "the batch's cumulations have not finished");
            if (rc != SDR_OK)
                return rc;
        }
        std::lock_guard<std::mutex> guard(mu_);
        BatchMeta m = S.meta;
        m.slots = 0;
        return be_->deliver(S.block, m, out);
    }

    // (tests)
    int64_t deliver_next()
    {
        std::lock_guard<std::mutex> g(mu_);
        return deliver_next_;
    }
    size_t parked_count()
    {
        std::lock_guard<std::mutex> g(mu_);
        return parked_.size();
    }

private:
    int poll_parked(void *out)  // mu_ held: the oldest undelivered batch sits at the front of the parked queue
    {
        if (parked_.empty() || parked_.front().meta.batch != deliver_next_)
            return be_->report(SDR_ERR_STATE, "results of the next batch are not where they should be");
        const Parked &p = parked_.front();
        const int rc = be_->deliver(p.block.get(), p.meta, out);
        if (rc == SDR_OK) {
            parked_.pop_front();
            deliver_next_++;
        }
        return rc;
    }

    DeliveryBackend *be_;
    const int ring_, graph_span_;
    std::mutex mu_;
    std::condition_variable cv_;  // a batch was delivered (a producer about to reuse a set may be waiting for that)
    std::deque<ResultSet> sets_;
    std::deque<Parked> parked_;
    std::atomic<bool> on_{false};
    bool graph_on_ = false;
    int64_t graph_base_ = 0;
    int64_t deliver_next_ = 0;      // batch index poll() hands out next
    int64_t batches_enqueued_ = 0;  // batches whose kernels (both halves) are enqueued
    int pollers_waiting_ = 0;       // threads inside poll(wait = true)
    std::chrono::steady_clock::time_point last_poll_{};  // when poll last returned
};

}  // namespace host
