// sdr_capi.hip — the C ABI of libsdrainer_hip.so (include/sdrainer_hip.h): owns the HBM-resident
// state of a bank of receivers and sequences the kernels on HIP streams.
//
// Scheduling.  The FFT/projection kernel is throughput work that fills the whole chip; everything
// after it is a set of short, strictly ordered chains (float64 noise-floor sums, the rolling means,
// the per-signal decoders) that occupy a handful of CUs for a long time.  Run back to back they would
// leave the chip idle most of the step, so a bank is a software pipeline over four streams (four is
// also the number of hardware queues HIP maps streams to by default; more streams alias and serialise):
//
//   fft     k_fft_psd(i)                                          (the caller's stream)
//   noise   k_window_means(i) -> k_noise_stats(i)
//   peaks   k_thresholds(i) -> k_cumulate(i) -> k_find_peaks(i) (-> k_pack_peaks(i))
//   listen  k_listen_gather(i) -> k_listen_decode(i) (-> k_pack_listen(i))
//
// Batch i's per-batch buffers (psd, tap, frame records, keying bits, peaks ...) live in set i % RING, and one
// event per kernel orders the stages across streams (kDefaultPlan, process_device_body): window means and
// cumulate after the FFT; thresholds after the noise statistics; gather after thresholds; find_peaks after cumulate
// and thresholds; fft(i) after every reader of set i % RING from batch i - RING.  State that is carried from frame to
// frame is only ever touched by one kernel, whose stream keeps it in batch order.  Results leave the device in bulk
// (sdr_enable_results / sdr_poll: two pack kernels per batch into pinned host memory, no pipeline drain) or are read
// after sdr_sync(), which drains every stream.
//
// Three more ways to drive the same body: graph mode (sdr_graph_*: six batches as kernel-only graphs, one per stream,
// four replays in flight over buffer sets of their own), the deferred listen half (sdr_defer_listen ...: the spectral
// stages of a batch first, listeners bound to frames inside it, then the listen stages) and the staged host input
// (sdr_push_* / sdr_process_staged: three pinned staging sets, uploads on a copy stream).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdrainer_hip.h"
#include "host/frequency_mapping.h"
#include "sdr_device.h"
#include "twiddles.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(SDR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        n = count;
        if (count == 0)
            return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess)
            return e;
        e = hipMemset(p, 0, count * sizeof(T));
        if (e != hipSuccess)
            return e;
        // the memset runs on the null stream, which non-blocking streams (the bank's own, the copy stream) do not
        // wait for: finish it before anybody can write into the buffer
        return hipStreamSynchronize(nullptr);
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
    }
};

int ilog2(int n)
{
    int s = 0;
    while ((1 << s) < n)
        s++;
    return s;
}

const char *kKernelNames[sdr::K_COUNT] = {"k_fft_psd",       "k_window_means", "k_noise_stats", "k_thresholds",
                                          "k_listen_gather", "k_cumulate",     "k_find_peaks",  "k_listen_decode"};

// graph mode records the four streams' kernels plus, as a graph of its own in front of the peaks stream's, the thresholds
// (the listen graph starts behind them, not behind the cumulations)
constexpr int G_THRESHOLDS = 4, N_GRAPHS = 5;
constexpr int GRAPH_PHASES = 4;  // graph mode: replays in flight, each with RING buffer sets of its own (sdr_graph_capture)
constexpr int RING = 6;  // per-batch buffer sets in flight (a batch lives about four FFT launches from its FFT to its last result)
enum Stage { S_FFT = 0, S_NOISE, S_LISTEN, S_PEAKS, N_STAGES };  // the bank's streams: four = the hardware queues HIP gives a process; with six streams created (two unused!) the step was 0.49 ms instead of 0.25, with GPU_MAX_HW_QUEUES=8 and five or six in use 0.29-0.60

// Everything one batch produces.
struct BatchSet {
    DevBuf<float> psd;                // [band][max_batch][N] float32(re^2 + im^2), fft-shifted
    DevBuf<float> tap;                // [band][max_batch][L] psd of each listener slot's bin
    DevBuf<double> win_mean;          // [band][max_batch][10]
    DevBuf<sdr_frame_rec> recs;       // [band][max_batch]
    DevBuf<uint64_t> raw_bits, bits;  // [band][L][bit_words] before / after the debouncer
    DevBuf<sdr_edge> edges;           // [band][L][edge_cap]
    DevBuf<uint32_t> edge_counts;     // [band][L] edges produced by this batch
    DevBuf<float> tr_values;          // [band][max_batch][L] (trace only)
    DevBuf<uint8_t> tr_raw, tr_deb;
    DevBuf<sdr::ListenerSlot> slots_before;  // [band][L] the slots as the batch's decoders found them (trace only: sdr_scope_read_decode)
    DevBuf<float> cum_out;            // [band][max_chunks][N]
    DevBuf<sdr::DevPeak> dev_peaks;   // [band][max_chunks][max_peaks]
    DevBuf<int> peak_counts;          // [band][max_chunks]
    hipEvent_t done[sdr::K_COUNT] = {};  // recorded behind each kernel of the batch that used this set
    // bulk delivery (k_results.hip): this set's block of pinned host memory and what the host knows about the batch in it
    unsigned char *res_host = nullptr;
    hipEvent_t res_listen = nullptr, res_peaks = nullptr;
    int64_t res_batch = -1;  // batch whose results sit in res_host, not delivered yet (-1: none)
    int64_t res_first_frame = 0;
    int res_frames = 0, res_chunks = 0, res_count0 = 0, res_slots = 0;
    std::vector<int64_t> res_center;  // the bands' centre frequencies when the batch was enqueued (peak frequencies)
    void release()
    {
        if (res_host)
            (void)hipHostFree(res_host);
        if (res_listen)
            (void)hipEventDestroy(res_listen);
        if (res_peaks)
            (void)hipEventDestroy(res_peaks);
        psd.release();
        tap.release();
        win_mean.release();
        recs.release();
        raw_bits.release();
        bits.release();
        edges.release();
        edge_counts.release();
        tr_values.release();
        tr_raw.release();
        tr_deb.release();
        slots_before.release();
        cum_out.release();
        dev_peaks.release();
        peak_counts.release();
        for (auto &e : done)
            if (e)
                (void)hipEventDestroy(e);
    }
};

}  // namespace

namespace sdr {
int set_error(int code, const char *msg) { return fail(code, msg); }
}  // namespace sdr

struct sdr_bank {
    sdr_config cfg{};
    int logn = 0;
    int device = 0;
    hipStream_t stream[N_STAGES] = {};  // stream[S_FFT] is the caller's (or the null stream)
    bool own_stream[N_STAGES] = {};

    int max_chunks = 0;
    int text_cap = 2048;
    int edge_cap = 0;
    int bit_words = 0;

    DevBuf<fft64::cplx> tw;
    DevBuf<unsigned char> db_tab;   // gomath.h tables of the certified dB shortcut (k_cumulate)
    DevBuf<int32_t> tap_bins;       // [band][L] bin of every listener slot, -1 = free (k_fft_psd tap)
    DevBuf<float> spectrum_row;     // scratch of sdr_read_spectrum
    std::vector<BatchSet> set;  // RING sets; graph mode adds its own (sdr_graph_capture)
    DevBuf<sdr::BandState> band_state;
    DevBuf<sdr::ListenerSlot> slots;  // [band][max_listeners]
    DevBuf<uint16_t> morse;
    DevBuf<uint32_t> text;         // [band][L][text_cap] decoded runes not yet read / delivered
    DevBuf<uint32_t> text_frames;  // [band][L][text_cap] bank frame index of the Tick that wrote each rune
    DevBuf<float> carry[2];  // [band][N] cumulation carried between batches (double buffered)

    std::vector<sdr::BandState> h_band_state;
    std::vector<sdr::ListenerSlot> h_slots;  // authoritative only for active/bin at attach time
    std::vector<int> n_slots;                // high-water mark of used slots per band
    std::vector<int64_t> center_frequency;
    int carry_cur = 0;
    int cum_count = 0;  // cumulationCount, identical for every band of the bank
    int64_t total_frames = 0;
    int64_t batch_index = 0;
    int last_set = 0, last_frames = 0, last_chunks = 0, last_count0 = 0;
    int edge_width = 0;
    int find_peaks = 1;
    bool failed = false;  // a HIP call failed in the middle of a launch sequence: device state is unknown
    DevBuf<sdr::DropCounters> drops;
    // graph mode (sdr_graph_*): RING consecutive batches as one linear, kernel-only hipGraph PER STREAM; GRAPH_PHASES
    // such groups of four graphs, each over buffer sets of its own, so that consecutive replays overlap stage by stage
    DevBuf<sdr::BatchCursor> cursors;  // [GRAPH_PHASES][RING]
    hipGraph_t graph[GRAPH_PHASES][N_GRAPHS] = {};
    hipGraphExec_t graph_exec[GRAPH_PHASES][N_GRAPHS] = {};
    hipGraphNode_t graph_cursor_node[GRAPH_PHASES] = {};  // the FFT graph's first node: writes the replay's cursors
    hipEvent_t phase_done[GRAPH_PHASES][N_GRAPHS] = {};  // recorded behind each graph of a replay
    int64_t graph_base = 0;     // batch_index at the capture
    int64_t graph_replays = 0;  // launches since the capture
    bool graph_ready = false;
    int graph_frames = 0, graph_slots = 0;
    uint64_t attach_gen = 0, graph_attach_gen = 0;  // sdr_attach / sdr_detach calls so far; as of the capture
    // deferred listen half (sdr_defer_listen): the batch whose spectra exist and whose listeners have not run yet
    bool defer_listen = false, listen_pending = false;
    struct PendingListen {
        int set = 0, frames = 0;
        int64_t first_frame = 0, batch = 0;
    } pend;
    std::vector<int> late_attached;  // flattened slot indices bound by sdr_attach_at, not on the device yet
    // bulk delivery
    bool results_on = false;
    sdr::ResultsLayout res_layout{};
    int64_t deliver_next = 0;  // batch index sdr_poll hands out next
    struct Parked {            // a finished batch moved off its ring set before delivery
        int64_t batch, first_frame;
        int frames, chunks, count0, slots;
        std::vector<int64_t> center;
        std::unique_ptr<unsigned char[]> block;  // laid out like a set's block; only the used entries are filled in
    };
    std::deque<Parked> parked;
    // sdr_poll may run on a consumer thread of its own beside the producer's process calls (the reference's
    // Reporter is called from other goroutines too): the delivery bookkeeping - parked, deliver_next, the sets'
    // res_* fields, batch_index as sdr_poll reads it - is guarded by this mutex.
    std::mutex res_mu;
    std::condition_variable res_cv;  // a batch was delivered (a producer about to reuse a set may be waiting for that)
    int pollers_waiting = 0;         // threads inside sdr_poll(wait = 1)
    std::chrono::steady_clock::time_point last_poll{};  // when sdr_poll last returned
    int64_t batches_enqueued = 0;  // == batch_index, published under res_mu

    // Host-fed input (sdr_push_iq / sdr_push_kiwi_snd -> sdr_process_staged).  Three staging sets rotate, so the
    // caller's copy into pinned memory, the upload (its own stream) and the FFT of consecutive batches overlap:
    // nothing on this path waits for the device unless the ring has wrapped around onto work still in flight.
    struct Staging {
        float *h_f32 = nullptr;        // pinned [band][max_batch][2N] float32 frames
        uint8_t *h_raw = nullptr;      // pinned [band][max_batch][2N] big-endian int16 (KiwiSDR payloads), on demand
        DevBuf<float> d_f32;           // [band][n][2N]: what the FFT kernel reads
        DevBuf<uint8_t> d_raw;         // raw payload bytes, unpacked on the device (k_unpack.hip)
        hipEvent_t uploaded = nullptr;  // the upload has left the pinned buffers (they may be overwritten)
        hipEvent_t consumed = nullptr;  // the FFT has read d_f32 (it may be overwritten)
    };
    static constexpr int STAGE_RING = 3;
    Staging stage[STAGE_RING];
    int stage_cur = 0;  // the set sdr_push_* currently fills
    hipStream_t copy_stream = nullptr;
    std::vector<int> staged;
    std::vector<int> staged_kind;  // per band: 0 nothing staged, 1 float32 frames, 2 int16be frames

    bool profiling = false;
    double prof_ms[sdr::K_COUNT] = {};
    int prof_n[sdr::K_COUNT] = {};
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;

    sdr::NoiseGeom noise_geom() const
    {
        sdr::NoiseGeom g;
        g.n = cfg.block_size;
        g.edge = edge_width;
        g.window = (cfg.block_size - 2 * edge_width) / 10;
        const int span = cfg.block_size - 2 * edge_width;
        // window w is evaluated at i = edge + (w+1)*window, which must be < N - edge (dsp/fft.go:226-238)
        g.n_windows = (g.window > 0 && span > 10 * g.window) ? 10 : 9;
        g.inv_n2 = 1.0 / ((double)cfg.block_size * (double)cfg.block_size);
        return g;
    }
};

namespace {

struct ProfScope {
    sdr_bank *b;
    int k;
    hipStream_t s;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(sdr_bank *bank, int kernel, hipStream_t st) : b(bank), k(kernel), s(st)
    {
        if (b->profiling) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, s);
        }
    }
    ~ProfScope()
    {
        if (b->profiling) {
            (void)hipEventRecord(e1, s);
            b->pending.push_back({k, {e0, e1}});
        }
    }
};

void resolve_profile(sdr_bank *b)
{
    for (auto &p : b->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
            b->prof_ms[p.first] += ms;
            b->prof_n[p.first] += 1;
        }
        (void)hipEventDestroy(p.second.first);
        (void)hipEventDestroy(p.second.second);
    }
    b->pending.clear();
}

int flush_late_attached(sdr_bank *b);
int sync_bank(sdr_bank *b)
{
    HIP_TRY(hipSetDevice(b->device));
    {
        const int frc = flush_late_attached(b);  // (whoever synchronises next may read or write slots)
        if (frc)
            return frc;
    }
    for (int s = 0; s < N_STAGES; s++)
        HIP_TRY(hipStreamSynchronize(b->stream[s]));
    resolve_profile(b);
    return SDR_OK;
}

int check_band(sdr_bank *b, int band)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    if (band < 0 || band >= b->cfg.n_bands)
        return fail(SDR_ERR_BAD_ARG, "band out of range");
    return SDR_OK;
}

int check_listener(sdr_bank *b, int band, int lid)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (lid < 0 || lid >= b->n_slots[band])
        return fail(SDR_ERR_BAD_ARG, "listener id out of range");
    return SDR_OK;
}

size_t utf8_encode(uint32_t r, char *out)
{
    if (r < 0x80) {
        out[0] = (char)r;
        return 1;
    }
    if (r < 0x800) {
        out[0] = (char)(0xC0 | (r >> 6));
        out[1] = (char)(0x80 | (r & 0x3F));
        return 2;
    }
    out[0] = (char)(0xE0 | (r >> 12));
    out[1] = (char)(0x80 | ((r >> 6) & 0x3F));
    out[2] = (char)(0x80 | (r & 0x3F));
    return 3;
}

// graph mode: what differs between the batches of a replay lives in device-side cursors; batch k's is written by a node
// of the FFT graph - its first one, for all the replay's batches (no kernel of an earlier replay reads these cursors any
// more: the FFT graph starts behind every reader of its phase)
struct CursorPack {
    sdr::BatchCursor c[RING];
};
__global__ void k_set_cursors(sdr::BatchCursor *dst, CursorPack v)
{
    if (threadIdx.x < RING)
        dst[threadIdx.x] = v.c[threadIdx.x];
}

// sdr_attach_at: the new listener's slot and tap bin reach the device as kernel arguments, in stream order, without a
// synchronous copy (the pipeline keeps running while the host binds listeners)
constexpr int PUT_SLOTS = 16, PUT_BINS = 128;  // per launch (kernel arguments: 16 slots are about 2.5 KB)
struct SlotPack {
    int32_t n;
    int32_t index[PUT_SLOTS];
    sdr::ListenerSlot slot[PUT_SLOTS];
};
struct BinPack {
    int32_t n;
    int32_t index[PUT_BINS], bin[PUT_BINS];
};
__global__ void k_put_slots(sdr::ListenerSlot *slots, SlotPack p)
{
    // (word-wise: a slot is a few dozen words)
    constexpr int W = sizeof(sdr::ListenerSlot) / 4;
    static_assert(sizeof(sdr::ListenerSlot) % 4 == 0, "word copy");
    for (int i = threadIdx.x; i < p.n * W; i += blockDim.x)
        reinterpret_cast<uint32_t *>(slots + p.index[i / W])[i % W] = reinterpret_cast<const uint32_t *>(&p.slot[i / W])[i % W];
}
__global__ void k_put_bins(int32_t *bins, BinPack p)
{
    for (int i = threadIdx.x; i < p.n; i += blockDim.x)
        bins[p.index[i]] = p.bin[i];
}

// The listeners bound by sdr_attach_at since the last flush, to the device: their slots on the listen stream (the only
// stream that touches slots), their tap bins on the FFT stream (read by the next FFT) - a handful of launches whatever
// their number, and no synchronous copy.
int flush_late_attached(sdr_bank *b)
{
    if (b->late_attached.empty())
        return SDR_OK;
    const std::vector<int> &v = b->late_attached;
    for (size_t at = 0; at < v.size(); at += PUT_SLOTS) {
        SlotPack p{};
        p.n = (int32_t)std::min<size_t>(PUT_SLOTS, v.size() - at);
        for (int i = 0; i < p.n; i++) {
            p.index[i] = v[at + i];
            p.slot[i] = b->h_slots[(size_t)v[at + i]];
        }
        hipLaunchKernelGGL(k_put_slots, dim3(1), dim3(256), 0, b->stream[S_LISTEN], b->slots.p, p);
        HIP_TRY(hipGetLastError());
    }
    for (size_t at = 0; at < v.size(); at += PUT_BINS) {
        BinPack p{};
        p.n = (int32_t)std::min<size_t>(PUT_BINS, v.size() - at);
        for (int i = 0; i < p.n; i++) {
            p.index[i] = v[at + i];
            p.bin[i] = b->h_slots[(size_t)v[at + i]].bin;
        }
        hipLaunchKernelGGL(k_put_bins, dim3(1), dim3(128), 0, b->stream[S_FFT], b->tap_bins.p, p);
        HIP_TRY(hipGetLastError());
    }
    b->late_attached.clear();
    return SDR_OK;
}

enum Parts { PART_SPECTRA = 1, PART_LISTEN = 2, PART_ALL = 3 };
int process_device_body(sdr_bank *b, const float *iq_dev, int n_frames, int in_stride, int capture_k = -1, int capture_stage = -1,
                        int parts = PART_ALL);

// A failure after the first launch leaves the pipeline half enqueued (some stages of this batch ran, the
// carried state of others did not advance): no later batch can be trusted, so the bank refuses further work.
int process_device_impl(sdr_bank *b, const float *iq_dev, int n_frames, int in_stride)
{
    if (b->failed)
        return fail(SDR_ERR_STATE, "an earlier process call failed half way; destroy the bank");
    if (b->graph_ready)
        return fail(SDR_ERR_STATE, "a graph is captured: process through sdr_graph_launch, or sdr_graph_release first");
    if (b->listen_pending)
        return fail(SDR_ERR_STATE, "the previous batch still waits for its listen half (sdr_process_listen)");
    const int rc = process_device_body(b, iq_dev, n_frames, in_stride, -1, -1, b->defer_listen ? PART_SPECTRA : PART_ALL);
    if (rc == SDR_ERR_HIP)
        b->failed = true;
    return rc;
}

// ---- bulk delivery -------------------------------------------------------------------------------------------
sdr::ResultsLayout make_results_layout(const sdr_bank *b)
{
    const sdr_config &c = b->cfg;
    sdr::ResultsLayout l{};
    l.max_listeners = c.max_listeners;
    l.max_chunks = b->max_chunks;
    l.max_peaks = c.max_peaks;
    l.edge_cap = b->edge_cap;
    l.text_cap = b->text_cap;
    const size_t B = (size_t)c.n_bands, L = (size_t)c.max_listeners, C = (size_t)b->max_chunks;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 63) & ~(size_t)63;
        return at;
    };
    l.off_drops = take(sizeof(sdr::DropCounters));
    l.off_peak_counts = take(B * C * 2 * sizeof(int));
    l.off_peaks = take(B * C * (size_t)c.max_peaks * sizeof(sdr::DevPeak));
    l.off_edge_counts = take(B * L * sizeof(uint32_t));
    l.off_edges = take(B * L * (size_t)b->edge_cap * sizeof(sdr_edge));
    l.off_text_counts = take(B * L * sizeof(uint32_t));
    l.off_text = take(B * L * (size_t)b->text_cap * sizeof(uint32_t));
    l.off_text_frames = take(B * L * (size_t)b->text_cap * sizeof(uint32_t));
    l.bytes = off;
    return l;
}

// The set is about to be reused while its results were never polled: wait for them, and then either let a consumer
// thread that is polling take the batch (it is blocked on the same events, or between two polls) or keep a copy on the
// host (the reference's io.Writer never drops; a caller that enqueues many batches and polls afterwards must find them
// all).  Only the used entries of the block are copied.
int park_results(sdr_bank *b, BatchSet &S)
{
    std::unique_lock<std::mutex> guard(b->res_mu);
    if (S.res_batch < 0)
        return SDR_OK;
    const int64_t batch = S.res_batch;
    // (not under the mutex: the consumer must be able to take older batches, and this one, meanwhile; nobody but this
    // thread - the producer - puts a new batch into the set)
    guard.unlock();
    hipError_t we = hipEventSynchronize(S.res_listen);
    if (we == hipSuccess)
        we = hipEventSynchronize(S.res_peaks);
    guard.lock();
    HIP_TRY(we);
    // a consumer is at work if a thread sits in sdr_poll(wait) or left it a moment ago: it delivers in order, so it
    // gets to this batch as long as it makes progress
    while (S.res_batch == batch) {
        const bool consumer = b->pollers_waiting > 0 || std::chrono::steady_clock::now() - b->last_poll < std::chrono::milliseconds(2);
        if (!consumer)
            break;
        const int64_t before = b->deliver_next;
        b->res_cv.wait_for(guard, std::chrono::microseconds(500));
        if (b->deliver_next == before && S.res_batch == batch && b->pollers_waiting == 0)
            break;  // it went away
    }
    if (S.res_batch != batch)
        return SDR_OK;
    const sdr::ResultsLayout &lay = b->res_layout;
    const sdr_config &c = b->cfg;
    sdr_bank::Parked p;
    p.batch = S.res_batch;
    p.first_frame = S.res_first_frame;
    p.frames = S.res_frames;
    p.chunks = S.res_chunks;
    p.count0 = S.res_count0;
    p.slots = S.res_slots;
    p.center = S.res_center;
    p.block.reset(new unsigned char[lay.bytes]);
    unsigned char *dst = p.block.get();
    const unsigned char *src = S.res_host;
    auto copy = [&](size_t off, size_t bytes) { memcpy(dst + off, src + off, bytes); };
    const size_t B = (size_t)c.n_bands, L = (size_t)c.max_listeners, C = (size_t)lay.max_chunks;
    copy(lay.off_drops, sizeof(sdr::DropCounters));
    copy(lay.off_peak_counts, B * C * 2 * sizeof(int));
    copy(lay.off_edge_counts, B * L * sizeof(uint32_t));
    copy(lay.off_text_counts, B * L * sizeof(uint32_t));
    const int *peak_counts = reinterpret_cast<const int *>(src + lay.off_peak_counts);
    const uint32_t *edge_counts = reinterpret_cast<const uint32_t *>(src + lay.off_edge_counts);
    const uint32_t *text_counts = reinterpret_cast<const uint32_t *>(src + lay.off_text_counts);
    for (size_t band = 0; band < B; band++) {
        for (size_t ch = 0; ch < (size_t)p.chunks && ch < C; ch++) {
            const size_t idx = band * C + ch;
            const size_t n = (size_t)std::min(std::max(peak_counts[2 * idx], 0), lay.max_peaks);
            copy(lay.off_peaks + idx * (size_t)lay.max_peaks * sizeof(sdr::DevPeak), n * sizeof(sdr::DevPeak));
        }
        for (size_t l = 0; l < (size_t)p.slots && l < L; l++) {
            const size_t idx = band * L + l;
            const size_t ne = std::min<size_t>(edge_counts[idx], (size_t)lay.edge_cap), nr = std::min<size_t>(text_counts[idx], (size_t)lay.text_cap);
            copy(lay.off_edges + idx * (size_t)lay.edge_cap * sizeof(sdr_edge), ne * sizeof(sdr_edge));
            copy(lay.off_text + idx * (size_t)lay.text_cap * sizeof(uint32_t), nr * sizeof(uint32_t));
            copy(lay.off_text_frames + idx * (size_t)lay.text_cap * sizeof(uint32_t), nr * sizeof(uint32_t));
        }
    }
    b->parked.push_back(std::move(p));
    S.res_batch = -1;
    return SDR_OK;
}

// Which of the bank's four streams each kernel runs on (index = sdr::KernelId).  The step is as long as the
// longest stream, and kernels that carry state from batch to batch (thresholds, decode, cumulate) must keep their
// stream so that the stream orders the batches.  SDR_DIAG builds read an override from SDR_DIAG_PLAN (eight
// digits) to try other plans.
constexpr int kDefaultPlan[sdr::K_COUNT] = {
    /* fft */ S_FFT, /* window means */ S_NOISE, /* noise stats */ S_NOISE, /* thresholds */ S_PEAKS,
    /* gather */ S_LISTEN, /* cumulate */ S_PEAKS, /* find peaks */ S_PEAKS, /* decode */ S_LISTEN};

// capture_stage: while capturing, ONE stream records at a time (sdr_graph_capture walks the batches once per stream):
// only the kernels of that stream are issued, everything else of the batch is skipped in that walk, and NO event is
// recorded or waited for - what orders the streams of a replay are events around whole graphs (sdr_graph_launch).
// capture_k >= 0: the call is being recorded into a graph as its batch number capture_k (sdr_graph_capture).  Then
// the batch uses buffer set RING + capture_k, everything that differs from batch to batch comes from the device-side
// cursor of that number instead of the launch parameters, grids cover the most chunks a batch of this length can
// complete, nothing is asked of the host (no event queries, no profiling, no parking) and no host state changes.
// parts: PART_SPECTRA leaves the batch's listeners for a later PART_LISTEN call (sdr_defer_listen / sdr_process_listen:
// the host binds listeners to peaks of this very batch in between, rx/receiver.go:409-426); the later call takes the
// batch's set, length and first frame from b->pend.
int process_device_body(sdr_bank *b, const float *iq_dev, int n_frames, int in_stride, int capture_k, int capture_stage, int parts)
{
    const bool cap = capture_k >= 0;
    const bool do_spectra = (parts & PART_SPECTRA) != 0, do_listen = (parts & PART_LISTEN) != 0;
    if (!do_spectra)
        n_frames = b->pend.frames;
    const sdr::BatchCursor *cur = cap ? b->cursors.p + capture_k : nullptr;
    const sdr_config &c = b->cfg;
    if (n_frames <= 0)
        return SDR_OK;
    // an armed stage event must never outlive this call (an error return between SDR_ARM and the launch would
    // otherwise hand it to the next kernel this thread launches, possibly on another bank)
    struct DisarmOnExit {
        ~DisarmOnExit() { sdr::t_done_event = nullptr; }
    } disarm_on_exit;
    if (n_frames > c.max_batch_frames)
        return fail(SDR_ERR_BAD_ARG, "n_frames exceeds max_batch_frames");
    HIP_TRY(hipSetDevice(b->device));
    if (!cap) {
        const int frc = flush_late_attached(b);
        if (frc)
            return frc;
    }
    const int B = c.n_bands, N = c.block_size, stride = c.max_batch_frames;
    const sdr::NoiseGeom ng = b->noise_geom();
    const int si = cap ? RING + capture_k : do_spectra ? (int)(b->batch_index % RING) : b->pend.set;  // (capture: the sets sdr_graph_capture added)
    const int64_t first_frame = do_spectra ? b->total_frames : b->pend.first_frame;
    BatchSet &S = b->set[si];
    int plan[sdr::K_COUNT];
    for (int k = 0; k < sdr::K_COUNT; k++)
        plan[k] = kDefaultPlan[k];
    // is kernel k part of the graph that is recording (always, outside a capture)?
#define SDR_ON(k) (!cap || (capture_stage == G_THRESHOLDS ? (k) == sdr::K_THRESHOLDS : (plan[k] == capture_stage && (k) != sdr::K_THRESHOLDS)))
#if defined(SDR_DIAG)
    // diagnostic builds only (tools/abl): SDR_DIAG_SKIP = bit mask of kernel ids not to launch, to see
    // which stage holds the pipelined step up (results are wrong by construction); SDR_DIAG_PLAN = stream plan.
    static const int diag_skip = getenv("SDR_DIAG_SKIP") ? atoi(getenv("SDR_DIAG_SKIP")) : 0;
    if (const char *e = getenv("SDR_DIAG_PLAN"))
        for (int k = 0; k < sdr::K_COUNT && e[k] >= '0' && e[k] < '0' + N_STAGES; k++)
            plan[k] = e[k] - '0';
#define SDR_LAUNCH(id, call) \
    do {                     \
        if (!(diag_skip >> (id) & 1) && SDR_ON(id)) \
            HIP_TRY(call);   \
    } while (0)
#else
#define SDR_LAUNCH(id, call)  \
    do {                      \
        if (SDR_ON(id))       \
            HIP_TRY(call);    \
    } while (0)
#endif
    auto stream_of = [&](int k) { return b->stream[plan[k]]; };
    // kernel k of this batch may start once kernel `dep` of this batch is done (nothing to do on the same stream)
    auto after = [&](int k, int dep) -> int {
        if (stream_of(k) != stream_of(dep) && !cap)
            HIP_TRY(hipStreamWaitEvent(stream_of(k), S.done[dep], 0));
        return SDR_OK;
    };
#define SDR_AFTER(k, dep)          \
    do {                           \
        int _rc = after((k), (dep)); \
        if (_rc)                   \
            return _rc;            \
    } while (0)
    // A stage's event is recorded by its kernel's own dispatch (sdr::launch_kernel, sdr_device.h): SDR_ARM hands the
    // event to the next launch, SDR_DONE records it the ordinary way if no launch took it (stage left out, capture)
    static const bool stop_events = !(getenv("SDR_STOP_EVENTS") && atoi(getenv("SDR_STOP_EVENTS")) == 0);
    const bool ride = stop_events && !cap;
#define SDR_ARM(k) (sdr::t_done_event = ride ? S.done[k] : nullptr)
#define SDR_DONE(k)                                                                                   \
    do {                                                                                              \
        if ((!ride || sdr::t_done_event) && !cap) {                                                   \
            sdr::t_done_event = nullptr;                                                              \
            HIP_TRY(hipEventRecord(S.done[k], stream_of(k)));                                         \
        }                                                                                             \
    } while (0)

    // FFT + PSD + tap, once every reader of this set (batch i - RING) is done with it (with RING sets the
    // previous user is four batches back and has almost always finished: ask the host first, a barrier packet in
    // the FFT queue costs the command processor tens of microseconds)
    // (inside a graph a set is used once per replay and replays are serialised by their stream)
    // A caller that enqueues faster than the GPU works is soon more than RING batches ahead; then these events have
    // not happened yet at enqueue time and the FFT queue gets barrier packets: one per other stream (its last stage
    // stands for the stream), not one per stage - with nothing else running that was 0.200 -> 0.177 ms per step for a
    // 0.166 ms kernel.  SDR_HOST_THROTTLE=1 (development) makes the HOST wait instead (the call blocks until the
    // set is free, the FFT queue holds kernels only): 0.161 ms with nothing else running, but 0.237 against 0.234
    // with the whole pipeline, where the FFT launches are spaced by the CUs the tail holds, not by their queue.
    static const bool host_waits = getenv("SDR_HOST_THROTTLE") && atoi(getenv("SDR_HOST_THROTTLE")) != 0;
    int max_slots = 0, slots_in_use = 0;
    for (int i = 0; i < B; i++) {
        max_slots = std::max(max_slots, b->n_slots[i]);
        slots_in_use += b->n_slots[i];
    }
    if (do_spectra) {
    // With bulk delivery on, the set's previous batch must have been delivered (or be parked) before its block is
    // written again - and a delivered batch is a finished one: every reader of the set is done, the queries below
    // succeed and the FFT queue gets no barrier packets at all (each costs the command processor microseconds between
    // two FFT kernels, and the FFT queue is the one that bounds the step).
    static const bool park_first = !(getenv("SDR_PARK_FIRST") && atoi(getenv("SDR_PARK_FIRST")) == 0);
    if (b->results_on && !cap && park_first) {
        const int prc = park_results(b, S);
        if (prc)
            return prc;
    }
    {
        // the last stage launched on a stream stands for all of that stream's
        static const int launch_order[] = {sdr::K_WINDOW_MEANS, sdr::K_NOISE_STATS, sdr::K_THRESHOLDS, sdr::K_LISTEN_GATHER,
                                           sdr::K_LISTEN_DECODE, sdr::K_CUMULATE,   sdr::K_FIND_PEAKS};
        int last_on[N_STAGES];
        for (int &l : last_on)
            l = -1;
        for (int k : launch_order)
            last_on[plan[k]] = k;
        for (int st = 0; st < N_STAGES; st++) {
            const int k = last_on[st];
            if (k < 0 || st == plan[sdr::K_FFT])
                continue;
            if (cap)  // (sdr_graph_launch waits for the earlier replay that used this phase's sets)
                continue;
            if (hipEventQuery(S.done[k]) == hipSuccess)
                continue;
            if (host_waits)
                HIP_TRY(hipEventSynchronize(S.done[k]));
            else
                HIP_TRY(hipStreamWaitEvent(stream_of(sdr::K_FFT), S.done[k], 0));
        }
    }
    if (b->results_on && !cap && !park_first) {
        const int prc = park_results(b, S);
        if (prc)
            return prc;
    }
    if (cap && SDR_ON(sdr::K_FFT) && capture_k % RING == 0)  // the replay's cursors, in front of its first FFT
        hipLaunchKernelGGL(k_set_cursors, dim3(1), dim3(64), 0, stream_of(sdr::K_FFT), b->cursors.p + capture_k, CursorPack{});
    {
        ProfScope ps(b, sdr::K_FFT, stream_of(sdr::K_FFT));
        SDR_ARM(sdr::K_FFT);
        const sdr::FftTap tap{b->tap_bins.p, S.tap.p, max_slots, c.max_listeners};
        SDR_LAUNCH(sdr::K_FFT, sdr::launch_fft(b->logn, iq_dev, cur, b->tw.p, S.psd.p, n_frames, B, in_stride, stride, tap,
                                               stream_of(sdr::K_FFT)));
    }
    SDR_DONE(sdr::K_FFT);

    // noise floor (stateless per batch), then the rolling means -> thresholds, in batch order
    SDR_AFTER(sdr::K_WINDOW_MEANS, sdr::K_FFT);
    {
        ProfScope ps(b, sdr::K_WINDOW_MEANS, stream_of(sdr::K_WINDOW_MEANS));
        SDR_ARM(sdr::K_WINDOW_MEANS);
        SDR_LAUNCH(sdr::K_WINDOW_MEANS, sdr::launch_window_means(S.psd.p, S.win_mean.p, ng, n_frames, B, stride,
                                                                 stream_of(sdr::K_WINDOW_MEANS)));
    }
    SDR_DONE(sdr::K_WINDOW_MEANS);
    SDR_AFTER(sdr::K_NOISE_STATS, sdr::K_WINDOW_MEANS);
    {
        ProfScope ps(b, sdr::K_NOISE_STATS, stream_of(sdr::K_NOISE_STATS));
        SDR_ARM(sdr::K_NOISE_STATS);
        SDR_LAUNCH(sdr::K_NOISE_STATS, sdr::launch_noise_stats(S.psd.p, S.win_mean.p, S.recs.p, ng, n_frames, B, stride,
                                                               stream_of(sdr::K_NOISE_STATS)));
    }
    SDR_DONE(sdr::K_NOISE_STATS);
    SDR_AFTER(sdr::K_THRESHOLDS, sdr::K_NOISE_STATS);
    {
        ProfScope ps(b, sdr::K_THRESHOLDS, stream_of(sdr::K_THRESHOLDS));
        SDR_ARM(sdr::K_THRESHOLDS);
        SDR_LAUNCH(sdr::K_THRESHOLDS, sdr::launch_thresholds(S.recs.p, b->band_state.p, n_frames, B, stride,
                                                             stream_of(sdr::K_THRESHOLDS)));
    }
    SDR_DONE(sdr::K_THRESHOLDS);
    }  // do_spectra

    // per-signal envelope + decoder
    sdr::ListenGeom lg;
    lg.n = N;
    lg.stride = stride;
    lg.max_listeners = c.max_listeners;
    lg.text_cap = b->text_cap;
    lg.edge_cap = b->edge_cap;
    lg.bit_words = b->bit_words;
    lg.trace = c.trace;
    lg.frame_base = (uint32_t)first_frame;
    if (do_listen) {
    SDR_AFTER(sdr::K_LISTEN_GATHER, sdr::K_THRESHOLDS);
    SDR_AFTER(sdr::K_LISTEN_GATHER, sdr::K_FFT);
    // (armed whether or not the stage launches: SDR_DONE records a stage event nobody took the ordinary way, and a
    // stage left out must still publish its event - the set-reuse wait reads the last stage of each stream)
    SDR_ARM(sdr::K_LISTEN_GATHER);
    if (max_slots > 0) {
        ProfScope ps(b, sdr::K_LISTEN_GATHER, stream_of(sdr::K_LISTEN_GATHER));
        SDR_LAUNCH(sdr::K_LISTEN_GATHER, sdr::launch_listen_gather(S.tap.p, S.psd.p, S.recs.p, b->slots.p, b->db_tab.p, S.raw_bits.p, S.tr_values.p,
                                                                   S.tr_raw.p, cur, lg, n_frames, max_slots, B,
                                                                   stream_of(sdr::K_LISTEN_GATHER)));
    }
    SDR_DONE(sdr::K_LISTEN_GATHER);
    SDR_AFTER(sdr::K_LISTEN_DECODE, sdr::K_LISTEN_GATHER);
    if (c.trace && max_slots > 0 && SDR_ON(sdr::K_LISTEN_DECODE))  // the decoders' state before this batch: the decoder scope replays from it
        HIP_TRY(hipMemcpyAsync(S.slots_before.p, b->slots.p, sizeof(sdr::ListenerSlot) * (size_t)B * (size_t)c.max_listeners,
                               hipMemcpyDeviceToDevice, stream_of(sdr::K_LISTEN_DECODE)));
    if (!b->results_on)
        SDR_ARM(sdr::K_LISTEN_DECODE);
    if (max_slots > 0) {
        ProfScope ps(b, sdr::K_LISTEN_DECODE, stream_of(sdr::K_LISTEN_DECODE));
        SDR_LAUNCH(sdr::K_LISTEN_DECODE, sdr::launch_listen_decode(b->slots.p, b->morse.p, S.raw_bits.p, S.bits.p, b->text.p,
                                                                   b->text_frames.p, S.edges.p, S.edge_counts.p, S.tr_deb.p, b->drops.p, cur,
                                                                   lg, n_frames, B, slots_in_use, stream_of(sdr::K_LISTEN_DECODE)));
    }
    if (b->results_on && SDR_ON(sdr::K_LISTEN_DECODE)) {
        // delivery of this batch's edges and runes, behind the decoder on its stream; the decoder's event is
        // recorded behind it so that the set is not reused before the copy to the host has happened
        SDR_ARM(sdr::K_LISTEN_DECODE);
        HIP_TRY(sdr::launch_pack_listen(b->slots.p, S.edges.p, S.edge_counts.p, b->text.p, b->text_frames.p, b->drops.p, b->res_layout, max_slots, B,
                                        S.res_host, stream_of(sdr::K_LISTEN_DECODE)));
        if (!cap)  // (a replay records it behind the listen graph)
            HIP_TRY(hipEventRecord(S.res_listen, stream_of(sdr::K_LISTEN_DECODE)));
    }
    SDR_DONE(sdr::K_LISTEN_DECODE);
    }  // do_listen
    if (!do_spectra) {
        // the batch is complete: sdr_poll may have it
        std::lock_guard<std::mutex> guard(b->res_mu);
        S.res_slots = max_slots;
        b->batches_enqueued = b->pend.batch + 1;
        b->listen_pending = false;
        return SDR_OK;
    }

    // dB projection + cumulation, peak scan (rx/receiver.go:404-409,459-460)
    const int count0 = b->cum_count;
    const int first_len = SDR_CUMULATION_SIZE - count0;
    int n_slots_c = 1, n_chunks = 0;
    if (n_frames >= first_len) {
        n_chunks = 1 + (n_frames - first_len) / SDR_CUMULATION_SIZE;
        const int rem = (n_frames - first_len) % SDR_CUMULATION_SIZE;
        n_slots_c = n_chunks + (rem > 0 ? 1 : 0);
    }
    if (cap) {  // whatever cumulationCount the replayed batch starts at
        n_chunks = sdr::chunks_completed(SDR_CUMULATION_SIZE - 1, n_frames);
        n_slots_c = n_chunks + 1;
    }
    SDR_AFTER(sdr::K_CUMULATE, sdr::K_FFT);
    {
        ProfScope ps(b, sdr::K_CUMULATE, stream_of(sdr::K_CUMULATE));
        SDR_ARM(sdr::K_CUMULATE);
        sdr::CumGeom cg{N, stride, n_frames, count0, b->max_chunks};
        SDR_LAUNCH(sdr::K_CUMULATE, sdr::launch_cumulate(S.psd.p, b->db_tab.p, b->carry[0].p, b->carry[1].p, b->carry_cur,
                                                         S.cum_out.p, cur, cg, n_slots_c, B, stream_of(sdr::K_CUMULATE)));
    }
    SDR_DONE(sdr::K_CUMULATE);
    const int new_count = (count0 + n_frames) % SDR_CUMULATION_SIZE;
    SDR_AFTER(sdr::K_FIND_PEAKS, sdr::K_CUMULATE);
    if (!b->results_on)
        SDR_ARM(sdr::K_FIND_PEAKS);
    if (b->find_peaks && n_chunks > 0) {
        SDR_AFTER(sdr::K_FIND_PEAKS, sdr::K_THRESHOLDS);  // needs the completing frame's peak threshold
        ProfScope ps(b, sdr::K_FIND_PEAKS, stream_of(sdr::K_FIND_PEAKS));
        sdr::PeakGeom pg{N, stride, count0, b->max_chunks, c.max_peaks};
        SDR_LAUNCH(sdr::K_FIND_PEAKS, sdr::launch_find_peaks(S.cum_out.p, S.recs.p, S.dev_peaks.p, S.peak_counts.p, cur, pg, n_frames,
                                                             n_chunks, B, stream_of(sdr::K_FIND_PEAKS)));
    }
    if (b->results_on && SDR_ON(sdr::K_FIND_PEAKS)) {
        SDR_ARM(sdr::K_FIND_PEAKS);
        HIP_TRY(sdr::launch_pack_peaks(S.dev_peaks.p, S.peak_counts.p, cur, b->res_layout, b->find_peaks, n_frames, n_chunks, B,
                                       S.res_host, stream_of(sdr::K_FIND_PEAKS)));
        if (!cap) {
            HIP_TRY(hipEventRecord(S.res_peaks, stream_of(sdr::K_FIND_PEAKS)));
            std::lock_guard<std::mutex> guard(b->res_mu);
            S.res_batch = b->batch_index;
            S.res_first_frame = b->total_frames;
            S.res_frames = n_frames;
            S.res_chunks = n_chunks;
            S.res_count0 = count0;
            S.res_slots = do_listen ? max_slots : 0;  // (sdr_poll_peaks delivers the spectral half; the listen half fills this in)
            S.res_center = b->center_frequency;
            if (do_listen)
                b->batches_enqueued = b->batch_index + 1;
        }
    }
    SDR_DONE(sdr::K_FIND_PEAKS);
#undef SDR_AFTER
#undef SDR_DONE
#undef SDR_ARM
#undef SDR_LAUNCH
#undef SDR_ON

    if (cap)
        return SDR_OK;
    // every launch of the batch is enqueued: commit the host's view of the carried state in one go.
    // The carry buffer flips only when this batch wrote a new partial cumulation; if the batch ended
    // exactly on a chunk boundary the next batch starts from zero (count0 == 0 ignores the carry)
    if (new_count != 0)
        b->carry_cur ^= 1;
    b->cum_count = new_count;
    b->last_set = si;
    b->last_frames = n_frames;
    b->last_chunks = n_chunks;
    b->last_count0 = count0;
    if (!do_listen) {
        b->pend.set = si;
        b->pend.frames = n_frames;
        b->pend.first_frame = b->total_frames;
        b->pend.batch = b->batch_index;
        b->listen_pending = true;
    }
    b->total_frames += n_frames;
    b->batch_index++;
    if (!b->results_on) {
        std::lock_guard<std::mutex> guard(b->res_mu);
        b->batches_enqueued = b->batch_index;
    }
    return SDR_OK;
}

// graph mode: the values that differ between the batches of a replay, written to the device-side cursors by the
// graph's first node from its kernel arguments
// (one node per batch; its kernel argument is what a replay updates)

void drop_graphs(sdr_bank *b)
{
    for (int ph = 0; ph < GRAPH_PHASES; ph++) {
        for (int st = 0; st < N_GRAPHS; st++) {
            if (b->graph_exec[ph][st])
                (void)hipGraphExecDestroy(b->graph_exec[ph][st]);
            if (b->graph[ph][st])
                (void)hipGraphDestroy(b->graph[ph][st]);
            b->graph_exec[ph][st] = nullptr;
            b->graph[ph][st] = nullptr;
        }
        b->graph_cursor_node[ph] = nullptr;
    }
}

// the buffer set of batch `batch` (graph mode: phase-major, the sets behind the eager ring's)
inline int set_index(const sdr_bank *b, int64_t batch)
{
    if (!b->graph_ready)
        return (int)(batch % RING);
    return RING + (int)((batch - b->graph_base) % (GRAPH_PHASES * RING));
}
}  // namespace

// one batch's buffers, stage events and (if bulk delivery is on) its block of pinned host memory
static hipError_t alloc_set(sdr_bank *b, BatchSet &S)
{
    const sdr_config &c = b->cfg;
    const size_t B = (size_t)c.n_bands, F = (size_t)c.max_batch_frames, L = (size_t)c.max_listeners, N = (size_t)c.block_size;
    hipError_t e = hipSuccess;
#define SET_ALLOC(buf, count)            \
    do {                                 \
        if (e == hipSuccess)             \
            e = (buf).alloc(count);      \
    } while (0)
    SET_ALLOC(S.psd, B * F * N);
    SET_ALLOC(S.tap, B * F * std::max<size_t>(L, 1));
    SET_ALLOC(S.win_mean, B * F * 10);
    SET_ALLOC(S.recs, B * F);
    SET_ALLOC(S.raw_bits, B * L * (size_t)b->bit_words);
    SET_ALLOC(S.bits, B * L * (size_t)b->bit_words);
    SET_ALLOC(S.edges, B * L * (size_t)b->edge_cap);
    SET_ALLOC(S.edge_counts, B * L);
    if (c.trace) {
        SET_ALLOC(S.tr_values, B * F * L);
        SET_ALLOC(S.tr_raw, B * F * L);
        SET_ALLOC(S.tr_deb, B * F * L);
        SET_ALLOC(S.slots_before, B * L);
    }
    SET_ALLOC(S.cum_out, B * (size_t)b->max_chunks * N);
    SET_ALLOC(S.dev_peaks, B * (size_t)b->max_chunks * (size_t)c.max_peaks);
    SET_ALLOC(S.peak_counts, B * (size_t)b->max_chunks);
#undef SET_ALLOC
#ifndef SDR_STAGE_EVENT_FLAGS
#define SDR_STAGE_EVENT_FLAGS hipEventDisableTiming
#endif
    for (auto &ev : S.done)
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&ev, SDR_STAGE_EVENT_FLAGS);
    if (e == hipSuccess && b->res_layout.bytes && !S.res_host) {
        e = hipHostMalloc(reinterpret_cast<void **>(&S.res_host), b->res_layout.bytes, hipHostMallocDefault);
        if (e == hipSuccess) {
            memset(S.res_host, 0, b->res_layout.bytes);
            e = hipEventCreateWithFlags(&S.res_listen, hipEventDisableTiming);
        }
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&S.res_peaks, hipEventDisableTiming);
    }
    return e;
}

extern "C" {
#pragma GCC visibility push(default)

const char *sdr_last_error(void) { return g_last_error.c_str(); }
int sdr_abi_version(void) { return SDR_ABI_VERSION; }
const char *sdr_kernel_name(int kernel) { return (kernel >= 0 && kernel < sdr::K_COUNT) ? kKernelNames[kernel] : ""; }

int sdr_create(const sdr_config *cfg, sdr_bank **out)
{
    if (!cfg || !out)
        return fail(SDR_ERR_BAD_ARG, "null argument");
    if (cfg->struct_size != (int32_t)sizeof(sdr_config))
        return fail(SDR_ERR_BAD_ARG, "sdr_config.struct_size mismatch (ABI)");
    const int N = cfg->block_size;
    if (N < 512 || N > 16384 || (N & (N - 1)))
        return fail(SDR_ERR_BAD_SIZE, "block_size must be a power of two in [512, 16384]");
    if (cfg->n_bands < 1 || cfg->sample_rate < 1 || cfg->max_batch_frames < 1 || cfg->max_listeners < 0 ||
        cfg->max_peaks < 1)
        return fail(SDR_ERR_BAD_ARG, "non-positive geometry");
    if (cfg->edge_width < 0 || N - 2 * cfg->edge_width < 10)
        return fail(SDR_ERR_BAD_ARG, "edge_width leaves fewer than 10 bins: the reference's windowSize would be 0 (NaN)");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(SDR_ERR_BAD_ARG, "device_id out of range");
    HIP_TRY(hipSetDevice(cfg->device_id));

    sdr_bank *b = new sdr_bank();
    b->cfg = *cfg;
    b->device = cfg->device_id;
    b->logn = ilog2(N);
    b->edge_width = cfg->edge_width;
    b->find_peaks = cfg->find_peaks;
    const size_t B = (size_t)cfg->n_bands, F = (size_t)cfg->max_batch_frames, L = (size_t)cfg->max_listeners;
    b->max_chunks = cfg->max_batch_frames / SDR_CUMULATION_SIZE + 2;
    b->edge_cap = cfg->max_batch_frames < 8192 ? cfg->max_batch_frames : 8192;
    b->bit_words = (cfg->max_batch_frames + 63) / 64;

#define ALLOC(buf, count)                                                                              \
    do {                                                                                               \
        hipError_t _e = (buf).alloc(count);                                                            \
        if (_e != hipSuccess) {                                                                        \
            sdr_destroy(b);                                                                            \
            return fail(SDR_ERR_HIP, std::string("hipMalloc " #buf ": ") + hipGetErrorString(_e));     \
        }                                                                                              \
    } while (0)

    // SDR_NO_OVERLAP=1 runs every stage on the caller's stream (kernel-by-kernel profiling)
    const char *no_overlap = getenv("SDR_NO_OVERLAP");
    for (int s = 1; s < N_STAGES && !(no_overlap && no_overlap[0] == '1'); s++) {
        hipError_t e = hipStreamCreateWithFlags(&b->stream[s], hipStreamNonBlocking);
        if (e != hipSuccess) {
            sdr_destroy(b);
            return fail(SDR_ERR_HIP, "hipStreamCreate failed");
        }
        b->own_stream[s] = true;
    }
    // twiddles: go-dsp's table, re-laid-out per register pass
    {
        std::vector<double> wre, wim;
        fft64::radix2_factors(N, wre, wim);
        const size_t ntw = (size_t)sdr::twiddle_count(b->logn);
        std::vector<fft64::cplx> h(ntw);
        sdr::build_twiddles(b->logn, wre.data(), wim.data(), h.data());
        ALLOC(b->tw, h.size());
        hipError_t e = hipMemcpy(b->tw.p, h.data(), h.size() * sizeof(fft64::cplx), hipMemcpyHostToDevice);
        // tables of the certified dB shortcut (they fold log2 N in)
        std::vector<unsigned char> tab(gomath::kDbTabBytes);
        gomath::build_db_tables(b->logn, tab.data());
        ALLOC(b->db_tab, tab.size());
        if (e == hipSuccess)
            e = hipMemcpy(b->db_tab.p, tab.data(), tab.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            sdr_destroy(b);
            return fail(SDR_ERR_HIP, "twiddle / table upload failed");
        }
    }
    b->set.resize(RING);
    for (int r = 0; r < RING; r++) {
        const hipError_t se = alloc_set(b, b->set[r]);
        if (se != hipSuccess) {
            sdr_destroy(b);
            return fail(SDR_ERR_HIP, std::string("allocating a batch buffer set: ") + hipGetErrorString(se));
        }
    }
    ALLOC(b->band_state, B);
    ALLOC(b->drops, 1);
    ALLOC(b->cursors, GRAPH_PHASES * RING);
    ALLOC(b->spectrum_row, (size_t)N);
    ALLOC(b->tap_bins, B * std::max<size_t>(L, 1));
    {
        std::vector<int32_t> free_bins(B * std::max<size_t>(L, 1), -1);
        hipError_t he = hipMemcpy(b->tap_bins.p, free_bins.data(), free_bins.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (he != hipSuccess) {
            sdr_destroy(b);
            return fail(SDR_ERR_HIP, "tap table upload failed");
        }
    }
    ALLOC(b->slots, B * L);
    ALLOC(b->morse, cw::kMorseTableSize);
    ALLOC(b->text, B * L * (size_t)b->text_cap);
    ALLOC(b->text_frames, B * L * (size_t)b->text_cap);
    ALLOC(b->carry[0], B * N);
    ALLOC(b->carry[1], B * N);
#undef ALLOC

    std::vector<uint16_t> h_morse(cw::kMorseTableSize);
    cw::build_morse_table(h_morse.data());
    hipError_t e = hipMemcpy(b->morse.p, h_morse.data(), sizeof(uint16_t) * cw::kMorseTableSize, hipMemcpyHostToDevice);
    b->h_band_state.assign(B, sdr::BandState{});
    for (auto &s : b->h_band_state)
        s.peak_threshold = cfg->peak_threshold;
    if (e == hipSuccess)
        e = hipMemcpy(b->band_state.p, b->h_band_state.data(), sizeof(sdr::BandState) * B, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        sdr_destroy(b);
        return fail(SDR_ERR_HIP, "state upload failed");
    }
    b->h_slots.assign(B * L, sdr::ListenerSlot{});
    b->n_slots.assign(B, 0);
    b->center_frequency.assign(B, 0);
    b->staged.assign(B, 0);
    b->staged_kind.assign(B, 0);
    *out = b;
    return SDR_OK;
}

int sdr_destroy(sdr_bank *b)
{
    if (!b)
        return SDR_OK;
    (void)hipSetDevice(b->device);
    for (int s = 0; s < N_STAGES; s++)
        (void)hipStreamSynchronize(b->stream[s]);
    resolve_profile(b);
    drop_graphs(b);
    for (auto &ph : b->phase_done)
        for (auto &ev : ph)
            if (ev)
                (void)hipEventDestroy(ev);
    b->tw.release();
    b->drops.release();
    b->cursors.release();
    b->db_tab.release();
    b->tap_bins.release();
    b->spectrum_row.release();
    for (auto &S : b->set)
        S.release();
    b->band_state.release();
    b->slots.release();
    b->morse.release();
    b->text.release();
    b->text_frames.release();
    b->carry[0].release();
    b->carry[1].release();
    for (int s = 0; s < N_STAGES; s++)
        if (b->own_stream[s] && b->stream[s])
            (void)hipStreamDestroy(b->stream[s]);
    if (b->copy_stream) {
        (void)hipStreamSynchronize(b->copy_stream);
        (void)hipStreamDestroy(b->copy_stream);
    }
    for (auto &st : b->stage) {
        if (st.h_f32)
            (void)hipHostFree(st.h_f32);
        if (st.h_raw)
            (void)hipHostFree(st.h_raw);
        st.d_f32.release();
        st.d_raw.release();
        if (st.uploaded)
            (void)hipEventDestroy(st.uploaded);
        if (st.consumed)
            (void)hipEventDestroy(st.consumed);
    }
    delete b;
    return SDR_OK;
}

int sdr_set_stream(sdr_bank *b, void *hip_stream)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    for (int s = 1; s < N_STAGES; s++)
        if (!b->own_stream[s])
            b->stream[s] = reinterpret_cast<hipStream_t>(hip_stream);  // SDR_NO_OVERLAP: one stream for all
    b->stream[S_FFT] = reinterpret_cast<hipStream_t>(hip_stream);
    return SDR_OK;
}

namespace {
// the staging set the caller is filling, with its buffers in place (allocated on first use)
// Copy into pinned staging memory.  One core moves about 12 GB/s into write-combined-free pinned pages; a large push
// (a whole batch at once) is split over a few threads so that the copy keeps up with the PCIe upload behind it.
static void staging_copy(void *dst, const void *src, size_t bytes)
{
    constexpr size_t kChunk = 8u << 20;
    const size_t parts = std::min<size_t>(bytes / kChunk, 6);
    if (parts < 2) {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t step = ((bytes / parts) + 4095) & ~(size_t)4095;
    for (size_t i = 1; i < parts; i++) {
        const size_t off = i * step, len = (i + 1 == parts) ? bytes - off : step;
        th.emplace_back([=] { memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, len); });
    }
    memcpy(dst, src, step);
    for (auto &t : th)
        t.join();
}

static int staging_ready(sdr_bank *b, bool raw)
{
    const sdr_config &c = b->cfg;
    sdr_bank::Staging &st = b->stage[b->stage_cur];
    const size_t per = 2 * (size_t)c.block_size;
    const size_t frames = (size_t)c.max_batch_frames * (size_t)c.n_bands;
    HIP_TRY(hipSetDevice(b->device));
    if (!b->copy_stream)
        HIP_TRY(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    if (!st.uploaded) {
        HIP_TRY(hipEventCreateWithFlags(&st.uploaded, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&st.consumed, hipEventDisableTiming));
    }
    if (!raw && !st.h_f32)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&st.h_f32), sizeof(float) * per * frames, hipHostMallocDefault));
    if (raw && !st.h_raw)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&st.h_raw), 2 * per * frames, hipHostMallocDefault));
    return SDR_OK;
}
}  // namespace

int sdr_push_iq(sdr_bank *b, int band, int sample_rate, const float *iq, size_t n_floats)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (!iq)
        return fail(SDR_ERR_BAD_ARG, "null iq");
    const sdr_config &c = b->cfg;
    if (sample_rate != c.sample_rate)  // rx/receiver.go:319-322
        return fail(SDR_ERR_BAD_RATE, "wrong incoming sample rate");
    const size_t per = 2 * (size_t)c.block_size;
    if (n_floats == 0 || n_floats % per != 0)  // rx/receiver.go:323-326
        return fail(SDR_ERR_BAD_SIZE, "wrong incoming block size");
    const size_t nf = n_floats / per;
    if ((size_t)b->staged[band] + nf > (size_t)c.max_batch_frames)  // rx/receiver.go:328-333
        return fail(SDR_ERR_WOULD_DROP, "IQ data skipped: staging queue full");
    if (b->staged[band] > 0 && b->staged_kind[band] != 1)
        return fail(SDR_ERR_STATE, "band already holds raw KiwiSDR frames in this batch");
    rc = staging_ready(b, false);
    if (rc)
        return rc;
    b->staged_kind[band] = 1;
    float *dst = b->stage[b->stage_cur].h_f32 + ((size_t)band * c.max_batch_frames + (size_t)b->staged[band]) * per;
    staging_copy(dst, iq, sizeof(float) * n_floats);  // copy on push: the caller may reuse its buffer (kiwi/client.go:203)
    b->staged[band] += (int)nf;
    return SDR_OK;
}

int sdr_push_kiwi_snd(sdr_bank *b, int band, int sample_rate, const uint8_t *payload, size_t n_bytes)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (!payload)
        return fail(SDR_ERR_BAD_ARG, "null payload");
    const sdr_config &c = b->cfg;
    if (sample_rate != c.sample_rate)
        return fail(SDR_ERR_BAD_RATE, "wrong incoming sample rate");
    constexpr size_t kHeader = 17;  // flags, sequence, S-meter, GPS (kiwi/client.go:285-290)
    const size_t per = 2 * (size_t)c.block_size * 2;  // bytes per frame: 2N int16
    if (n_bytes <= kHeader || (n_bytes - kHeader) % per != 0)  // kiwi/kiwi.go:96-98 panics on a partial block
        return fail(SDR_ERR_BAD_SIZE, "SND payload does not hold whole frames");
    const size_t nf = (n_bytes - kHeader) / per;
    if ((size_t)b->staged[band] + nf > (size_t)c.max_batch_frames)
        return fail(SDR_ERR_WOULD_DROP, "IQ data skipped: staging queue full");
    if (b->staged[band] > 0 && b->staged_kind[band] != 2)
        return fail(SDR_ERR_STATE, "band already holds float32 frames in this batch");
    rc = staging_ready(b, true);
    if (rc)
        return rc;
    b->staged_kind[band] = 2;
    staging_copy(b->stage[b->stage_cur].h_raw + ((size_t)band * c.max_batch_frames + (size_t)b->staged[band]) * per,
                 payload + kHeader, n_bytes - kHeader);
    b->staged[band] += (int)nf;
    return SDR_OK;
}

int sdr_staged_frames(sdr_bank *b, int band)
{
    if (check_band(b, band))
        return -1;
    return b->staged[band];
}

int sdr_process_staged(sdr_bank *b, int *n_frames_out)
{
    return sdr_process_staged_limit(b, b ? b->cfg.max_batch_frames : 0, n_frames_out);
}

int sdr_process_staged_limit(sdr_bank *b, int max_frames, int *n_frames_out)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    const sdr_config &c = b->cfg;
    int n = std::min(c.max_batch_frames, std::max(max_frames, 0));
    for (int v : b->staged)
        n = std::min(n, v);
    if (n_frames_out)
        *n_frames_out = n;
    if (n == 0)
        return SDR_OK;
    HIP_TRY(hipSetDevice(b->device));
    const size_t per = 2 * (size_t)c.block_size;
    const size_t F = (size_t)c.max_batch_frames;
    sdr_bank::Staging &st = b->stage[b->stage_cur];
    if (!st.d_f32.p) {
        hipError_t e = st.d_f32.alloc(per * F * (size_t)c.n_bands);
        if (e != hipSuccess)
            return fail(SDR_ERR_HIP, "hipMalloc iq staging failed");
    }
    // upload on the copy stream, once the FFT of this set's previous batch has read the device buffer
    HIP_TRY(hipStreamWaitEvent(b->copy_stream, st.consumed, 0));
    for (int band = 0; band < c.n_bands; band++) {
        float *dst = st.d_f32.p + (size_t)band * n * per;
        if (b->staged_kind[band] == 2) {
            // raw int16be payload: upload half the bytes, unpack in HBM (k_unpack.hip)
            if (!st.d_raw.p) {
                hipError_t e = st.d_raw.alloc(2 * per * F * (size_t)c.n_bands);
                if (e != hipSuccess)
                    return fail(SDR_ERR_HIP, "hipMalloc raw staging failed");
            }
            uint8_t *rdst = st.d_raw.p + (size_t)band * F * per * 2;
            HIP_TRY(hipMemcpyAsync(rdst, st.h_raw + (size_t)band * F * per * 2, 2 * per * (size_t)n, hipMemcpyHostToDevice,
                                   b->copy_stream));
            HIP_TRY(sdr::launch_unpack_be16(rdst, dst, per * (size_t)n, b->copy_stream));
        } else {
            HIP_TRY(hipMemcpyAsync(dst, st.h_f32 + (size_t)band * F * per, sizeof(float) * per * (size_t)n, hipMemcpyHostToDevice,
                                   b->copy_stream));
        }
    }
    HIP_TRY(hipEventRecord(st.uploaded, b->copy_stream));
    HIP_TRY(hipStreamWaitEvent(b->stream[S_FFT], st.uploaded, 0));
    int rc = process_device_impl(b, st.d_f32.p, n, n);
    if (rc)
        return rc;
    HIP_TRY(hipEventRecord(st.consumed, b->stream[S_FFT]));  // (behind the FFT launch: the only reader of d_f32)
    // the caller goes on filling the next set; what this batch did not take moves to its front
    const int next = (b->stage_cur + 1) % sdr_bank::STAGE_RING;
    const int prev = b->stage_cur;
    b->stage_cur = next;
    bool any_left = false;
    for (int band = 0; band < c.n_bands; band++)
        any_left = any_left || b->staged[band] > n;
    if (any_left) {
        bool raw = false, f32 = false;
        for (int band = 0; band < c.n_bands; band++)
            if (b->staged[band] > n)
                (b->staged_kind[band] == 2 ? raw : f32) = true;
        if (f32 && (rc = staging_ready(b, false)))
            return rc;
        if (raw && (rc = staging_ready(b, true)))
            return rc;
    }
    // the pinned buffers of the next set are free once ITS last upload has completed (two batches ago: a formality)
    if (b->stage[next].uploaded)
        HIP_TRY(hipEventSynchronize(b->stage[next].uploaded));
    for (int band = 0; band < c.n_bands; band++) {
        const int left = b->staged[band] - n;
        if (left > 0) {
            if (b->staged_kind[band] == 2)
                memcpy(b->stage[next].h_raw + (size_t)band * F * per * 2, b->stage[prev].h_raw + ((size_t)band * F + (size_t)n) * per * 2,
                       per * 2 * (size_t)left);
            else
                memcpy(b->stage[next].h_f32 + (size_t)band * F * per, b->stage[prev].h_f32 + ((size_t)band * F + (size_t)n) * per,
                       sizeof(float) * per * (size_t)left);
        } else {
            b->staged_kind[band] = 0;
        }
        b->staged[band] = std::max(left, 0);
    }
    return SDR_OK;
}

int sdr_process_device(sdr_bank *b, const float *iq_dev, int n_frames)
{
    if (!b || !iq_dev)
        return fail(SDR_ERR_BAD_ARG, "null argument");
    if (reinterpret_cast<uintptr_t>(iq_dev) & 15)
        return fail(SDR_ERR_BAD_ARG, "iq_dev must be 16-byte aligned (frames are copied to LDS 16 bytes per lane)");
    return process_device_impl(b, iq_dev, n_frames, n_frames);
}

int sdr_sync(sdr_bank *b)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    return sync_bank(b);
}

// Control calls touch state owned by a pipeline stage: drain the pipeline first (they are rare, and
// the reference also applies them between frames only, rx/receiver.go:166-172).
int sdr_attach(sdr_bank *b, int band, int bin, int *listener_id)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    const sdr_config &c = b->cfg;
    if (bin < 0 || bin >= c.block_size)
        return fail(SDR_ERR_BAD_ARG, "bin out of range");
    // reuse a released slot first, else grow (ListenerPool.BindNext, rx/listener.go:214-229)
    int lid = -1;
    for (int i = 0; i < b->n_slots[band]; i++)
        if (!b->h_slots[(size_t)band * c.max_listeners + i].active) {
            lid = i;
            break;
        }
    if (lid < 0) {
        if (b->n_slots[band] >= c.max_listeners)
            return fail(SDR_ERR_NO_SLOT, "listener pool exhausted");
        lid = b->n_slots[band]++;
    }
    sdr::ListenerSlot &s = b->h_slots[(size_t)band * c.max_listeners + lid];
    memset(&s, 0, sizeof s);
    s.active = 1;
    s.bin = bin;
    cw::debouncer_init(s.deb, c.signal_debounce);          // NewSpectralDemodulator, cw/spectral.go:25-33
    cw::decoder_init(s.dec, c.sample_rate, c.block_size);  // NewDecoder, cw/decode.go:131-147
    cw::decoder_reset(s.dec);                              // Listener.Attach -> demodulator.Reset, listener.go:88
    s.start_frame = s.tapped_from = (uint32_t)b->total_frames;  // listens from the next frame processed
    rc = sync_bank(b);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(b->slots.p + (size_t)band * c.max_listeners + lid, &s, sizeof s, hipMemcpyHostToDevice));
    const int32_t tap_bin = bin;
    HIP_TRY(hipMemcpy(b->tap_bins.p + (size_t)band * c.max_listeners + lid, &tap_bin, sizeof tap_bin, hipMemcpyHostToDevice));
    b->attach_gen++;
    if (listener_id)
        *listener_id = lid;
    return SDR_OK;
}

int sdr_detach(sdr_bank *b, int band, int lid)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    sdr::ListenerSlot &s = b->h_slots[(size_t)band * b->cfg.max_listeners + lid];
    if (!s.active)
        return fail(SDR_ERR_STATE, "listener not attached");
    s.active = 0;
    rc = sync_bank(b);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(&b->slots.p[(size_t)band * b->cfg.max_listeners + lid].active, &s.active, sizeof(int32_t),
                      hipMemcpyHostToDevice));
    const int32_t free_bin = -1;
    HIP_TRY(hipMemcpy(b->tap_bins.p + (size_t)band * b->cfg.max_listeners + lid, &free_bin, sizeof free_bin,
                      hipMemcpyHostToDevice));
    b->attach_gen++;
    return SDR_OK;
}

// ---- deferred listen half: strain-mode discovery without a host round trip per cumulation -----------------------
// rx/receiver.go:409-426 binds one listener per completed cumulation, to a peak of that cumulation, and the listener
// hears the very next frame.  Frame by frame that is a decision on the host every 100 frames.  Here the spectral half
// of a long batch runs first (FFT .. FindPeaks of EVERY cumulation in it), the host reads those peaks (sdr_poll_peaks),
// makes the same decisions in the same order and binds each listener with the frame it starts at (sdr_attach_at); then
// the listen half runs over the retained spectra (sdr_process_listen).  Listeners are independent of each other, so a
// listener that starts in the middle of the batch produces exactly what it would have produced attached there live.
int sdr_defer_listen(sdr_bank *b, int on)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    if (on && !b->results_on)
        return fail(SDR_ERR_STATE, "deferred listening needs bulk delivery (sdr_enable_results)");
    if (b->listen_pending)
        return fail(SDR_ERR_STATE, "a batch waits for its listen half (sdr_process_listen)");
    b->defer_listen = on != 0;
    return SDR_OK;
}

int sdr_listen_pending(sdr_bank *b) { return b && b->listen_pending ? 1 : 0; }

int sdr_process_listen(sdr_bank *b)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    if (b->failed)
        return fail(SDR_ERR_STATE, "an earlier process call failed half way; destroy the bank");
    if (!b->listen_pending)
        return fail(SDR_ERR_STATE, "no batch waits for its listen half");
    const int rc = process_device_body(b, nullptr, b->pend.frames, b->pend.frames, -1, -1, PART_LISTEN);
    if (rc == SDR_ERR_HIP)
        b->failed = true;
    return rc;
}

int sdr_attach_at(sdr_bank *b, int band, int bin, int64_t start_frame, int *listener_id)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    const sdr_config &c = b->cfg;
    if (bin < 0 || bin >= c.block_size)
        return fail(SDR_ERR_BAD_ARG, "bin out of range");
    const int64_t lo = b->listen_pending ? b->pend.first_frame : b->total_frames;
    if (start_frame < lo || start_frame > b->total_frames)
        return fail(SDR_ERR_BAD_ARG, "start_frame must lie in the batch that waits for its listen half (or be the next frame)");
    if (b->graph_ready)
        return fail(SDR_ERR_STATE, "a graph is captured (sdr_graph_release first)");
    HIP_TRY(hipSetDevice(b->device));
    int lid = -1;
    for (int i = 0; i < b->n_slots[band]; i++)
        if (!b->h_slots[(size_t)band * c.max_listeners + i].active) {
            lid = i;
            break;
        }
    if (lid < 0) {
        if (b->n_slots[band] >= c.max_listeners)
            return fail(SDR_ERR_NO_SLOT, "listener pool exhausted");
        lid = b->n_slots[band]++;
    }
    sdr::ListenerSlot &s = b->h_slots[(size_t)band * c.max_listeners + lid];
    memset(&s, 0, sizeof s);
    s.active = 1;
    s.bin = bin;
    cw::debouncer_init(s.deb, c.signal_debounce);
    cw::decoder_init(s.dec, c.sample_rate, c.block_size);
    cw::decoder_reset(s.dec);
    s.start_frame = (uint32_t)start_frame;
    s.tapped_from = (uint32_t)b->total_frames;  // the FFT of every frame before that has run without this listener
    // (reaches the device with the next sdr_process_listen / process call: flush_late_attached)
    b->late_attached.push_back(band * c.max_listeners + lid);
    b->attach_gen++;
    if (listener_id)
        *listener_id = lid;
    return SDR_OK;
}

int sdr_listener_count(sdr_bank *b, int band)
{
    if (check_band(b, band))
        return -1;
    int n = 0;
    for (int i = 0; i < b->n_slots[band]; i++)
        n += b->h_slots[(size_t)band * b->cfg.max_listeners + i].active;
    return n;
}

int sdr_listener_stop(sdr_bank *b, int band, int lid)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    const size_t idx = (size_t)band * b->cfg.max_listeners + lid;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(sdr::launch_listener_stop(b->slots.p + idx, b->morse.p, b->text.p + idx * b->text_cap,
                                      b->text_frames.p + idx * b->text_cap, b->text_cap,
                                      (uint32_t)std::max<int64_t>(b->total_frames - 1, 0), b->drops.p, b->stream[S_LISTEN]));
    return SDR_OK;
}

int sdr_set_peak_threshold(sdr_bank *b, int band, float threshold)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    b->h_band_state[band].peak_threshold = threshold;
    rc = sync_bank(b);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(&b->band_state.p[band].peak_threshold, &b->h_band_state[band].peak_threshold, sizeof(float),
                      hipMemcpyHostToDevice));
    return SDR_OK;
}

int sdr_set_edge_width(sdr_bank *b, int edge_width)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    if (edge_width < 0 || b->cfg.block_size - 2 * edge_width < 10)
        return fail(SDR_ERR_BAD_ARG, "edge_width leaves fewer than 10 bins");
    b->edge_width = edge_width;  // a launch parameter: picked up by the next batch
    return SDR_OK;
}

int sdr_set_signal_debounce(sdr_bank *b, int band, int debounce)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (b->n_slots[band] == 0)
        return SDR_OK;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(sdr::launch_set_debounce(b->slots.p + (size_t)band * b->cfg.max_listeners, b->n_slots[band], debounce,
                                     b->stream[S_LISTEN]));
    return SDR_OK;
}

int sdr_set_center_frequency(sdr_bank *b, int band, int64_t frequency)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    std::lock_guard<std::mutex> guard(b->res_mu);  // (sdr_poll's thread reads the batches' snapshots under it)
    b->center_frequency[band] = frequency;
    return SDR_OK;
}

int sdr_set_find_peaks(sdr_bank *b, int on)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    b->find_peaks = on ? 1 : 0;
    return SDR_OK;
}

int sdr_last_batch_frames(sdr_bank *b) { return b ? b->last_frames : -1; }
int64_t sdr_total_frames(sdr_bank *b) { return b ? b->total_frames : -1; }
int sdr_last_batch_chunks(sdr_bank *b) { return b ? b->last_chunks : -1; }

int sdr_read_peaks(sdr_bank *b, int band, int chunk, sdr_peak *out, int max, int *n_out, int *frame_in_batch)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (chunk < 0 || chunk >= b->last_chunks)
        return fail(SDR_ERR_BAD_ARG, "chunk out of range");
    rc = sync_bank(b);
    if (rc)
        return rc;
    const sdr_config &c = b->cfg;
    const BatchSet &S = b->set[b->last_set];
    if (frame_in_batch)
        *frame_in_batch = (SDR_CUMULATION_SIZE - b->last_count0) + chunk * SDR_CUMULATION_SIZE - 1;
    if (!b->find_peaks) {
        if (n_out)
            *n_out = 0;
        return SDR_OK;
    }
    int count = 0;
    HIP_TRY(hipMemcpy(&count, S.peak_counts.p + (size_t)band * b->max_chunks + chunk, sizeof(int), hipMemcpyDeviceToHost));
    if (n_out)
        *n_out = count;
    const int n = std::min(std::min(count, c.max_peaks), max);
    if (n <= 0 || !out)
        return SDR_OK;
    std::vector<sdr::DevPeak> dp((size_t)n);
    HIP_TRY(hipMemcpy(dp.data(), S.dev_peaks.p + ((size_t)band * b->max_chunks + chunk) * c.max_peaks,
                      sizeof(sdr::DevPeak) * (size_t)n, hipMemcpyDeviceToHost));
    host::FrequencyMapping fm(c.sample_rate, c.block_size, b->center_frequency[band]);
    for (int i = 0; i < n; i++) {
        const sdr::DevPeak &p = dp[i];
        sdr_peak &o = out[i];
        o.from = p.from;
        o.to = p.to;
        o.signal_bin = p.signal_bin;
        o.signal_value = p.signal_value;
        o.from_frequency = fm.BinToFrequency(p.from, host::BinFrom);
        o.to_frequency = fm.BinToFrequency(p.to, host::BinTo);
        const double corr = host::PeakCenterCorrection(p.signal_bin, c.block_size, p.y1, p.y2, p.y3);
        o.signal_frequency = fm.BinToFrequency(p.signal_bin, corr);
    }
    return SDR_OK;
}

int sdr_read_cumulation(sdr_bank *b, int band, int chunk, float *out)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (chunk < 0 || chunk >= b->last_chunks || !out)
        return fail(SDR_ERR_BAD_ARG, "chunk out of range");
    rc = sync_bank(b);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(out, b->set[b->last_set].cum_out.p + ((size_t)band * b->max_chunks + chunk) * b->cfg.block_size,
                      sizeof(float) * (size_t)b->cfg.block_size, hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_text(sdr_bank *b, int band, int lid, char *out, int max_bytes, int *n_bytes)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    rc = sync_bank(b);
    if (rc)
        return rc;
    const size_t idx = (size_t)band * b->cfg.max_listeners + lid;
    sdr::ListenerSlot s;
    HIP_TRY(hipMemcpy(&s, b->slots.p + idx, sizeof s, hipMemcpyDeviceToHost));
    std::vector<uint32_t> runes(s.text_count);
    if (s.text_count)
        HIP_TRY(hipMemcpy(runes.data(), b->text.p + idx * b->text_cap, sizeof(uint32_t) * s.text_count,
                          hipMemcpyDeviceToHost));
    int n = 0;
    uint32_t consumed = 0;
    for (; consumed < s.text_count; consumed++) {
        char tmp[4];
        const size_t k = utf8_encode(runes[consumed], tmp);
        if (n + (int)k > max_bytes)
            break;
        if (out)
            memcpy(out + n, tmp, k);
        n += (int)k;
    }
    if (n_bytes)
        *n_bytes = n;
    // drop what was handed out, keep the rest at the front of the buffer
    const uint32_t left = s.text_count - consumed;
    if (left && consumed)
        HIP_TRY(hipMemcpy(b->text.p + idx * b->text_cap, runes.data() + consumed, sizeof(uint32_t) * left,
                          hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&b->slots.p[idx].text_count, &left, sizeof(uint32_t), hipMemcpyHostToDevice));
    return SDR_OK;
}

int sdr_read_edges(sdr_bank *b, int band, int lid, sdr_edge *out, int max, int *n_out)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    rc = sync_bank(b);
    if (rc)
        return rc;
    const size_t idx = (size_t)band * b->cfg.max_listeners + lid;
    const BatchSet &S = b->set[b->last_set];
    uint32_t count = 0;
    HIP_TRY(hipMemcpy(&count, S.edge_counts.p + idx, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_out)
        *n_out = (int)count;
    const int n = std::min(std::min((int)count, b->edge_cap), max);
    if (n > 0 && out)
        HIP_TRY(hipMemcpy(out, S.edges.p + idx * b->edge_cap, sizeof(sdr_edge) * (size_t)n, hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_keying_bits(sdr_bank *b, int band, int lid, uint64_t *out, int max_words)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    rc = sync_bank(b);
    if (rc)
        return rc;
    const size_t idx = (size_t)band * b->cfg.max_listeners + lid;
    const int words = std::min((b->last_frames + 63) / 64, max_words);
    if (words > 0 && out)
        HIP_TRY(hipMemcpy(out, b->set[b->last_set].bits.p + idx * b->bit_words, sizeof(uint64_t) * (size_t)words,
                          hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_frame_records(sdr_bank *b, int band, sdr_frame_rec *out, int max)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    rc = sync_bank(b);
    if (rc)
        return rc;
    const int n = std::min(b->last_frames, max);
    if (n > 0 && out)
        HIP_TRY(hipMemcpy(out, b->set[b->last_set].recs.p + (size_t)band * b->cfg.max_batch_frames,
                          sizeof(sdr_frame_rec) * (size_t)n, hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_trace(sdr_bank *b, int band, int lid, float *values, uint8_t *raw, uint8_t *debounced, int max)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    if (!b->cfg.trace)
        return fail(SDR_ERR_STATE, "bank was created without trace");
    rc = sync_bank(b);
    if (rc)
        return rc;
    const sdr_config &c = b->cfg;
    const BatchSet &S = b->set[b->last_set];
    const int n = std::min(b->last_frames, max);
    if (n <= 0)
        return SDR_OK;
    const size_t L = (size_t)c.max_listeners;
    const size_t base = (size_t)band * c.max_batch_frames * L + lid;
    // strided gather: [frame][listener] -> per-listener row
    if (values)
        HIP_TRY(hipMemcpy2D(values, sizeof(float), S.tr_values.p + base, sizeof(float) * L, sizeof(float), (size_t)n,
                            hipMemcpyDeviceToHost));
    if (raw)
        HIP_TRY(hipMemcpy2D(raw, 1, S.tr_raw.p + base, L, 1, (size_t)n, hipMemcpyDeviceToHost));
    if (debounced)
        HIP_TRY(hipMemcpy2D(debounced, 1, S.tr_deb.p + base, L, 1, (size_t)n, hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_spectrum(sdr_bank *b, int band, int frame, float *spectrum, float *psd)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (frame < 0 || frame >= b->last_frames)
        return fail(SDR_ERR_BAD_ARG, "frame out of range");
    rc = sync_bank(b);
    if (rc)
        return rc;
    const size_t N = (size_t)b->cfg.block_size;
    const size_t off = ((size_t)band * b->cfg.max_batch_frames + frame) * N;
    const BatchSet &S = b->set[b->last_set];
    if (spectrum) {
        // the pipeline keeps psd only; the dB projection of the row is made on demand (dsp/fft.go:79-81)
        HIP_TRY(sdr::launch_spectrum_row(S.psd.p + off, b->spectrum_row.p, (int)N, b->stream[S_FFT]));
        HIP_TRY(hipStreamSynchronize(b->stream[S_FFT]));
        HIP_TRY(hipMemcpy(spectrum, b->spectrum_row.p, sizeof(float) * N, hipMemcpyDeviceToHost));
    }
    if (psd)
        HIP_TRY(hipMemcpy(psd, S.psd.p + off, sizeof(float) * N, hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_decoder_state(sdr_bank *b, int band, int lid, double *out12)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    if (!out12)
        return fail(SDR_ERR_BAD_ARG, "null out");
    rc = sync_bank(b);
    if (rc)
        return rc;
    sdr::ListenerSlot s;
    HIP_TRY(hipMemcpy(&s, b->slots.p + (size_t)band * b->cfg.max_listeners + lid, sizeof s, hipMemcpyDeviceToHost));
    const cw::DecoderState &d = s.dec;
    out12[0] = d.ticks;
    out12[1] = d.onStart;
    out12[2] = d.offStart;
    out12[3] = d.wpm;
    out12[4] = d.onThreshold.low;
    out12[5] = d.onThreshold.high;
    out12[6] = d.onThreshold.last;
    out12[7] = d.onThreshold.threshold;
    out12[8] = d.offThreshold.low;
    out12[9] = d.offThreshold.high;
    out12[10] = d.offThreshold.last;
    out12[11] = d.offThreshold.threshold;
    return SDR_OK;
}

// ---- graph mode ------------------------------------------------------------------------------------------------
// RING consecutive batches recorded once and replayed as LINEAR, KERNEL-ONLY GRAPHS, one per stream of the bank: the FFT
// graph (cursors + FFT of the six batches), the noise graph (window means, statistics), the peaks stream's two graphs
// (thresholds; then cumulate, find peaks, pack) and the listen graph (gather, decode, pack).  Inside a replay the streams
// are ordered by ordinary events around whole graphs (FFT -> noise -> thresholds -> {peaks, listen}); replay r+1's FFT
// graph runs while replay r's noise graph and replay r-1's peaks / listen graphs do - the same kernels side by side as
// in the eager pipeline, only taken from different replays.  That needs buffer sets per replay in flight: GRAPH_PHASES
// groups of RING sets, each group with graphs of its own (the buffers are baked into the kernel nodes), used round robin.
// The host enqueues per six batches: 5 graph launches, 3-6 event waits, 5 + 12 event records (sixty-odd commands eager).
// (Thresholds in the noise graph instead - one graph fewer - made the noise stream the longest: c3 143 GS/s against 152.)
// Why not one graph per replay, or events inside the graphs (both were built and measured, rounds 2 and 3):
//  - one graph with fork / join over four streams: the runtime maps its branches to queues of its own choosing (7-10 %
//    slower than eager; one process in three, three times slower) and a replay, being one stream operation, cannot
//    overlap the next one;
//  - per-stream graphs stitched by EXTERNAL event nodes at batch granularity (the eager path's events, in the same
//    places): correct, but every such node costs 70-120 us at replay (c3: 15-21 GS/s against 153 eager); and the
//    capture API for them is broken in this runtime (hipStreamWaitEvent(External) behind a kernel node throws
//    std::bad_alloc, several captures open at once corrupt memory: tools/experiments/probe_graph_ext.hip).
// What differs between batches (input pointer, frame numbering, cumulation phase, carry buffer) is read by the kernels
// from device-side cursors, written by the first node of the FFT graph; its kernel argument is the only thing a replay
// updates (hipGraphExecKernelNodeSetParams), so no host memory is read
// while a replay runs.
int sdr_graph_batches(sdr_bank *b) { return b ? RING : 0; }

int sdr_graph_release(sdr_bank *b)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    // results not polled yet move to the host-side queue: the eager ring takes over from here
    if (b->results_on && b->graph_ready)
        for (int64_t i = std::max(b->deliver_next, b->graph_base); i < b->batch_index; i++)
            if ((rc = park_results(b, b->set[set_index(b, i)])))
                return rc;
    drop_graphs(b);
    b->graph_ready = false;
    return SDR_OK;
}

int sdr_graph_capture(sdr_bank *b, int n_frames)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    if (n_frames <= 0 || n_frames > b->cfg.max_batch_frames)
        return fail(SDR_ERR_BAD_ARG, "n_frames out of range");
    if (b->failed)
        return fail(SDR_ERR_STATE, "an earlier process call failed half way; destroy the bank");
    if (b->batch_index % RING != 0)
        return fail(SDR_ERR_STATE, "capture needs the bank at a multiple of sdr_graph_batches() processed batches");
    if (b->listen_pending || b->defer_listen)
        return fail(SDR_ERR_STATE, "graph mode and the deferred listen half exclude each other (sdr_process_listen / sdr_defer_listen(0) first)");
    if (!b->own_stream[S_NOISE])
        return fail(SDR_ERR_STATE, "graph mode needs the bank's own side streams (SDR_NO_OVERLAP is set)");
    int rc = sdr_graph_release(b);  // (also drains the pipeline)
    if (rc)
        return rc;
    if (b->results_on)
        for (auto &S : b->set)
            if ((rc = park_results(b, S)))
                return rc;
    HIP_TRY(hipSetDevice(b->device));
    // the replays' buffer sets and events, once
    const size_t want = (size_t)RING + (size_t)GRAPH_PHASES * RING;
    if (b->set.size() < want) {
        const size_t have = b->set.size();
        b->set.resize(want);
        for (size_t i = have; i < want; i++) {
            const hipError_t se = alloc_set(b, b->set[i]);
            if (se != hipSuccess) {
                for (size_t j = have; j < want; j++)
                    b->set[j].release();
                b->set.resize(have);
                return fail(SDR_ERR_HIP, std::string("graph mode needs ") + std::to_string(GRAPH_PHASES * RING) +
                                             " more batch buffer sets: " + hipGetErrorString(se));
            }
        }
    }
    for (auto &ph : b->phase_done)
        for (auto &ev : ph)
            if (!ev)
                HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const bool was_profiling = b->profiling;
    b->profiling = false;
    // One stream records at a time (the legacy default stream cannot be captured: the bank must have been given a
    // stream, sdr_set_stream); the captures hold kernels only and share nothing.
    int max_slots = 0;
    for (int i = 0; i < b->cfg.n_bands; i++)
        max_slots = std::max(max_slots, b->n_slots[i]);
    hipError_t e = hipSuccess;
    rc = SDR_OK;
    for (int ph = 0; ph < GRAPH_PHASES && rc == SDR_OK; ph++)
        for (int st = 0; st < N_GRAPHS && rc == SDR_OK; st++) {
            hipStream_t cs = b->stream[st == G_THRESHOLDS ? S_PEAKS : st];
            e = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
            if (e != hipSuccess) {
                rc = fail(SDR_ERR_HIP, std::string("hipStreamBeginCapture (the bank's stream must not be the null stream): ") + hipGetErrorString(e));
                break;
            }
            for (int k = 0; k < RING && rc == SDR_OK; k++)
                rc = process_device_body(b, nullptr, n_frames, n_frames, ph * RING + k, st);
            e = hipStreamEndCapture(cs, &b->graph[ph][st]);
            if ((e != hipSuccess || !b->graph[ph][st]) && rc == SDR_OK)
                rc = fail(SDR_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        }
    b->profiling = was_profiling;
    if (rc != SDR_OK) {
        drop_graphs(b);
        return rc;
    }
    // the cursor node: the kernel node of an FFT graph that runs k_set_cursors
    for (int ph = 0; ph < GRAPH_PHASES; ph++) {
        size_t n_nodes = 0;
        hipGraph_t g = b->graph[ph][S_FFT];
        e = hipGraphGetNodes(g, nullptr, &n_nodes);
        std::vector<hipGraphNode_t> nodes(n_nodes);
        if (e == hipSuccess)
            e = hipGraphGetNodes(g, nodes.data(), &n_nodes);
        int found = 0;
        for (hipGraphNode_t nd : nodes) {
            hipGraphNodeType t;
            if (hipGraphNodeGetType(nd, &t) != hipSuccess || t != hipGraphNodeTypeKernel)
                continue;
            hipKernelNodeParams kp{};
            if (hipGraphKernelNodeGetParams(nd, &kp) != hipSuccess || kp.func != reinterpret_cast<void *>(&k_set_cursors))
                continue;
            b->graph_cursor_node[ph] = nd;
            found++;
        }
        if (e != hipSuccess || found != 1) {
            drop_graphs(b);
            return fail(SDR_ERR_HIP, "captured FFT graph does not hold exactly one cursor node");
        }
    }
    for (int ph = 0; ph < GRAPH_PHASES; ph++)
        for (int st = 0; st < N_GRAPHS; st++) {
            e = hipGraphInstantiate(&b->graph_exec[ph][st], b->graph[ph][st], nullptr, nullptr, 0);
            if (e != hipSuccess) {
                drop_graphs(b);
                return fail(SDR_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
            }
        }
    b->graph_ready = true;
    b->graph_base = b->batch_index;
    b->graph_replays = 0;
    b->graph_frames = n_frames;
    b->graph_slots = max_slots;
    b->graph_attach_gen = b->attach_gen;
    return SDR_OK;
}

int sdr_graph_launch(sdr_bank *b, const float *const *iq_dev)
{
    if (!b || !iq_dev)
        return fail(SDR_ERR_BAD_ARG, "null argument");
    if (!b->graph_ready)
        return fail(SDR_ERR_STATE, "no graph captured (sdr_graph_capture)");
    if (b->failed)
        return fail(SDR_ERR_STATE, "an earlier process call failed half way; destroy the bank");
    int max_slots = 0;
    for (int i = 0; i < b->cfg.n_bands; i++)
        max_slots = std::max(max_slots, b->n_slots[i]);
    if (max_slots != b->graph_slots || b->attach_gen != b->graph_attach_gen)
        return fail(SDR_ERR_STATE, "listeners were attached or detached since the capture: capture again");
    HIP_TRY(hipSetDevice(b->device));
    static const bool dbg = getenv("SDR_GRAPH_DEBUG") != nullptr;
    double tdbg[8] = {};
    int ndbg = 0;
    auto stamp = [&] {
        if (dbg && ndbg < 8)
            tdbg[ndbg++] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    stamp();
    const int n_frames = b->graph_frames;
    const int ph = (int)(b->graph_replays % GRAPH_PHASES);
    const int set0 = RING + ph * RING;
    CursorPack pack{};
    sdr::BatchCursor *cursor = pack.c;
    int count = b->cum_count, carry = b->carry_cur;
    int64_t total = b->total_frames;
    struct Meta {
        int count0, chunks;
        int64_t first_frame;
    } meta[RING];
    for (int k = 0; k < RING; k++) {
        if (!iq_dev[k] || (reinterpret_cast<uintptr_t>(iq_dev[k]) & 15))
            return fail(SDR_ERR_BAD_ARG, "every input pointer must be non-null and 16-byte aligned");
        cursor[k] = sdr::BatchCursor{};
        cursor[k].iq = iq_dev[k];
        cursor[k].frame_base = (uint32_t)total;
        cursor[k].count0 = count;
        cursor[k].carry_in = carry;
        meta[k] = {count, sdr::chunks_completed(count, n_frames), total};
        const int new_count = (count + n_frames) % SDR_CUMULATION_SIZE;
        if (new_count != 0)
            carry ^= 1;
        count = new_count;
        total += n_frames;
    }
    // results of the replay that used these sets GRAPH_PHASES replays ago and were never polled: to the host-side queue
    if (b->results_on)
        for (int k = 0; k < RING; k++) {
            const int prc = park_results(b, b->set[set0 + k]);
            if (prc)
                return prc;
        }
    stamp();
    {
        sdr::BatchCursor *dst = b->cursors.p + ph * RING;
        void *args[2] = {&dst, &pack};
        hipKernelNodeParams kp{};
        kp.func = reinterpret_cast<void *>(&k_set_cursors);
        kp.gridDim = dim3(1);
        kp.blockDim = dim3(64);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        const hipError_t e = hipGraphExecKernelNodeSetParams(b->graph_exec[ph][S_FFT], b->graph_cursor_node[ph], &kp);
        if (e != hipSuccess)
            return fail(SDR_ERR_HIP, std::string("hipGraphExecKernelNodeSetParams: ") + hipGetErrorString(e));
    }
    // FFT graph: once every reader of this phase's sets (the replay GRAPH_PHASES back) is done; then
    // noise behind FFT, peaks and listen behind noise (which ends with the thresholds).  From the first failure on the
    // replay is half enqueued and the bank refuses further work.
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) {
        if (e == hipSuccess)
            e = r;
    };
    stamp();
    hipEvent_t *done = b->phase_done[ph];
    if (b->graph_replays >= GRAPH_PHASES)
        for (int st : {S_NOISE, S_PEAKS, S_LISTEN})
            if (hipEventQuery(done[st]) != hipSuccess)  // (ask the host first, as the eager path does: a wait is a barrier packet in the FFT queue)
                step(hipStreamWaitEvent(b->stream[S_FFT], done[st], 0));
    step(hipGraphLaunch(b->graph_exec[ph][S_FFT], b->stream[S_FFT]));
    step(hipEventRecord(done[S_FFT], b->stream[S_FFT]));
    stamp();
    step(hipStreamWaitEvent(b->stream[S_NOISE], done[S_FFT], 0));
    step(hipGraphLaunch(b->graph_exec[ph][S_NOISE], b->stream[S_NOISE]));
    step(hipEventRecord(done[S_NOISE], b->stream[S_NOISE]));
    stamp();
    // the peaks stream: thresholds (batch order, behind the noise statistics), then cumulate / find peaks / pack, which
    // need this replay's spectra (implied by the noise graph) and thresholds (same stream)
    step(hipStreamWaitEvent(b->stream[S_PEAKS], done[S_NOISE], 0));
    step(hipGraphLaunch(b->graph_exec[ph][G_THRESHOLDS], b->stream[S_PEAKS]));
    step(hipEventRecord(done[G_THRESHOLDS], b->stream[S_PEAKS]));
    step(hipGraphLaunch(b->graph_exec[ph][S_PEAKS], b->stream[S_PEAKS]));
    step(hipEventRecord(done[S_PEAKS], b->stream[S_PEAKS]));
    step(hipStreamWaitEvent(b->stream[S_LISTEN], done[G_THRESHOLDS], 0));
    step(hipGraphLaunch(b->graph_exec[ph][S_LISTEN], b->stream[S_LISTEN]));
    step(hipEventRecord(done[S_LISTEN], b->stream[S_LISTEN]));
    if (b->results_on)
        for (int k = 0; k < RING; k++) {
            step(hipEventRecord(b->set[set0 + k].res_peaks, b->stream[S_PEAKS]));
            step(hipEventRecord(b->set[set0 + k].res_listen, b->stream[S_LISTEN]));
        }
    stamp();
    if (dbg)
        fprintf(stderr, "[graph launch %lld] park %.0f us, cursors %.0f us, fft %.0f us, noise %.0f us, peaks+listen %.0f us\n", (long long)b->graph_replays,
                tdbg[1] - tdbg[0], tdbg[2] - tdbg[1], tdbg[3] - tdbg[2], tdbg[4] - tdbg[3], tdbg[5] - tdbg[4]);
    if (e != hipSuccess) {
        b->failed = true;
        return fail(SDR_ERR_HIP, std::string("enqueueing a replay: ") + hipGetErrorString(e));
    }
    // the host's view of the carried state, batch by batch, as the eager path commits it (sdr_poll on another thread
    // finds a batch only now: its events are recorded)
    std::lock_guard<std::mutex> guard(b->res_mu);
    for (int k = 0; k < RING; k++) {
        BatchSet &S = b->set[set0 + k];
        if (b->results_on) {
            S.res_batch = b->batch_index;
            S.res_first_frame = meta[k].first_frame;
            S.res_frames = n_frames;
            S.res_chunks = meta[k].chunks;
            S.res_count0 = meta[k].count0;
            S.res_slots = max_slots;
            S.res_center = b->center_frequency;
        }
        b->last_set = set0 + k;
        b->last_frames = n_frames;
        b->last_chunks = meta[k].chunks;
        b->last_count0 = meta[k].count0;
        b->batch_index++;
    }
    b->batches_enqueued = b->batch_index;
    b->cum_count = count;
    b->carry_cur = carry;
    b->total_frames = total;
    b->graph_replays++;
    return SDR_OK;
}

// ---- scope tap (scope/scope.go:14-37) -------------------------------------------------------------------------
int sdr_scope_active(sdr_bank *b) { return b ? (b->cfg.trace ? 1 : 0) : 0; }

int sdr_scope_read_spectral(sdr_bank *b, int band, int chunk, sdr_scope_spectral_frame *frame, double *values, int max_values)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (!b->cfg.trace)
        return fail(SDR_ERR_STATE, "scope inactive: the bank was created without trace");
    if (chunk < 0 || chunk >= b->last_chunks || !frame)
        return fail(SDR_ERR_BAD_ARG, "chunk out of range");
    rc = sync_bank(b);
    if (rc)
        return rc;
    const sdr_config &c = b->cfg;
    const int N = c.block_size;
    const BatchSet &S = b->set[b->last_set];
    const int end_frame = (SDR_CUMULATION_SIZE - b->last_count0) + chunk * SDR_CUMULATION_SIZE - 1;
    sdr_frame_rec rec;
    HIP_TRY(hipMemcpy(&rec, S.recs.p + (size_t)band * c.max_batch_frames + end_frame, sizeof rec, hipMemcpyDeviceToHost));
    frame->frame = (b->total_frames - b->last_frames) + end_frame;
    frame->from_frequency = 0.0;
    frame->to_frequency = 1.0;
    frame->threshold = (double)rec.peak_thr;
    frame->n_values = N;
    frame->reserved = 0;
    // DecodeMode: the listener; StrainMode: the pool's first listener (rx/receiver.go:430-441); -1 without one
    frame->signal_bin = -1.0;
    for (int i = 0; i < b->n_slots[band]; i++) {
        const sdr::ListenerSlot &sl = b->h_slots[(size_t)band * c.max_listeners + i];
        if (sl.active) {
            frame->signal_bin = (double)sl.bin;
            break;
        }
    }
    if (values) {
        std::vector<float> cum((size_t)N);
        HIP_TRY(hipMemcpy(cum.data(), S.cum_out.p + ((size_t)band * b->max_chunks + chunk) * N, sizeof(float) * (size_t)N,
                          hipMemcpyDeviceToHost));
        const double scale = 1.0 / (double)SDR_CUMULATION_SIZE;  // scaledValuesForScope(cumulation, 1.0/float64(cumulationSize))
        for (int i = 0; i < std::min(N, max_values); i++)
            values[i] = (double)cum[i] * scale;
    }
    return SDR_OK;
}

int sdr_scope_read_demod(sdr_bank *b, int band, int lid, sdr_scope_time_frame *out, int max, int *n_out)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    if (!b->cfg.trace)
        return fail(SDR_ERR_STATE, "scope inactive: the bank was created without trace");
    const int n = std::min(b->last_frames, max);
    if (n_out)
        *n_out = b->last_frames;
    if (n <= 0 || !out)
        return SDR_OK;
    std::vector<float> v((size_t)n);
    std::vector<uint8_t> raw((size_t)n), deb((size_t)n);
    rc = sdr_read_trace(b, band, lid, v.data(), raw.data(), deb.data(), n);  // synchronises
    if (rc)
        return rc;
    std::vector<sdr_frame_rec> recs((size_t)n);
    HIP_TRY(hipMemcpy(recs.data(), b->set[b->last_set].recs.p + (size_t)band * b->cfg.max_batch_frames,
                      sizeof(sdr_frame_rec) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        out[i].threshold = (double)recs[i].listen_thr;
        out[i].value = (double)v[i];
        out[i].state = raw[i] ? 100.0 : -1.0;      // cw/spectral.go:61-64
        out[i].debounced = deb[i] ? 80.0 : -1.0;   // cw/spectral.go:65-68
    }
    return SDR_OK;
}

// cw.Decoder's scope streams (cw/decode.go:228-243 and :433-491): per tick, what scopeDecode / scopeSignalTiming /
// scopeGapTiming / scopeSignal show - the current run's duration, both adaptive thresholds with their low and high, the
// state.  The decoders run on the device in closed form between edges and keep none of this per tick; it is replayed
// here, on the host, tick by tick with the literal Tick (cw_decoder.h decoder_tick, the function the device's run-length
// form is checked against), from the decoder's state before the batch and the batch's debounced keying.
int sdr_scope_read_decode(sdr_bank *b, int band, int lid, sdr_scope_decode_frame *out, int max, int *n_out)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    if (!b->cfg.trace)
        return fail(SDR_ERR_STATE, "scope inactive: the bank was created without trace");
    if (n_out)
        *n_out = 0;
    const int frames = b->last_frames;
    if (frames <= 0)
        return SDR_OK;
    std::vector<uint8_t> deb((size_t)frames);
    rc = sdr_read_trace(b, band, lid, nullptr, nullptr, deb.data(), frames);  // synchronises
    if (rc)
        return rc;
    const BatchSet &S = b->set[b->last_set];
    sdr::ListenerSlot slot;
    HIP_TRY(hipMemcpy(&slot, S.slots_before.p + (size_t)band * b->cfg.max_listeners + lid, sizeof slot, hipMemcpyDeviceToHost));
    if (!slot.active)
        return fail(SDR_ERR_STATE, "listener was not attached during the last batch");
    // a listener bound inside the batch (sdr_attach_at) ticks from its first frame on
    const int64_t first = b->total_frames - frames;
    const int skip = (int)std::max<int64_t>(0, (int64_t)(int32_t)(slot.start_frame - (uint32_t)first));
    std::vector<uint16_t> table(cw::kMorseTableSize);
    cw::build_morse_table(table.data());
    struct NullSink {
        void put(uint32_t) {}
    } sink;
    cw::DecoderState d = slot.dec;
    int n = 0;
    for (int f = skip; f < frames; f++) {
        const bool state = deb[(size_t)f] != 0;
        cw::decoder_tick(d, state, table.data(), sink);
        if (out && n < max) {
            sdr_scope_decode_frame &o = out[n];
            o.frame = first + f;
            o.duration = state ? d.ticks - d.onStart : d.ticks - d.offStart;  // currentDuration :222-227
            o.state = state ? 1.0 : 0.0;
            o.on_threshold = d.onThreshold.threshold;
            o.on_threshold_low = d.onThreshold.low;
            o.on_threshold_high = d.onThreshold.high;
            o.off_threshold = d.offThreshold.threshold;
            o.off_threshold_low = d.offThreshold.low;
            o.off_threshold_high = d.offThreshold.high;
        }
        n++;
    }
    if (n_out)
        *n_out = n;
    return SDR_OK;
}

// ---- bulk delivery -------------------------------------------------------------------------------------------
int sdr_enable_results(sdr_bank *b, int on)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    if (on && !b->set[0].res_host) {
        HIP_TRY(hipSetDevice(b->device));
        b->res_layout = make_results_layout(b);
        for (auto &S : b->set) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.res_host), b->res_layout.bytes, hipHostMallocDefault));
            memset(S.res_host, 0, b->res_layout.bytes);
            HIP_TRY(hipEventCreateWithFlags(&S.res_listen, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&S.res_peaks, hipEventDisableTiming));
        }
    }
    if (!on) {
        // undelivered batches are discarded with the mode
        for (auto &S : b->set)
            S.res_batch = -1;
        b->parked.clear();
    }
    std::lock_guard<std::mutex> guard(b->res_mu);
    b->results_on = on != 0;
    b->deliver_next = b->batch_index;
    b->batches_enqueued = b->batch_index;
    return SDR_OK;
}

int sdr_results_pending(sdr_bank *b)
{
    if (!b || !b->results_on)
        return 0;
    std::lock_guard<std::mutex> guard(b->res_mu);
    return (int)(b->batches_enqueued - b->deliver_next);
}

namespace {
struct BatchMeta {
    int64_t batch, first_frame;
    int frames, chunks, count0, slots;
    const std::vector<int64_t> *center;  // centre frequencies as of the batch, not as of the poll
};

// block (pinned set or parked copy) -> the caller's buffers
static int deliver_block(sdr_bank *b, const unsigned char *blk, const BatchMeta &m, sdr_results *r);
// res_mu held: the oldest undelivered batch sits in the parked queue
static int sdr_poll_parked(sdr_bank *b, sdr_results *r)
{
    if (b->parked.empty() || b->parked.front().batch != b->deliver_next)
        return fail(SDR_ERR_STATE, "results of the next batch are not where they should be");
    const sdr_bank::Parked &p = b->parked.front();
    const BatchMeta m{p.batch, p.first_frame, p.frames, p.chunks, p.count0, p.slots, &p.center};
    const int rc = deliver_block(b, p.block.get(), m, r);
    if (rc == SDR_OK) {
        b->parked.pop_front();
        b->deliver_next++;
    }
    return rc;
}

static int deliver_block(sdr_bank *b, const unsigned char *blk, const BatchMeta &m, sdr_results *r)
{
    const sdr_config &c = b->cfg;
    const sdr::ResultsLayout &lay = b->res_layout;
    const int B = c.n_bands, L = c.max_listeners;
    const int *peak_counts = reinterpret_cast<const int *>(blk + lay.off_peak_counts);
    const sdr::DevPeak *peaks = reinterpret_cast<const sdr::DevPeak *>(blk + lay.off_peaks);
    const uint32_t *edge_counts = reinterpret_cast<const uint32_t *>(blk + lay.off_edge_counts);
    const sdr_edge *edges = reinterpret_cast<const sdr_edge *>(blk + lay.off_edges);
    const uint32_t *text_counts = reinterpret_cast<const uint32_t *>(blk + lay.off_text_counts);
    const uint32_t *text = reinterpret_cast<const uint32_t *>(blk + lay.off_text);
    const uint32_t *text_frames = reinterpret_cast<const uint32_t *>(blk + lay.off_text_frames);
    // what is needed
    int64_t need_peaks = 0, need_edges = 0, need_runes = 0;
    int need_listeners = 0;
    for (int band = 0; band < B; band++) {
        for (int ch = 0; ch < m.chunks; ch++)
            need_peaks += peak_counts[2 * ((size_t)band * lay.max_chunks + ch)];
        for (int l = 0; l < m.slots; l++) {
            const size_t idx = (size_t)band * L + l;
            need_edges += edge_counts[idx];
            need_runes += text_counts[idx];
            need_listeners += (edge_counts[idx] || text_counts[idx]) ? 1 : 0;
        }
    }
    const int need_chunks = m.chunks * B;
    const bool fits = need_chunks <= r->chunks_cap && need_peaks <= r->peaks_cap && need_listeners <= r->listeners_cap &&
                      need_edges <= r->edges_cap && need_runes <= r->runes_cap &&
                      (need_chunks == 0 || r->chunks) && (need_peaks == 0 || r->peaks) &&
                      (need_listeners == 0 || r->listeners) && (need_edges == 0 || r->edges) &&
                      (need_runes == 0 || (r->runes && r->rune_frames));
    r->n_chunks = need_chunks;
    r->n_peaks = (int32_t)need_peaks;
    r->n_listeners = need_listeners;
    r->n_edges = (int32_t)need_edges;
    r->n_runes = (int32_t)need_runes;
    r->n_frames = m.frames;
    r->batch_index = m.batch;
    r->first_frame = m.first_frame;
    const sdr::DropCounters *dc = reinterpret_cast<const sdr::DropCounters *>(blk + lay.off_drops);
    r->runes_dropped = dc->runes;
    r->edges_dropped = dc->edges;
    if (!fits)
        return fail(SDR_ERR_BAD_SIZE, "sdr_poll: a result buffer is too small (the n_* fields say what is needed)");
    int ci = 0, pi = 0, li = 0, ei = 0, ri = 0;
    for (int band = 0; band < B; band++) {
        host::FrequencyMapping fm(c.sample_rate, c.block_size, (*m.center)[band]);
        for (int ch = 0; ch < m.chunks; ch++) {
            const size_t cidx = (size_t)band * lay.max_chunks + ch;
            const int n = peak_counts[2 * cidx];
            sdr_chunk_result &cr = r->chunks[ci++];
            cr.band = band;
            cr.n_peaks = n;
            cr.frame = m.first_frame + (SDR_CUMULATION_SIZE - m.count0) + (int64_t)ch * SDR_CUMULATION_SIZE - 1;
            cr.first_peak = pi;
            cr.peaks_found = peak_counts[2 * cidx + 1];
            for (int i = 0; i < n; i++) {
                const sdr::DevPeak &p = peaks[cidx * lay.max_peaks + i];
                sdr_peak &o = r->peaks[pi++];
                o.from = p.from;
                o.to = p.to;
                o.signal_bin = p.signal_bin;
                o.signal_value = p.signal_value;
                o.from_frequency = fm.BinToFrequency(p.from, host::BinFrom);
                o.to_frequency = fm.BinToFrequency(p.to, host::BinTo);
                o.signal_frequency = fm.BinToFrequency(p.signal_bin, host::PeakCenterCorrection(p.signal_bin, c.block_size, p.y1, p.y2, p.y3));
            }
        }
    }
    for (int band = 0; band < B; band++)
        for (int l = 0; l < m.slots; l++) {
            const size_t idx = (size_t)band * L + l;
            const int ne = (int)edge_counts[idx], nr = (int)text_counts[idx];
            if (!ne && !nr)
                continue;
            sdr_listener_result &lr = r->listeners[li++];
            lr.band = band;
            lr.listener = l;
            lr.first_edge = ei;
            lr.n_edges = ne;
            lr.first_rune = ri;
            lr.n_runes = nr;
            if (ne)
                memcpy(r->edges + ei, edges + idx * lay.edge_cap, sizeof(sdr_edge) * (size_t)ne);
            if (nr) {
                memcpy(r->runes + ri, text + idx * lay.text_cap, sizeof(uint32_t) * (size_t)nr);
                memcpy(r->rune_frames + ri, text_frames + idx * lay.text_cap, sizeof(uint32_t) * (size_t)nr);
            }
            ei += ne;
            ri += nr;
        }
    return SDR_OK;
}
}  // namespace

int sdr_poll(sdr_bank *b, sdr_results *r, int wait)
{
    if (!b || !r)
        return fail(SDR_ERR_BAD_ARG, "null argument");
    if (r->struct_size != (int32_t)sizeof(sdr_results))
        return fail(SDR_ERR_BAD_ARG, "sdr_results.struct_size mismatch (ABI)");
    if (!b->results_on)
        return fail(SDR_ERR_STATE, "bulk delivery is off (sdr_enable_results)");
    std::unique_lock<std::mutex> guard(b->res_mu);
    // (a producer that needs a set back gives a polling consumer the chance to take its batch: park_results)
    struct Polling {
        sdr_bank *b;
        bool waiting;
        Polling(sdr_bank *b_, bool w) : b(b_), waiting(w) { b->pollers_waiting += waiting ? 1 : 0; }
        ~Polling()  // (the mutex is held again whenever sdr_poll returns)
        {
            b->pollers_waiting -= waiting ? 1 : 0;
            b->last_poll = std::chrono::steady_clock::now();
            b->res_cv.notify_all();
        }
    } polling(b, wait != 0);
    if (b->deliver_next >= b->batches_enqueued)
        return fail(SDR_ERR_WOULD_BLOCK, "no batch waiting");
    // oldest first: parked batches are older than anything still in the ring
    if (!b->parked.empty() && b->parked.front().batch == b->deliver_next)
        return sdr_poll_parked(b, r);
    BatchSet &S = b->set[set_index(b, b->deliver_next)];
    if (S.res_batch != b->deliver_next)
        return fail(SDR_ERR_STATE, "results of the next batch are not where they should be");
    HIP_TRY(hipSetDevice(b->device));
    for (hipEvent_t e : {S.res_listen, S.res_peaks}) {
        if (wait) {
            // (the producer must not be held up while this thread waits for the device: the set cannot be parked
            // or reused meanwhile - its batch is the oldest undelivered one and parking waits for the same events)
            guard.unlock();
            const hipError_t we = hipEventSynchronize(e);
            guard.lock();
            HIP_TRY(we);
            if (S.res_batch != b->deliver_next)  // the producer parked it in the meantime: take it from there
                return sdr_poll_parked(b, r);
        } else {
            const hipError_t q = hipEventQuery(e);
            if (q == hipErrorNotReady)
                return fail(SDR_ERR_WOULD_BLOCK, "the oldest undelivered batch has not finished");
            HIP_TRY(q);
        }
    }
    const BatchMeta m{S.res_batch, S.res_first_frame, S.res_frames, S.res_chunks, S.res_count0, S.res_slots, &S.res_center};
    const int rc = deliver_block(b, S.res_host, m, r);
    if (rc == SDR_OK) {
        S.res_batch = -1;
        b->deliver_next++;
    }
    return rc;
}

int sdr_poll_peaks(sdr_bank *b, sdr_results *r, int wait)
{
    if (!b || !r)
        return fail(SDR_ERR_BAD_ARG, "null argument");
    if (r->struct_size != (int32_t)sizeof(sdr_results))
        return fail(SDR_ERR_BAD_ARG, "sdr_results.struct_size mismatch (ABI)");
    if (!b->results_on || !b->listen_pending)
        return fail(SDR_ERR_STATE, "no batch waits for its listen half");
    HIP_TRY(hipSetDevice(b->device));
    BatchSet &S = b->set[b->pend.set];
    if (wait) {
        HIP_TRY(hipEventSynchronize(S.res_peaks));
    } else {
        const hipError_t q = hipEventQuery(S.res_peaks);
        if (q == hipErrorNotReady)
            return fail(SDR_ERR_WOULD_BLOCK, "the batch's cumulations have not finished");
        HIP_TRY(q);
    }
    std::unique_lock<std::mutex> guard(b->res_mu);
    // (no listeners' output yet: res_slots is 0 until the listen half has run; the batch stays undelivered)
    const BatchMeta m{S.res_batch, S.res_first_frame, S.res_frames, S.res_chunks, S.res_count0, 0, &S.res_center};
    return deliver_block(b, S.res_host, m, r);
}

int sdr_read_drop_counters(sdr_bank *b, uint64_t *runes_dropped, uint64_t *edges_dropped)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    sdr::DropCounters dc{};
    HIP_TRY(hipMemcpy(&dc, b->drops.p, sizeof dc, hipMemcpyDeviceToHost));
    if (runes_dropped)
        *runes_dropped = dc.runes;
    if (edges_dropped)
        *edges_dropped = dc.edges;
    return SDR_OK;
}

int sdr_profile_enable(sdr_bank *b, int on)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    b->profiling = on != 0;
    return SDR_OK;
}

int sdr_profile_read(sdr_bank *b, int kernel, double *total_ms, int *launches)
{
    if (!b || kernel < 0 || kernel >= sdr::K_COUNT)
        return fail(SDR_ERR_BAD_ARG, "bad kernel id");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    if (total_ms)
        *total_ms = b->prof_ms[kernel];
    if (launches)
        *launches = b->prof_n[kernel];
    return SDR_OK;
}

int sdr_profile_reset(sdr_bank *b)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    for (int i = 0; i < sdr::K_COUNT; i++) {
        b->prof_ms[i] = 0;
        b->prof_n[i] = 0;
    }
    return SDR_OK;
}

#pragma GCC visibility pop
}  // extern "C"
