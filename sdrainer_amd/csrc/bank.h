// bank.h — internal to libsdrainer_hip.so: the bank of receivers behind the C ABI (include/sdrainer_hip.h) and what the
// translation units that implement it share.
//   capi_bank.hip     create / destroy / control calls (attach, detach, setters), profiling
//   capi_process.hip  the scheduler: one batch's kernels over the bank's four streams (eager, captured, deferred listen
//                     half), the staged host input
//   capi_results.hip  bulk delivery (sdr_enable_results / sdr_poll / sdr_poll_peaks) over host/delivery.h
//   capi_graph.hip    graph mode (sdr_graph_*)
//   capi_read.hip     per-listener / per-batch reads and the scope tap (they synchronise)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdrainer_hip.h"
#include "host/delivery.h"
#include "host/frequency_mapping.h"
#include "sdr_device.h"

namespace sdrcapi {

inline thread_local std::string g_last_error;

inline int fail(int code, const std::string &msg)
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

inline int ilog2(int n)
{
    int s = 0;
    while ((1 << s) < n)
        s++;
    return s;
}

inline const char *const kKernelNames[sdr::K_COUNT] = {"k_fft_psd",       "k_window_means", "k_noise_stats", "k_thresholds",
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
    DevBuf<float> tapw;               // [band][max_batch][L][4] psd at bin - 1, bin, bin + 1, 0 (k_fft_r32's wide tap; N = 16384 only)
    DevBuf<int32_t> tap_used;         // [band][L] the bins tapw was taken at (-1 = none)
    DevBuf<double> win_mean;          // [band][max_batch][10] (the chain kernels of SDR_NOISE_PATH=chains only)
    DevBuf<sdr_frame_rec> recs;       // [band][max_batch]
    DevBuf<uint64_t> raw_bits, bits;  // [band][L][bit_words] before / after the debouncer
    DevBuf<sdr_edge> edges;           // [band][L][edge_cap]
    DevBuf<uint32_t> edge_counts;     // [band][L] edges produced by this batch
    DevBuf<float> tr_values;          // [band][max_batch][L] (trace only)
    DevBuf<uint8_t> tr_raw, tr_deb;
    DevBuf<sdr::ListenerSlot> slots_before;  // [band][L] the slots as the batch's decoders found them (trace only: sdr_scope_read_decode)
    DevBuf<float> cum_out;            // [band][max_chunks][N]
    DevBuf<float> cum_part;           // [band][max_chunks][N]: the second partial unit count of k_psd_scan (a slot's frames over two workgroups)
    DevBuf<sdr::DevPeak> dev_peaks;   // [band][max_chunks][max_peaks]
    DevBuf<int> peak_counts;          // [band][max_chunks]
    hipEvent_t done[sdr::K_COUNT] = {};  // recorded behind each kernel of the batch that used this set
    // (bulk delivery: the set's block of pinned host memory, its two events and what the host knows about the batch in
    // it live in the bank's host::Delivery - results.set(i) belongs to set i)
    void release()
    {
        psd.release();
        tap.release();
        tapw.release();
        tap_used.release();
        win_mean.release();
        cum_part.release();
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
        for (auto &e : done) {
            if (e)
                (void)hipEventDestroy(e);
            e = nullptr;
        }
    }
};


enum Parts { PART_SPECTRA = 1, PART_LISTEN = 2, PART_ALL = 3 };

// graph mode: what differs between the batches of a replay lives in device-side cursors, written by the first node of
// the FFT graph from its kernel argument (capi_graph.hip)
struct CursorPack {
    sdr::BatchCursor c[RING];
};

}  // namespace sdrcapi

struct sdr_bank {
    // (the names of namespace sdrcapi, unqualified)
    template <class T>
    using DevBuf = sdrcapi::DevBuf<T>;
    using BatchSet = sdrcapi::BatchSet;
    static constexpr int N_STAGES = sdrcapi::N_STAGES, GRAPH_PHASES = sdrcapi::GRAPH_PHASES, N_GRAPHS = sdrcapi::N_GRAPHS, RING = sdrcapi::RING;

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
    std::vector<BatchSet> set;  // RING sets; graph mode adds its own (sdr_graph_capture); capacity reserved at creation: references stay valid
    DevBuf<sdr::BandState> band_state;
    DevBuf<sdr::ListenerSlot> slots;  // [band][max_listeners]
    DevBuf<uint16_t> morse;
    DevBuf<uint32_t> text;         // [band][L][text_cap] decoded runes not yet read / delivered
    DevBuf<uint32_t> text_frames;  // [band][L][text_cap] bank frame index of the Tick that wrote each rune
    DevBuf<uint32_t> edge_pos;     // [band][L][max_batch_frames] k_listen_decode's scratch: every edge's position in the batch (one decoder launch runs at a time: each needs the state the one before left)
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
    int last_carry_in = 0;  // which carry buffer the last batch's first cumulation started from (exact rows on demand)
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
    // what the captured graphs have baked in besides the listeners: the packing kernels exist only if results were on, the
    // refinement / peak-scan nodes only if find_peaks was
    bool graph_results_on = false;
    int graph_find_peaks = 0;
    // deferred listen half (sdr_defer_listen): the batch whose spectra exist and whose listeners have not run yet
    bool defer_listen = false, listen_pending = false;
    struct PendingListen {
        int set = 0, frames = 0;
        int64_t first_frame = 0, batch = 0;
    } pend;
    std::vector<int> late_attached;  // flattened slot indices bound by sdr_attach_at, not on the device yet
    // bulk delivery (host/delivery.h): sdr_poll may run on a consumer thread of its own beside the producer's process
    // calls (the reference's Reporter is called from other goroutines too); the bookkeeping - which finished batch sits in
    // which set's pinned block or in the parked queue, who takes it - is the Delivery's, under its mutex
    bool results_on = false;
    sdr::ResultsLayout res_layout{};
    std::unique_ptr<host::DeliveryBackend> res_backend;
    std::unique_ptr<host::Delivery> results;
    std::mutex center_mu;  // center_frequency: written by sdr_set_center_frequency, snapshot per batch

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

namespace sdrcapi {

// (capi_bank.hip)
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
void resolve_profile(sdr_bank *b);
int sync_bank(sdr_bank *b);
int check_band(sdr_bank *b, int band);
int check_listener(sdr_bank *b, int band, int lid);
hipError_t alloc_set(sdr_bank *b, BatchSet &S);
// (capi_process.hip)
int flush_late_attached(sdr_bank *b);
int process_device_body(sdr_bank *b, const float *iq_dev, int n_frames, int in_stride, int capture_k = -1, int capture_stage = -1,
                        int parts = PART_ALL);
int process_device_impl(sdr_bank *b, const float *iq_dev, int n_frames, int in_stride);
// (capi_results.hip)
sdr::ResultsLayout make_results_layout(const sdr_bank *b);
int results_attach_set(sdr_bank *b, int set_idx);  // the set's pinned block and events, once bulk delivery is on
// (capi_graph.hip)
void drop_graphs(sdr_bank *b);
hipError_t launch_set_cursors(sdr::BatchCursor *dst, const CursorPack &pack, hipStream_t stream);

// the buffer set of batch `batch` (graph mode: phase-major, the sets behind the eager ring's)
inline int set_index(const sdr_bank *b, int64_t batch)
{
    if (!b->graph_ready)
        return (int)(batch % RING);
    return RING + (int)((batch - b->graph_base) % (GRAPH_PHASES * RING));
}

}  // namespace sdrcapi
