// capi_process.hip — the scheduler of libsdrainer_hip.so: one batch's kernels over the bank's four streams.
//
// The FFT kernel is throughput work that fills the whole chip; everything after it is a set of short, strictly ordered
// chains (float64 noise-floor sums, the rolling means, the per-signal decoders) that occupy a handful of CUs for a long
// time.  Run back to back they would leave the chip idle most of the step, so a bank is a software pipeline over four
// streams (four is also the number of hardware queues HIP maps streams to by default; more streams alias and serialise):
//
//   fft     k_fft_psd(i)                                          (the caller's stream)
//   noise   k_window_means(i) -> k_noise_stats(i)
//   peaks   k_thresholds(i) -> k_cum_bound(i), k_cumulate(i) -> k_find_peaks(i) (-> k_pack_peaks(i))
//   listen  k_listen_gather(i) -> k_listen_decode(i) (-> k_pack_listen(i))
//
// Batch i's per-batch buffers (psd, tap, frame records, keying bits, peaks ...) live in set i % RING, and one event per
// kernel orders the stages across streams (kDefaultPlan, process_device_body): window means and cumulate after the FFT;
// thresholds after the noise statistics; gather after thresholds; find_peaks after cumulate and thresholds; fft(i) after
// every reader of set i % RING from batch i - RING.  State that is carried from frame to frame is only ever touched by one
// kernel, whose stream keeps it in batch order.  Results leave the device in bulk (capi_results.hip) or are read after
// sdr_sync(), which drains every stream (capi_read.hip).
//
// The same body is driven three more ways: recorded into graphs (capi_graph.hip: capture_k / capture_stage), as the
// deferred listen half (sdr_defer_listen ...: the spectral stages of a batch first, listeners bound to frames inside it,
// then the listen stages) and from the staged host input (sdr_push_* / sdr_process_staged: three pinned staging sets,
// uploads on a copy stream).
#include <string>

#include "bank.h"

using namespace sdrcapi;

namespace sdrcapi {

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
    host::ResultSet &RS = b->results->set(si);  // (its block and events exist once bulk delivery is on)
    int plan[sdr::K_COUNT];
    for (int k = 0; k < sdr::K_COUNT; k++)
        plan[k] = kDefaultPlan[k];
    // Small geometries (one band of N <= 8192, two of 4096 ...): the FFT of a batch is shorter than its decoders, whose time
    // goes with the frames, not the samples - the listen stream is the longest, and the gather, which carries no state
    // from batch to batch and so may run on any stream, moves behind the thresholds it waits for anyway (config 2:
    // 0.206 -> 0.142 ms per 4096-frame step, 80 -> 118 GS/s; config 3 unchanged within a percent either way, config 5's
    // share 10 % SLOWER with it: its peaks stream is the full one).  Not under capture: a replay's graphs are cut by stream.
    if (!cap && (long)B * N <= 8192)
        plan[sdr::K_LISTEN_GATHER] = S_PEAKS;
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
    int max_slots = 0;
    for (int i = 0; i < B; i++)
        max_slots = std::max(max_slots, b->n_slots[i]);
    if (do_spectra) {
    // With bulk delivery on, the set's previous batch must have been delivered (or be parked) before its block is
    // written again - and a delivered batch is a finished one: every reader of the set is done, the queries below
    // succeed and the FFT queue gets no barrier packets at all (each costs the command processor microseconds between
    // two FFT kernels, and the FFT queue is the one that bounds the step).
    static const bool park_first = !(getenv("SDR_PARK_FIRST") && atoi(getenv("SDR_PARK_FIRST")) == 0);
    if (b->results_on && !cap && park_first) {
        const int prc = b->results->park(si);
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
        const int prc = b->results->park(si);
        if (prc)
            return prc;
    }
    if (cap && SDR_ON(sdr::K_FFT) && capture_k % RING == 0)  // the replay's cursors, in front of its first FFT
        HIP_TRY(launch_set_cursors(b->cursors.p + capture_k, CursorPack{}, stream_of(sdr::K_FFT)));
    {
        ProfScope ps(b, sdr::K_FFT, stream_of(sdr::K_FFT));
        SDR_ARM(sdr::K_FFT);
        sdr::FftTap tap{b->tap_bins.p, S.tap.p, max_slots, c.max_listeners};
        tap.wide = S.tapw.p;
        tap.used = S.tap_used.p;
        SDR_LAUNCH(sdr::K_FFT, sdr::launch_fft(b->logn, iq_dev, cur, b->tw.p, S.psd.p, n_frames, B, in_stride, stride, tap,
                                               stream_of(sdr::K_FFT)));
    }
    SDR_DONE(sdr::K_FFT);

    // noise floor (stateless per batch), then the rolling means -> thresholds, in batch order.
    // Two ways (SDR_NOISE_PATH = scan | chains; scan unless told otherwise):
    //   scan    k_noise_scan.hip: ONE pass over the psd for FindNoiseFloor's sums and - where it pays - the bounds of the
    //           cumulations the batch completes; the reference's values where they are consumed (noise_cert.h), the
    //           literal loops for the few frames that cannot be certified;
    //   chains  k_noise.hip: the ordered float64 chains of rounds 1-4 (and k_cum_bound on the peaks stream).
    static const bool scan_path = !(getenv("SDR_NOISE_PATH") && std::string(getenv("SDR_NOISE_PATH")) == "chains");
    static const int force_exact = getenv("SDR_NOISE_FORCE_EXACT") ? atoi(getenv("SDR_NOISE_FORCE_EXACT")) : 0;  // (tests)
    const int scan_count0 = b->cum_count;
    int scan_slots = 1;
    if (n_frames >= SDR_CUMULATION_SIZE - scan_count0)
        scan_slots = 1 + (n_frames - (SDR_CUMULATION_SIZE - scan_count0) + SDR_CUMULATION_SIZE - 1) / SDR_CUMULATION_SIZE;  // (the last one may stay open)
    if (cap)  // whatever cumulationCount the replayed batch starts at
        scan_slots = sdr::chunks_completed(SDR_CUMULATION_SIZE - 1, n_frames) + 1;
    const bool scan_bound = scan_path && sdr::cum_bound_pays(n_frames, B, N);
    SDR_AFTER(sdr::K_WINDOW_MEANS, sdr::K_FFT);
    {
        ProfScope ps(b, sdr::K_WINDOW_MEANS, stream_of(sdr::K_WINDOW_MEANS));
        SDR_ARM(sdr::K_WINDOW_MEANS);
        if (scan_path) {
            const sdr::CumGeom scg{N, stride, n_frames, scan_count0, b->max_chunks};
            SDR_LAUNCH(sdr::K_WINDOW_MEANS, sdr::launch_psd_scan(S.psd.p, S.recs.p, S.cum_out.p, S.cum_part.p, cur, ng, scg, scan_slots, B, scan_bound,
                                                                 force_exact, stream_of(sdr::K_WINDOW_MEANS)));
        } else {
            SDR_LAUNCH(sdr::K_WINDOW_MEANS, sdr::launch_window_means(S.psd.p, S.win_mean.p, ng, n_frames, B, stride,
                                                                     stream_of(sdr::K_WINDOW_MEANS)));
        }
    }
    SDR_DONE(sdr::K_WINDOW_MEANS);
    SDR_AFTER(sdr::K_NOISE_STATS, sdr::K_WINDOW_MEANS);
    {
        ProfScope ps(b, sdr::K_NOISE_STATS, stream_of(sdr::K_NOISE_STATS));
        SDR_ARM(sdr::K_NOISE_STATS);
        if (!scan_path)  // (the scan kernel has finished the records itself: this stage launches nothing, its event is recorded below)
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
                                                                   lg, n_frames, B, b->edge_pos.p, c.max_batch_frames, stream_of(sdr::K_LISTEN_DECODE)));
    }
    if (b->results_on && SDR_ON(sdr::K_LISTEN_DECODE)) {
        // delivery of this batch's edges and runes, behind the decoder on its stream; the decoder's event is
        // recorded behind it so that the set is not reused before the copy to the host has happened
        SDR_ARM(sdr::K_LISTEN_DECODE);
        HIP_TRY(sdr::launch_pack_listen(b->slots.p, S.edges.p, S.edge_counts.p, b->text.p, b->text_frames.p, b->drops.p, b->res_layout, max_slots, B,
                                        RS.block, stream_of(sdr::K_LISTEN_DECODE)));
        if (!cap)  // (a replay records it behind the listen graph)
            HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(RS.ev_listen), stream_of(sdr::K_LISTEN_DECODE)));
    }
    SDR_DONE(sdr::K_LISTEN_DECODE);
    }  // do_listen
    if (!do_spectra) {
        // the batch is complete: sdr_poll may have it
        b->results->complete(si, max_slots, b->pend.batch);
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
    // (the scan wrote the bounds of the completed cumulations on the noise stream: the carry is added to slot 0's here)
    static const bool scan_path_c = !(getenv("SDR_NOISE_PATH") && std::string(getenv("SDR_NOISE_PATH")) == "chains");
    const bool scan_bound_done = scan_path_c && sdr::cum_bound_pays(n_frames, B, N);
    SDR_AFTER(sdr::K_CUMULATE, sdr::K_FFT);
    if (scan_bound_done)
        SDR_AFTER(sdr::K_CUMULATE, sdr::K_WINDOW_MEANS);
    {
        ProfScope ps(b, sdr::K_CUMULATE, stream_of(sdr::K_CUMULATE));
        SDR_ARM(sdr::K_CUMULATE);
        sdr::CumGeom cg{N, stride, n_frames, count0, b->max_chunks};
        SDR_LAUNCH(sdr::K_CUMULATE, sdr::launch_cumulate(S.psd.p, b->db_tab.p, b->carry[0].p, b->carry[1].p, b->carry_cur,
                                                         S.cum_out.p, S.cum_part.p, cur, cg, n_slots_c, B, scan_bound_done, stream_of(sdr::K_CUMULATE)));
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
        // (reads the carry buffer this batch's cumulation started from: the next batch's k_cumulate, which writes that
        // buffer, follows on the same stream)
        // the wide tap this batch's FFT left, if it was the kernel that leaves one (k_cum_refine reads the signals' columns there)
        sdr::FftTap wide_tap{nullptr, nullptr, max_slots, c.max_listeners};
        if (S.tapw.p && sdr::fft_writes_wide_tap(b->logn, n_frames, B, max_slots)) {
            wide_tap.wide = S.tapw.p;
            wide_tap.used = S.tap_used.p;
        }
        SDR_LAUNCH(sdr::K_FIND_PEAKS, sdr::launch_find_peaks(S.cum_out.p, S.psd.p, b->db_tab.p, b->carry[0].p, b->carry[1].p, b->carry_cur, S.recs.p,
                                                             S.dev_peaks.p, S.peak_counts.p, cur, pg, n_frames, n_chunks, B, wide_tap,
                                                             stream_of(sdr::K_FIND_PEAKS)));
    }
    if (b->results_on && SDR_ON(sdr::K_FIND_PEAKS)) {
        SDR_ARM(sdr::K_FIND_PEAKS);
        HIP_TRY(sdr::launch_pack_peaks(S.dev_peaks.p, S.peak_counts.p, cur, b->res_layout, b->find_peaks, n_frames, n_chunks, B,
                                       RS.block, stream_of(sdr::K_FIND_PEAKS)));
        if (!cap) {
            HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(RS.ev_peaks), stream_of(sdr::K_FIND_PEAKS)));
            host::BatchMeta m;
            m.batch = b->batch_index;
            m.first_frame = b->total_frames;
            m.frames = n_frames;
            m.chunks = n_chunks;
            m.count0 = count0;
            m.slots = do_listen ? max_slots : 0;  // (sdr_poll_peaks delivers the spectral half; the listen half fills this in)
            {
                std::lock_guard<std::mutex> guard(b->center_mu);
                m.center = b->center_frequency;
            }
            b->results->publish(si, std::move(m), do_listen);
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
    b->last_carry_in = b->carry_cur;
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
    if (!b->results_on)
        b->results->note_enqueued(b->batch_index);
    return SDR_OK;
}

}  // namespace sdrcapi

extern "C" {
#pragma GCC visibility push(default)

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


#pragma GCC visibility pop
}  // extern "C"
