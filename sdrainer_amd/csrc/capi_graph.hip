// capi_graph.hip — graph mode (sdr_graph_capture / _launch / _release; BASELINE config 5 asks for a hipGraph-captured
// steady state of the body of Receiver.run, rx/receiver.go:353-463).  See the comment block in front of sdr_graph_batches.
#include "bank.h"

using namespace sdrcapi;

namespace sdrcapi {

// graph mode: what differs between the batches of a replay lives in device-side cursors; batch k's is written by a node
// of the FFT graph - its first one, for all the replay's batches (no kernel of an earlier replay reads these cursors any
// more: the FFT graph starts behind every reader of its phase)
__global__ void k_set_cursors(sdr::BatchCursor *dst, CursorPack v)
{
    if (threadIdx.x < RING)
        dst[threadIdx.x] = v.c[threadIdx.x];
}

hipError_t launch_set_cursors(sdr::BatchCursor *dst, const CursorPack &pack, hipStream_t stream)
{
    hipLaunchKernelGGL(k_set_cursors, dim3(1), dim3(64), 0, stream, dst, pack);
    return hipGetLastError();
}

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

}  // namespace sdrcapi

extern "C" {
#pragma GCC visibility push(default)

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

namespace {
// keep_sets: the replays' buffer sets stay allocated (a re-capture follows: sdr_graph_capture releases first)
static int graph_release(sdr_bank *b, bool keep_sets)
{
    int rc = sync_bank(b);
    if (rc)
        return rc;
    // results not polled yet move to the host-side queue (oldest first; only the last GRAPH_PHASES * RING batches can
    // still sit in a set): the eager ring takes over from here
    if ((rc = b->results->graph_end(b->batch_index)))
        return rc;
    drop_graphs(b);
    b->graph_ready = false;
    if (!keep_sets && b->set.size() > (size_t)RING) {
        // Graph mode's GRAPH_PHASES * RING buffer sets go back (3.4 GB at config 3's 2048-frame batches, 13 GB at its
        // 8192-frame ones or at config 5's share): a bank that tried graph mode once does not keep paying for it.  Every
        // undelivered batch of theirs has just been parked; the eager ring never indexes them.
        HIP_TRY(hipSetDevice(b->device));
        for (size_t i = RING; i < b->set.size(); i++) {
            b->set[i].release();
            host::ResultSet &rs = b->results->set((int)i);
            if (rs.block)
                (void)hipHostFree(rs.block);
            if (rs.ev_listen)
                (void)hipEventDestroy(static_cast<hipEvent_t>(rs.ev_listen));
            if (rs.ev_peaks)
                (void)hipEventDestroy(static_cast<hipEvent_t>(rs.ev_peaks));
            rs.block = nullptr;
            rs.ev_listen = rs.ev_peaks = nullptr;
        }
        b->set.resize(RING);
        if (b->last_set >= RING) {
            // the last batch lived in one of them: what it left on the device is gone with the set (its results were
            // parked above); the per-batch reads (sdr_read_*) answer "nothing" until the next batch
            b->last_set = 0;
            b->last_frames = 0;
            b->last_chunks = 0;
        }
    }
    return SDR_OK;
}
}  // namespace

int sdr_graph_release(sdr_bank *b)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    return graph_release(b, false);
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
    int rc = graph_release(b, true);  // (also drains the pipeline; the sets of an earlier capture are reused)
    if (rc)
        return rc;
    if (b->results_on)
        for (int i = 0; i < RING; i++)  // (the eager ring's sets, oldest batch first)
            if ((rc = b->results->park(set_index(b, std::max<int64_t>(b->batch_index - RING, 0) + i))))
                return rc;
    HIP_TRY(hipSetDevice(b->device));
    // the replays' buffer sets and events, once
    const size_t want = (size_t)RING + (size_t)GRAPH_PHASES * RING;
    if (b->set.size() < want) {
        const size_t have = b->set.size();
        b->set.resize(want);  // (capacity reserved by sdr_create: no reallocation)
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
    b->results->grow((int)want);
    if (b->results_on)
        for (size_t i = 0; i < want; i++)
            if ((rc = results_attach_set(b, (int)i)))
                return rc;
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
    b->results->graph_begin(b->batch_index);
    b->graph_replays = 0;
    b->graph_frames = n_frames;
    b->graph_slots = max_slots;
    b->graph_attach_gen = b->attach_gen;
    b->graph_results_on = b->results_on;
    b->graph_find_peaks = b->find_peaks;
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
    // the kernels that pack results, refine the cumulation and scan for peaks are nodes of the captured graphs or they are
    // not: a replay after sdr_enable_results / sdr_set_find_peaks changed either would publish batches no kernel fills
    if (b->results_on != b->graph_results_on || b->find_peaks != b->graph_find_peaks)
        return fail(SDR_ERR_STATE, "sdr_enable_results / sdr_set_find_peaks changed since the capture: capture again");
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
            const int prc = b->results->park(set0 + k);
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
            step(hipEventRecord(static_cast<hipEvent_t>(b->results->set(set0 + k).ev_peaks), b->stream[S_PEAKS]));
            step(hipEventRecord(static_cast<hipEvent_t>(b->results->set(set0 + k).ev_listen), b->stream[S_LISTEN]));
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
    for (int k = 0; k < RING; k++) {
        if (b->results_on) {
            host::BatchMeta m;
            m.batch = b->batch_index;
            m.first_frame = meta[k].first_frame;
            m.frames = n_frames;
            m.chunks = meta[k].chunks;
            m.count0 = meta[k].count0;
            m.slots = max_slots;
            {
                std::lock_guard<std::mutex> guard(b->center_mu);
                m.center = b->center_frequency;
            }
            b->results->publish(set0 + k, std::move(m), true);
        }
        b->last_set = set0 + k;
        b->last_frames = n_frames;
        b->last_chunks = meta[k].chunks;
        b->last_count0 = meta[k].count0;
        b->last_carry_in = pack.c[k].carry_in;
        b->batch_index++;
    }
    if (!b->results_on)
        b->results->note_enqueued(b->batch_index);
    b->cum_count = count;
    b->carry_cur = carry;
    b->total_frames = total;
    b->graph_replays++;
    return SDR_OK;
}

#pragma GCC visibility pop
}  // extern "C"
