// k_results.hip — the last stage of a batch when bulk delivery is on (sdr_enable_results / sdr_poll): copies what
// the batch produced for the host - keying edges and new text runes per listener, peaks per completed cumulation -
// straight into a block of pinned HOST memory, used entries only, 512 contiguous bytes per wave instruction.  The
// host never issues a copy and never drains a stream for results: it looks at an event per block.
// Consumer side of rx/rx.go:11-17 (Reporter) and rx/receiver.go:123,508-539 (the listeners' io.Writer).
#include <hip/hip_runtime.h>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "sdr_device.h"

namespace sdr {

// Listeners (or cumulations) per workgroup, one wave each.  One-wave workgroups go to 256 different CUs and each keeps
// an FFT workgroup - which needs a whole CU - off its CU while it lasts; sixteen waves per workgroup make the copy
// itself slow (one CU's path to PCIe).  Per pipelined step: 1: 0.2338 ms, 4: 0.2311, 8: 0.2347, 16: 0.2436.
#ifndef SDR_PACK_WAVES
#define SDR_PACK_WAVES 4
#endif
constexpr int PACK_WAVES = SDR_PACK_WAVES;

// one wave per (listener slot, band)
__global__ __launch_bounds__(64 * PACK_WAVES) void k_pack_listen(ListenerSlot *__restrict__ slots, const sdr_edge *__restrict__ edges,
                                                    const uint32_t *__restrict__ edge_counts,
                                                    const uint32_t *__restrict__ text,
                                                    const uint32_t *__restrict__ text_frames,
                                                    const DropCounters *__restrict__ drops, ResultsLayout lay,
                                                    int n_slots, unsigned char *__restrict__ host)
{
    const int l = blockIdx.x * PACK_WAVES + (int)(threadIdx.x >> 6), band = blockIdx.y, lane = threadIdx.x & 63;
    // the bank's drop counters travel with every batch, also one without a single listener slot (a bank created
    // with max_listeners == 0 has no slot array at all: nothing below may be touched then)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *reinterpret_cast<DropCounters *>(host + lay.off_drops) = *drops;  // (the decoder of this batch has finished)
    if (l >= n_slots)
        return;
    const size_t idx = (size_t)band * lay.max_listeners + l;
    ListenerSlot *slot = &slots[idx];
    uint32_t *h_edge_counts = reinterpret_cast<uint32_t *>(host + lay.off_edge_counts);
    uint32_t *h_text_counts = reinterpret_cast<uint32_t *>(host + lay.off_text_counts);
    if (!slot->active) {
        if (lane == 0) {
            h_edge_counts[idx] = 0;
            h_text_counts[idx] = 0;
        }
        return;
    }
    const uint32_t n_edges = min(edge_counts[idx], (uint32_t)lay.edge_cap);
    const uint32_t n_runes = min(slot->text_count, (uint32_t)lay.text_cap);
    const sdr_edge *src_e = edges + idx * lay.edge_cap;
    sdr_edge *dst_e = reinterpret_cast<sdr_edge *>(host + lay.off_edges) + idx * lay.edge_cap;
    for (uint32_t i = lane; i < n_edges; i += 64)
        dst_e[i] = src_e[i];
    const uint32_t *src_t = text + idx * lay.text_cap;
    uint32_t *dst_t = reinterpret_cast<uint32_t *>(host + lay.off_text) + idx * lay.text_cap;
    const uint32_t *src_f = text_frames + idx * lay.text_cap;
    uint32_t *dst_f = reinterpret_cast<uint32_t *>(host + lay.off_text_frames) + idx * lay.text_cap;
    for (uint32_t i = lane; i < n_runes; i += 64) {
        dst_t[i] = src_t[i];
        dst_f[i] = src_f[i];
    }
    if (lane == 0) {
        h_edge_counts[idx] = n_edges;
        h_text_counts[idx] = n_runes;
        slot->text_count = 0;  // delivered: the text buffer starts empty again
    }
}

// one wave per (completed cumulation, band)
__global__ __launch_bounds__(64 * PACK_WAVES) void k_pack_peaks(const DevPeak *__restrict__ peaks, const int *__restrict__ counts,
                                                   const BatchCursor *__restrict__ cur, ResultsLayout lay, int find_peaks,
                                                   int n_frames, int n_chunks, unsigned char *__restrict__ host)
{
    const int chunk = blockIdx.x * PACK_WAVES + (int)(threadIdx.x >> 6), band = blockIdx.y, lane = threadIdx.x & 63;
    if (chunk >= n_chunks || (cur && chunk >= chunks_completed(cur->count0, n_frames)))
        return;
    const size_t cidx = (size_t)band * lay.max_chunks + chunk;
    const int n_all = find_peaks ? counts[cidx] : 0;
    const int n = min(n_all, lay.max_peaks);
    const DevPeak *src = peaks + cidx * lay.max_peaks;
    DevPeak *dst = reinterpret_cast<DevPeak *>(host + lay.off_peaks) + cidx * lay.max_peaks;
    for (int i = lane; i < n; i += 64)
        dst[i] = src[i];
    if (lane == 0) {
        int *h_counts = reinterpret_cast<int *>(host + lay.off_peak_counts);
        h_counts[2 * cidx] = n;
        h_counts[2 * cidx + 1] = n_all;
    }
}

hipError_t launch_pack_listen(ListenerSlot *slots, const sdr_edge *edges, const uint32_t *edge_counts, const uint32_t *text,
                              const uint32_t *text_frames, const DropCounters *drops, ResultsLayout lay, int n_slots, int n_bands, unsigned char *host,
                              hipStream_t stream)
{
    // (at least one workgroup: it also delivers the drop counters)
    launch_kernel(k_pack_listen, dim3(n_slots > 0 ? (n_slots + PACK_WAVES - 1) / PACK_WAVES : 1, n_bands), dim3(64 * PACK_WAVES),
                       0, stream, slots, edges, edge_counts, text, text_frames, drops, lay, n_slots, host);
    return hipGetLastError();
}

hipError_t launch_pack_peaks(const DevPeak *peaks, const int *counts, const BatchCursor *cur, ResultsLayout lay, int find_peaks,
                             int n_frames, int n_chunks, int n_bands, unsigned char *host, hipStream_t stream)
{
    if (n_chunks <= 0)
        return hipSuccess;
    launch_kernel(k_pack_peaks, dim3((n_chunks + PACK_WAVES - 1) / PACK_WAVES, n_bands), dim3(64 * PACK_WAVES), 0, stream, peaks,
                       counts, cur, lay, find_peaks, n_frames, n_chunks, host);
    return hipGetLastError();
}

}  // namespace sdr
