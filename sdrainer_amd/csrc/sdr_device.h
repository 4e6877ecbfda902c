// sdr_device.h — HBM-resident state and launch geometry shared by the kernels and the C-ABI host code.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "fft_f64.h"

namespace sdr {

// Per-band state carried across batches (locals of Receiver.run, rx/receiver.go:339-346).
struct BandState {
    float nf_ring[SDR_NOISE_WINDOW];   // noiseFloorMean.values
    float dev_ring[SDR_NOISE_WINDOW];  // noiseDeviationMean.values
    float nf_sum, dev_sum;             // sumForMean of both
    int32_t next;                      // shared ring cursor (both means are Put once per frame)
    float peak_threshold;              // r.peakThreshold
};

// One listener of a band's pool (rx/listener.go:19-32 + cw/spectral.go:19-23).
struct ListenerSlot {
    int32_t active;
    int32_t bin;  // Peak.SignalBin
    cw::Debouncer deb;
    cw::DecoderState dec;
    uint32_t text_count;    // runes in the text buffer not yet read by the host
    uint32_t text_dropped;  // runes dropped because the buffer was full (since attach)
    // Bank frame index (32 bits, wrapping like every frame number on the device) from which the listener listens, and
    // from which the FFT kernel has tapped its bin (k_fft_psd.hip "The tap").  Equal for a listener attached between
    // batches.  A listener bound to a batch whose spectra already exist (sdr_attach_at: strain-mode discovery without a
    // host round trip per cumulation) starts inside that batch - frames before start_frame do not reach its debouncer
    // or decoder - and reads its bin from the retained psd rows up to tapped_from.
    uint32_t start_frame;
    uint32_t tapped_from;
};

// Bank-wide overflow counters (device memory): what the reference's never-dropping io.Writer would have kept.
struct DropCounters {
    unsigned long long runes;  // decoded runes that found the listener's text buffer full
    unsigned long long edges;  // keying edges beyond a batch's per-listener edge buffer
};

// What k_find_peaks hands to the host, which finishes dsp.Peak (frequencies are float64 -> int).
struct DevPeak {
    int32_t from, to, signal_bin;
    float signal_value;
    float y1, y2, y3;  // cumulation at signal_bin-1, signal_bin, signal_bin+1 (PeakCenterCorrection)
};

struct NoiseGeom {
    int n;          // block size
    int edge;       // edgeWidth
    int window;     // windowSize = (N - 2 edge) / 10
    int n_windows;  // windows the reference's loop actually evaluates (9 or 10)
    double inv_n2;  // 1 / N^2 (exact power of two)
};

// What changes from batch to batch and would otherwise be a launch parameter.  In graph mode (sdr_graph_*) the
// launches of several batches are captured once and replayed, so these values live in device memory: the host
// writes the cursors of the batches of one replay into pinned memory and the graph's first node uploads them.
// A null cursor pointer means "use the launch parameters" (the eager path).
struct BatchCursor {
    const float *iq;      // this batch's input frames
    uint32_t frame_base;  // bank frame index of its first frame
    int32_t count0;       // cumulationCount before its first frame
    int32_t carry_in;     // which of the two carry buffers holds the cumulation carried in (0 / 1)
    int32_t reserved;
};

struct ListenGeom {
    int n, stride, max_listeners, text_cap, edge_cap, bit_words, trace;
    uint32_t frame_base;
};

struct CumGeom {
    int n, stride, n_frames, count0, max_chunks;
};

// k_noise_scan.hip: one pass over a batch's psd for FindNoiseFloor's sums and the cumulations' bounds
struct ScanGeom {
    int n, edge, window, n_windows;          // NoiseGeom
    int stride, n_frames, count0, max_chunks;  // CumGeom
    int piece;                               // edge pieces hold at most this many bins (64 per-lane slots)
    int do_bound;                            // also the bound of every cumulation the batch completes
    int fpw;                                 // do_bound == 0: frames per workgroup (the slots play no part)
    int parts;                               // do_bound == 1: workgroups a slot's frames are dealt over (1 or 2)
};

struct PeakGeom {
    int n, stride, count0, max_chunks, max_peaks;
};

// Layout of one batch's block of pinned host memory (k_results.hip): byte offsets of its arrays
//   peak_counts [band][max_chunks][2] int32 (stored, found)   peaks [band][max_chunks][max_peaks] DevPeak
//   edge_counts [band][L] uint32                              edges [band][L][edge_cap] sdr_edge
//   text_counts [band][L] uint32                              text  [band][L][text_cap] uint32 runes
//                                                             text_frames [band][L][text_cap] uint32
//   drops       DropCounters of the bank as of this batch
struct ResultsLayout {
    int max_listeners, max_chunks, max_peaks, edge_cap, text_cap;
    size_t off_peak_counts, off_peaks, off_edge_counts, off_edges, off_text_counts, off_text, off_text_frames, off_drops, bytes;
};

enum KernelId {
    K_FFT = 0, K_WINDOW_MEANS, K_NOISE_STATS, K_THRESHOLDS, K_LISTEN_GATHER, K_CUMULATE, K_FIND_PEAKS, K_LISTEN_DECODE,
    K_COUNT
};

// A stage's completion event can ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL's stopEvent)
// instead of a hipEventRecord behind the kernel.  The record is a barrier packet of its own: the queue's next kernel
// waits for the command processor to retire it, which on the FFT queue was 27-36 us per batch with nothing running
// (0.200 ms per step for a 0.166 ms kernel launched back to back).  The caller arms `t_done_event` right before a
// launch_* call; the first kernel launched through launch_kernel takes it.
inline thread_local hipEvent_t t_done_event = nullptr;
template <class F, class... A>
inline void launch_kernel(F kernel, dim3 grid, dim3 block, unsigned lds_bytes, hipStream_t stream, A... args)
{
    hipEvent_t done = t_done_event;
    t_done_event = nullptr;
    if (done)
        hipExtLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, nullptr, done, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, args...);
}

// The listeners' bins of every band and where their psd values go (k_fft_psd.hip "The tap").
struct FftTap {
    const int32_t *bins;  // [band][stride], -1 = free slot
    float *out;           // [band][out_stride frames][stride]
    int n;                // slots in use (high-water mark over the bands); 0 = no tap
    int stride;           // max_listeners
    // k_fft_r32 only (null otherwise): psd at bin - 1, bin, bin + 1 of every slot, [band][out_stride frames][stride][4],
    // and the bins those rows were taken at, [band][stride] (-1 = none): what k_cum_refine reads instead of psd columns
    float *wide = nullptr;
    int32_t *used = nullptr;
};

hipError_t launch_fft(int logn, const float *iq, const BatchCursor *cur, const fft64::cplx *tw, float *psd, int n_frames,
                      int n_bands, int in_stride, int out_stride, FftTap tap, hipStream_t stream);
int twiddle_count(int logn);
void build_twiddles(int logn, const double *wre, const double *wim, fft64::cplx *out);
// k_fft_r32.hip: N = 16384 as 512 threads x 32 points with the next frame prefetched into registers (own twiddle layout)
hipError_t launch_fft_r32(const float *iq, const BatchCursor *cur, const fft64::cplx *tw, float *psd, int n_frames, int n_bands,
                          int in_stride, int out_stride, FftTap tap, hipStream_t stream);
int r32_twiddle_count();
void r32_build_twiddles(const double *wre, const double *wim, fft64::cplx *out);
hipError_t launch_window_means(const float *psd, double *win_mean, NoiseGeom g, int n_frames, int n_bands, int stride,
                               hipStream_t stream);
hipError_t launch_noise_stats(const float *psd, const double *win_mean, sdr_frame_rec *recs, NoiseGeom g, int n_frames,
                              int n_bands, int stride, hipStream_t stream);
hipError_t launch_mfma_order_probe(unsigned *mismatches, int order, hipStream_t stream);  // k_noise.hip: sdr_self_check
hipError_t launch_thresholds(sdr_frame_rec *recs, BandState *st, int n_frames, int n_bands, int stride,
                             hipStream_t stream);
hipError_t launch_listen_gather(const float *tap, const float *psd, const sdr_frame_rec *recs, const ListenerSlot *slots, const void *db_tab,
                                uint64_t *raw_bits, float *tr_values, uint8_t *tr_raw, const BatchCursor *cur, ListenGeom g, int n_frames,
                                int n_slots, int n_bands, hipStream_t stream);
hipError_t launch_listen_decode(ListenerSlot *slots, const uint16_t *morse, const uint64_t *raw_bits,
                                uint64_t *deb_bits, uint32_t *text, uint32_t *text_frames, sdr_edge *edges,
                                uint32_t *edge_counts, uint8_t *tr_deb, DropCounters *drops, const BatchCursor *cur, ListenGeom g,
                                int n_frames, int n_bands, uint32_t *edge_pos, int pos_stride, hipStream_t stream);
hipError_t launch_listener_stop(ListenerSlot *slot, const uint16_t *morse, uint32_t *text, uint32_t *text_frames, int text_cap,
                                uint32_t frame, DropCounters *drops, hipStream_t stream);
hipError_t launch_set_debounce(ListenerSlot *slots, int n, int threshold, hipStream_t stream);
// bound_done: k_psd_scan has written the bounds of the completed cumulations (slot 0's as its raw unit count: the carry
// is added here, on the stream the carry is produced on)
hipError_t launch_cumulate(const float *psd, const void *db_tab, float *carry0, float *carry1, int carry_in, float *cum_out,
                           const float *cum_part, const BatchCursor *cur, CumGeom g, int n_slots, int n_bands, bool bound_done, hipStream_t stream);
int scan_parts(int n_slots, int n_bands);
bool cum_bound_pays(int n_frames, int n_bands, int n);
// k_noise_scan.hip: the FindNoiseFloor fields of every frame's record, certified or literal, and (do_bound) the unit counts
// of the completed cumulations - one kernel
hipError_t launch_psd_scan(const float *psd, sdr_frame_rec *recs, float *cum_out, float *cum_part, const BatchCursor *cur, NoiseGeom ng,
                           CumGeom cg, int n_slots, int n_bands, bool do_bound, int force_exact, hipStream_t stream);
hipError_t launch_noise_exact_check(const float *psd_band, sdr_frame_rec *recs_band, NoiseGeom ng, int n_frames, unsigned *mismatches,
                                    hipStream_t stream);
hipError_t launch_spectrum_row(const float *psd_row, float *out, int n, hipStream_t stream);
hipError_t launch_pack_listen(ListenerSlot *slots, const sdr_edge *edges, const uint32_t *edge_counts, const uint32_t *text,
                              const uint32_t *text_frames, const DropCounters *drops, ResultsLayout lay, int n_slots, int n_bands, unsigned char *host,
                              hipStream_t stream);
hipError_t launch_pack_peaks(const DevPeak *peaks, const int *counts, const BatchCursor *cur, ResultsLayout lay, int find_peaks,
                             int n_frames, int n_chunks, int n_bands, unsigned char *host, hipStream_t stream);
// cumulations a batch of n_frames completes when it starts at cumulationCount count0
__host__ __device__ inline int chunks_completed(int count0, int n_frames)
{
    const int first_len = SDR_CUMULATION_SIZE - count0;
    return n_frames >= first_len ? 1 + (n_frames - first_len) / SDR_CUMULATION_SIZE : 0;
}
hipError_t launch_unpack_be16(const uint8_t *raw, float *out, size_t n_values, hipStream_t stream);
hipError_t launch_find_peaks(float *cum, const float *psd, const void *db_tab, const float *carry0, const float *carry1, int carry_in,
                             const sdr_frame_rec *recs, DevPeak *peaks, int *counts, const BatchCursor *cur, PeakGeom g, int n_frames,
                             int n_chunks, int n_bands, FftTap tap, hipStream_t stream);  // tap: .wide / .used / .n / .stride of this batch's FFT (or null)
// does launch_fft, called like this, leave the wide tap (k_fft_r32 does; the sixteen-point kernels do not)?
bool fft_writes_wide_tap(int logn, int n_frames, int n_bands, int tap_n);
hipError_t launch_cumulation_row(const float *psd_band, const void *db_tab, const float *carry_in_band, float *row_out, CumGeom g, int slot,
                                 hipStream_t stream);

}  // namespace sdr
