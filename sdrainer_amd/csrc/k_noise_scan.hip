// k_noise_scan.hip — ONE pass over a batch's psd rows for everything the tail needs from all of them:
//   * dsp.FindNoiseFloor (dsp/fft.go:215-252): per frame and window the order-free sums S1 = sum x, S2 = sum x^2, from
//     which k_noise_finish takes the reference's minimum mean, variance and rolling-mean inputs exactly where they are
//     consumed (noise_cert.h: brackets around what the reference's ORDERED sums can be; a frame whose brackets straddle a
//     float32 rounding boundary or another window's mean is flagged and k_noise_exact_list runs the literal loops);
//   * the cumulation (rx/receiver.go:404-407): the upper bound of every cumulation the batch completes, from the psd words'
//     top halves (k_peaks.hip, gomath.h cum_bound_*), which k_cum_refine / k_find_peaks turn into the exact peak list.
// Rounds 1-4 read the psd three times for this (window chains 73 % of it, variance chains up to 73 %, the bound all of it:
// 31 of a step's 181 CU-ms; a stage costs the CU time it HOLDS, and these hold it for bytes / what a CU's memory pipeline
// delivers) and ran 24 000 strictly ordered float64 additions per frame that nothing downstream can tell from any other
// order except three times in 10^5.
//
// Geometry.  A row is cut into SEGMENTS: the reference's windows (W values each) and pieces of the two edges (at most
// 64 JMAX values).  A WAVE owns one segment of one cumulation slot (up to 100 consecutive frames): lane l holds the bins
// begin + l + 64 j; per frame it adds its values into the frame's S1 / S2 (a lane's <= JMAX values in sequence, then six
// butterfly levels across the lanes: noise_cert.h kScanTerms) and into its bins' running unit counts, which become the
// slot's bound after the last frame.  A workgroup = the segments of one (band, slot): 14 waves at N = 16384.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "gomath.h"
#include "noise_cert.h"
#include "sdr_device.h"

namespace sdr {

constexpr int kScanMaxWaves = 16;
#if !defined(SDR_SCAN_AHEAD)
#define SDR_SCAN_AHEAD 2  // frames whose loads are in flight ahead of the one being added up
#endif

__device__ __forceinline__ void scan_slot_frames(int slot, int count0, int *begin, int *len)
{
    const int first_len = SDR_CUMULATION_SIZE - count0;
    *begin = slot == 0 ? 0 : first_len + (slot - 1) * SDR_CUMULATION_SIZE;
    *len = slot == 0 ? first_len : SDR_CUMULATION_SIZE;
}

// Sum of a float64 over the wave, in lane 63, by DPP moves (no LDS round trips: as twelve ds_bpermute pairs per frame the
// reduction WAS the kernel - 0.27 ms for a batch whose loads and additions need 0.07).  The classic sequence: row_shr 1, 2,
// 3 (-> every fourth lane holds its group of four... ) is replaced by the shift / broadcast ladder below; lanes a step
// does not write receive 0.0 (old = 0 with the row / bank masks), which adds nothing.  Six additions per lane deep, like
// the butterfly it replaces (noise_cert.h kScanTerms).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_moved(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, BANK_MASK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double x)
{
    x += dpp_moved<0x111, 0xf, 0xf>(x);  // row_shr:1
    x += dpp_moved<0x112, 0xf, 0xf>(x);  // row_shr:2
    x += dpp_moved<0x114, 0xf, 0xe>(x);  // row_shr:4, banks 1-3
    x += dpp_moved<0x118, 0xf, 0xc>(x);  // row_shr:8, banks 2-3: lane 15 of every row holds the row's sum
    x += dpp_moved<0x142, 0xa, 0xf>(x);  // row_bcast:15 into rows 1 and 3
    x += dpp_moved<0x143, 0xc, 0xf>(x);  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
    return x;
}

template <int JMAX>
__global__ __launch_bounds__(64 * kScanMaxWaves) void k_psd_scan(const float *__restrict__ psd, double *__restrict__ wsum,
                                                                  float *__restrict__ cum_out, const BatchCursor *__restrict__ cur,
                                                                  ScanGeom g, double a128, double per_frame)
{
    int count0 = g.count0;
    if (cur)  // graph replay: this batch's cumulation phase comes from device memory
        count0 = cur->count0;
    const int slot = blockIdx.x, band = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n_waves = (int)blockDim.x >> 6;
    int f_begin, f_len;
    scan_slot_frames(slot, count0, &f_begin, &f_len);
    if (f_begin >= g.n_frames)
        return;  // (workgroup-uniform)
    const int f_end = min(f_begin + f_len, g.n_frames);
    const bool complete = f_begin + f_len <= g.n_frames;  // the batch completes this cumulation: its bound is wanted
    const int edge_hi = g.edge + g.n_windows * g.window;   // first bin behind the last evaluated window
    const int n_left = (g.edge + g.piece - 1) / g.piece, n_right = (g.n - edge_hi + g.piece - 1) / g.piece;
    const int n_items = g.n_windows + ((g.do_bound && complete) ? n_left + n_right : 0);
    const float *__restrict__ rows = psd + ((size_t)band * g.stride + f_begin) * g.n;
    for (int item = wave; item < n_items; item += n_waves) {  // (wave-uniform)
        int b0, len, w = -1;
        if (item < g.n_windows) {
            w = item;
            b0 = g.edge + w * g.window;
            len = g.window;
        } else if (item < g.n_windows + n_left) {
            b0 = (item - g.n_windows) * g.piece;
            len = min(g.piece, g.edge - b0);
        } else {
            b0 = edge_hi + (item - g.n_windows - n_left) * g.piece;
            len = min(g.piece, g.n - b0);
        }
        const bool bound = g.do_bound && complete;
        unsigned units[JMAX], special[JMAX];
#pragma unroll
        for (int j = 0; j < JMAX; j++)
            units[j] = special[j] = 0u;
        // lane's bins: b0 + lane + 64 j, j < JMAX, valid while < b0 + len (an invalid one re-reads the lane's first bin
        // and is masked out of every sum)
        const unsigned off0 = (unsigned)(b0 + (lane < len ? lane : 0));
        float v[SDR_SCAN_AHEAD + 1][JMAX];
        auto fetch = [&](int f, int k) {
            const float *__restrict__ row = rows + (size_t)(f - f_begin) * g.n;
#pragma unroll
            for (int j = 0; j < JMAX; j++)
                v[k][j] = (lane + 64 * j < len) ? __builtin_nontemporal_load(row + off0 + 64 * j) : 0.0f;
        };
        // (the frame loop is unrolled over the ring of SDR_SCAN_AHEAD + 1 register sets)
        constexpr int RING = SDR_SCAN_AHEAD + 1;
#pragma unroll
        for (int k = 0; k < SDR_SCAN_AHEAD; k++)
            if (f_begin + k < f_end)
                fetch(f_begin + k, k);
        for (int f0 = f_begin; f0 < f_end; f0 += RING) {
#pragma unroll
            for (int k = 0; k < RING; k++) {
                const int f = f0 + k;
                if (f >= f_end)
                    break;
                if (f + SDR_SCAN_AHEAD < f_end)
                    fetch(f + SDR_SCAN_AHEAD, (k + SDR_SCAN_AHEAD) % RING);
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int j = 0; j < JMAX; j++) {
                    const float x = v[k][j];
                    if (w >= 0) {  // (an invalid slot holds 0: it adds nothing)
                        const double xd = (double)x;
                        s1 += xd;
                        s2 += xd * xd;  // (exact: a float32 squared has 48 bits)
                    }
                    if (bound && lane + 64 * j < len) {
                        const unsigned hw = __float_as_uint(x) >> 16;  // gomath::cum_bound_units / cum_bound_special
                        units[j] += (hw < 128u ? 128u : hw) + 1u;
                        special[j] |= hw + 0x8080u;  // bit 16 set  <=>  hw >= 0x7f80: infinity, NaN or a sign bit
                    }
                }
                if (w >= 0) {
                    s1 = wave_sum_to_lane63(s1);
                    s2 = wave_sum_to_lane63(s2);
                    if (lane == 63) {
                        double *o = wsum + ((size_t)band * g.stride + f) * (2 * noise::kMaxWindows) + 2 * w;
                        o[0] = s1;
                        o[1] = s2;
                    }
                }
            }
        }
        if (bound) {
#pragma unroll
            for (int j = 0; j < JMAX; j++) {
                if (lane + 64 * j < len) {
                    const int bin = b0 + lane + 64 * j;
                    const bool sp = (special[j] >> 16) != 0u;
                    // slot 0 continues the cumulation carried in from the previous batch, and that carry is produced on
                    // another stream (k_cumulate, behind this batch's predecessor): the raw unit count goes out instead
                    // (all ones: a special value in the column) and k_bound_slot0 adds the carry where it is known
                    cum_out[((size_t)band * g.max_chunks + slot) * g.n + bin] =
                        slot == 0 ? __uint_as_float(sp ? 0xffffffffu : units[j]) : gomath::cum_bound(0.0, units[j], f_len, a128, per_frame, sp);
                }
            }
        }
    }
}

// k_noise_finish - one thread per frame: the reference's minimum mean, variance and rolling-mean inputs from the scan's
// sums where they can be had for certain (noise_cert.h certify), the frame's number onto the list of the others.
__global__ __launch_bounds__(256) void k_noise_finish(const float *__restrict__ psd, const double *__restrict__ wsum,
                                                      sdr_frame_rec *__restrict__ recs, noise::Geom g, int n_frames, int stride,
                                                      unsigned *__restrict__ exact_list, int force_exact)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x, band = blockIdx.y;
    if (f >= n_frames)
        return;
    const size_t frame = (size_t)band * stride + f;
    const float *__restrict__ row = psd + frame * g.n;
    const double *s = wsum + frame * (2 * noise::kMaxWindows);
    double s1[noise::kMaxWindows], s2[noise::kMaxWindows];
#pragma unroll
    for (int w = 0; w < noise::kMaxWindows; w++) {
        s1[w] = w < g.n_windows ? s[2 * w] : 0.0;
        s2[w] = w < g.n_windows ? s[2 * w + 1] : 0.0;
    }
    const noise::Result r = noise::certify(s1, s2, g, [row](int i) { return (double)row[i]; });
    sdr_frame_rec rec;
    rec.min_mean = r.min_mean;
    rec.variance = r.variance;
    rec.dev_in = r.dev_in;
    rec.nf_in = r.nf_in;
    rec.noise_dev = rec.noise_floor = rec.peak_thr = rec.listen_thr = 0;
    rec.pad = 0;
    recs[frame] = rec;
    // (force_exact: 1 = every frame, k > 1 = every k-th - tests and sdr_read_frame_records' exact variances)
    if (!r.ok || force_exact == 1 || (force_exact > 1 && f % force_exact == 0)) {
        const unsigned at = atomicAdd(exact_list, 1u);
        exact_list[1 + at] = (unsigned)frame;
    }
}

// k_noise_exact_list - the literal FindNoiseFloor (noise_cert.h exact_frame: the oracle's loops) for the listed frames,
// one wave each: the window sums are independent chains (a lane each), the rest runs on lane 0.  Overwrites the four
// FindNoiseFloor fields of the frame's record; the thresholds kernel runs behind it.
// (First version: a wave per frame straight from global memory, lane 0 walking 12 000 dependent loads and additions -
// a millisecond per flagged frame, a quarter of a millisecond per batch on average, on the stream every threshold waits
// for.  Now a workgroup per frame: the row is staged in LDS once, the ten window chains run side by side from there, and
// the variance chain's terms - fl(fl(x - mean)^2), each its own rounding, any thread can form them - are produced chunk
// by chunk into an LDS double buffer by three waves while thread 0 adds the previous chunk in order: the chain runs at the
// pace of a dependent float64 add with its operand already on chip, 45 us for 12 000 terms.)
constexpr int kExactThreads = 256, kExactChunk = 2048;
__global__ __launch_bounds__(kExactThreads) void k_noise_exact_list(const float *__restrict__ psd, sdr_frame_rec *__restrict__ recs, noise::Geom g,
                                                                    const unsigned *__restrict__ exact_list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *row = reinterpret_cast<float *>(smem);                                  // [n]
    double *terms = reinterpret_cast<double *>(smem + (size_t)g.n * sizeof(float));  // [2][kExactChunk]
    __shared__ double s_sums[noise::kMaxWindows];
    __shared__ noise::Selection s_sel;
    const unsigned count = exact_list[0];
    const int tid = threadIdx.x;
    for (unsigned k = blockIdx.x; k < count; k += gridDim.x) {  // (workgroup-uniform)
        const size_t frame = exact_list[1 + k];
        const float *__restrict__ src = psd + frame * g.n;
        for (int i = tid; i < g.n; i += kExactThreads)
            row[i] = src[i];
        __syncthreads();
        auto x_at = [row](int i) { return (double)row[i]; };
        if (tid < g.n_windows)
            s_sums[tid] = noise::window_sum(g, x_at, tid);
        __syncthreads();
        if (tid == 0)
            s_sel = noise::select_window(g, x_at(0), s_sums);
        __syncthreads();
        const noise::Selection sel = s_sel;
        const int total = noise::result_to(g, sel.window) - g.edge + 1, n_chunks = (total + kExactChunk - 1) / kExactChunk;
        auto produce = [&](int c) {  // chunk c of the terms, by the threads of waves 1-3
            double *dst = terms + (size_t)(c & 1) * kExactChunk;
            const int base = c * kExactChunk, len = min(kExactChunk, total - base);
            for (int i = tid - 64; i < len; i += kExactThreads - 64)
                dst[i] = noise::variance_term(x_at(g.edge + base + i), sel.result_mean);
        };
        if (tid >= 64)
            produce(0);
        __syncthreads();
        double sum = 0;
        for (int c = 0; c < n_chunks; c++) {
            if (tid >= 64 && c + 1 < n_chunks)
                produce(c + 1);
            if (tid == 0) {
                const double *src_t = terms + (size_t)(c & 1) * kExactChunk;
                const int len = min(kExactChunk, total - c * kExactChunk);
                for (int i = 0; i < len; i++)
                    sum += src_t[i];  // :246-247, in order
            }
            __syncthreads();
        }
        if (tid == 0) {
            const noise::Result r = noise::finish_frame(g, sel, sum);
            recs[frame].min_mean = r.min_mean;
            recs[frame].variance = r.variance;
            recs[frame].dev_in = r.dev_in;
            recs[frame].nf_in = r.nf_in;
        }
        __syncthreads();  // (the next frame's staging overwrites the row)
    }
}

__global__ void k_noise_list_reset(unsigned *exact_list) { exact_list[0] = 0u; }

// k_noise_exact_check - the literal FindNoiseFloor for EVERY frame of one band's batch, a wave per frame (reads that drain
// the pipeline: sdr_read_frame_records).  The record's variance - which the hot path only brackets: nothing consumes its
// float64 bits - is replaced by the reference's, and the three values the hot path did produce for consumption (min_mean,
// dev_in, nf_in) are compared with the literal ones bit for bit: `mismatches` counts the frames where the certification
// let a wrong value through.  It must stay zero; every parity test that reads frame records checks it.
__global__ __launch_bounds__(64) void k_noise_exact_check(const float *__restrict__ psd, sdr_frame_rec *__restrict__ recs, noise::Geom g,
                                                          int n_frames, unsigned *__restrict__ mismatches)
{
    const int lane = threadIdx.x;
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float *__restrict__ row = psd + (size_t)f * g.n;
        auto x_at = [row](int i) { return (double)row[i]; };
        double mine = 0.0;
        if (lane < g.n_windows)
            mine = noise::window_sum(g, x_at, lane);
        double sums[noise::kMaxWindows];
#pragma unroll
        for (int w = 0; w < noise::kMaxWindows; w++)
            sums[w] = __shfl(mine, w);
        if (lane == 0) {
            const noise::Result r = noise::exact_frame(g, x_at, sums);
            sdr_frame_rec &rec = recs[f];
            if (__float_as_uint(rec.min_mean) != __float_as_uint(r.min_mean) || __float_as_uint(rec.dev_in) != __float_as_uint(r.dev_in) ||
                __float_as_uint(rec.nf_in) != __float_as_uint(r.nf_in))
                atomicAdd(mismatches, 1u);
            rec.variance = r.variance;
        }
    }
}

hipError_t launch_noise_exact_check(const float *psd_band, sdr_frame_rec *recs_band, NoiseGeom ng, int n_frames, unsigned *mismatches,
                                    hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    noise::Geom g{ng.n, ng.edge, ng.window, ng.n_windows, ng.inv_n2};
    hipLaunchKernelGGL(k_noise_exact_check, dim3(n_frames < 4096 ? n_frames : 4096), dim3(64), 0, stream, psd_band, recs_band, g, n_frames,
                       mismatches);
    return hipGetLastError();
}

static int scan_jmax(int window)
{
    const int j = (window + 63) / 64;
    return j <= 2 ? 2 : j <= 5 ? 5 : j <= 10 ? 10 : j <= 19 ? 19 : 26;
}

// One batch's scan: S1 / S2 of every frame and window, and (do_bound) the bound of every cumulation the batch completes.
hipError_t launch_psd_scan(const float *psd, double *wsum, float *cum_out, const BatchCursor *cur, NoiseGeom ng, CumGeom cg, int n_slots,
                           int n_bands, bool do_bound, hipStream_t stream)
{
    if (cg.n_frames <= 0 || n_bands <= 0)
        return hipSuccess;
    ScanGeom g;
    g.n = ng.n;
    g.edge = ng.edge;
    g.window = ng.window;
    g.n_windows = ng.n_windows;
    g.stride = cg.stride;
    g.n_frames = cg.n_frames;
    g.count0 = cg.count0;
    g.max_chunks = cg.max_chunks;
    g.do_bound = do_bound ? 1 : 0;
    const int jmax = scan_jmax(ng.window);
    if (ng.window > 64 * jmax)
        return hipErrorInvalidValue;  // (a window of more than 1664 values: no block size up to 16384 has one)
    g.piece = 64 * jmax;
    const int edge_hi = g.edge + g.n_windows * g.window;
    const int n_items = g.n_windows + (do_bound ? (g.edge + g.piece - 1) / g.piece + (g.n - edge_hi + g.piece - 1) / g.piece : 0);
    const int waves = n_items < kScanMaxWaves ? n_items : kScanMaxWaves;
    double a128, per_frame;
    gomath::cum_bound_constants(g.n, &a128, &per_frame);
    // (graph mode: the grid must cover the slots of any cumulation phase - the kernel returns for slots beyond the batch)
    const dim3 grid(n_slots, n_bands), block(64 * waves);
    switch (jmax) {
    case 2: launch_kernel(k_psd_scan<2>, grid, block, 0, stream, psd, wsum, cum_out, cur, g, a128, per_frame); break;
    case 5: launch_kernel(k_psd_scan<5>, grid, block, 0, stream, psd, wsum, cum_out, cur, g, a128, per_frame); break;
    case 10: launch_kernel(k_psd_scan<10>, grid, block, 0, stream, psd, wsum, cum_out, cur, g, a128, per_frame); break;
    case 19: launch_kernel(k_psd_scan<19>, grid, block, 0, stream, psd, wsum, cum_out, cur, g, a128, per_frame); break;
    default: launch_kernel(k_psd_scan<26>, grid, block, 0, stream, psd, wsum, cum_out, cur, g, a128, per_frame); break;
    }
    return hipGetLastError();
}

// The frame records' FindNoiseFloor fields from the scan's sums (certified, or the literal loops for the flagged frames).
// exact_list: 1 + n_bands * stride words.  force_exact: see k_noise_finish.  (A stage event armed by the caller rides on
// the last launch.)
hipError_t launch_noise_finish(const float *psd, const double *wsum, sdr_frame_rec *recs, NoiseGeom ng, int n_frames, int n_bands,
                               int stride, unsigned *exact_list, int force_exact, hipStream_t stream)
{
    if (n_frames <= 0 || n_bands <= 0)
        return hipSuccess;
    noise::Geom g{ng.n, ng.edge, ng.window, ng.n_windows, ng.inv_n2};
    const hipEvent_t done = t_done_event;
    t_done_event = nullptr;
    hipLaunchKernelGGL(k_noise_list_reset, dim3(1), dim3(1), 0, stream, exact_list);
    hipLaunchKernelGGL(k_noise_finish, dim3((n_frames + 255) / 256, n_bands), dim3(256), 0, stream, psd, wsum, recs, g, n_frames, stride,
                       exact_list, force_exact);
    t_done_event = done;
    // a workgroup per flagged frame (they loop when there are more frames than workgroups: every frame flagged, in tests)
    const unsigned lds = (unsigned)((size_t)ng.n * sizeof(float) + 2u * kExactChunk * sizeof(double));
    static std::once_flag attr_once[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    if (dev < 0 || dev >= 64)
        return hipErrorInvalidDevice;
    hipError_t attr_err = hipSuccess;
    std::call_once(attr_once[dev], [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_noise_exact_list), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       16384 * (int)sizeof(float) + 2 * kExactChunk * (int)sizeof(double));
    });
    if (attr_err != hipSuccess)
        return attr_err;
    const int wgs = force_exact == 1 ? 1024 : 32;
    launch_kernel(k_noise_exact_list, dim3(wgs), dim3(kExactThreads), lds, stream, psd, recs, g, static_cast<const unsigned *>(exact_list));
    return hipGetLastError();
}

}  // namespace sdr
