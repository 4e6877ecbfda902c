// k_noise_scan.hip — ONE pass over a batch's psd rows for everything the tail needs from all of them, in ONE kernel:
//   * dsp.FindNoiseFloor (dsp/fft.go:215-252): per frame and window the order-free sums S1 = sum x, S2 = sum x^2, from which
//     the same workgroup takes the reference's minimum mean, variance and rolling-mean inputs exactly where they are
//     consumed (noise_cert.h: brackets around what the reference's ORDERED sums can be); a frame whose brackets straddle a
//     float32 rounding boundary or another window's mean - three in 10^5 - gets the literal loops, there and then;
//   * the cumulation (rx/receiver.go:404-407): the unit counts behind the upper bound of every cumulation the batch completes
//     (the psd words' top halves: gomath.h cum_bound_*), which k_bound_finish / k_cum_refine / k_find_peaks (k_peaks.hip)
//     turn into the exact peak list.
// Rounds 1-4 read the psd three times for this (window chains 73 % of it, variance chains up to 73 %, the bound all of it:
// 31 of a step's 181 CU-ms; a stage costs the CU time it HOLDS, and these hold it for bytes / what a CU's memory pipeline
// delivers) and ran 24 000 strictly ordered float64 additions per frame that nothing downstream can tell from any other
// order.  And one kernel, not four: the FFT's workgroups own whole CUs, so every launch of the tail waits for CUs to come
// free - a one-thread kernel took 14 us in the running pipeline - and the noise floor sits on the path every threshold,
// and with it every listener, waits for.
//
// Geometry.  A row is cut into SEGMENTS: the reference's windows (W values each) and pieces of the two edges (at most
// 64 JMAX values).  A WAVE owns one segment of one run of consecutive frames (a cumulation slot, or a share of one): lane l
// holds four neighbouring bins of every 256 (16-byte loads) and single bins of the segment's last < 256; per frame it adds its
// values into the frame's S1 / S2 (a lane's <= JMAX + 4 values in sequence, then a six-step ladder across the lanes:
// noise_cert.h kScanTerms) and into its bins' running unit counts.
// A workgroup = the segments of one run: 14 waves at N = 16384.  Behind a barrier its threads then finish one frame each
// (certify), and the workgroup as a whole walks the literal loops of the frames that were not accepted.
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
constexpr int kScanMaxFrames = SDR_CUMULATION_SIZE;  // frames a workgroup walks at most
constexpr int kExactChunk = 2048;                    // terms of the variance chain per LDS buffer
#if !defined(SDR_SCAN_AHEAD)
#define SDR_SCAN_AHEAD 2  // frames whose loads are in flight ahead of the one being added up
#endif
#if !defined(SDR_SCAN_AUX)
#define SDR_SCAN_AUX 2  // cache policy bits of the psd loads (2 = nt: the rows are not read again by this kernel)
#endif

using scan_rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned scan_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ scan_rsrc_t scan_make_rsrc(const void *base, unsigned bytes)
{
    // (inputs made provably wave-uniform, otherwise the descriptor is rebuilt per lane)
    const unsigned long long b = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes),
                                             0x00020000);
}

__device__ __forceinline__ void scan_slot_frames(int slot, int count0, int *begin, int *len)
{
    const int first_len = SDR_CUMULATION_SIZE - count0;
    *begin = slot == 0 ? 0 : first_len + (slot - 1) * SDR_CUMULATION_SIZE;
    *len = slot == 0 ? first_len : SDR_CUMULATION_SIZE;
}

// Sum of a float64 over the wave, in lane 63, by DPP moves (no LDS round trips: as twelve ds_bpermute pairs per frame the
// reduction WAS the kernel - 0.27 ms for a batch whose loads and additions need 0.07).  A shift / broadcast ladder; lanes a
// step does not write receive 0.0 (old = 0 under the row / bank masks), which adds nothing.  Six additions deep
// (noise_cert.h kScanTerms).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_moved(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, BANK_MASK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}
// a shift within the rows of sixteen, every lane written: lanes whose source lies outside the row read 0 (bound_ctrl), so
// the destination needs no initial value (as `old = 0` each move cost a v_mov of its own: 24 of a frame's 60 reduction
// instructions)
template <int CTRL>
__device__ __forceinline__ double dpp_row_shifted(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double x)
{
    x += dpp_row_shifted<0x111>(x);      // row_shr:1
    x += dpp_row_shifted<0x112>(x);      // row_shr:2
    x += dpp_row_shifted<0x114>(x);      // row_shr:4
    x += dpp_row_shifted<0x118>(x);      // row_shr:8: lane 15 of every row holds the row's sum (a prefix scan: lane l its first l + 1)
    x += dpp_moved<0x142, 0xa, 0xf>(x);  // row_bcast:15 into rows 1 and 3
    x += dpp_moved<0x143, 0xc, 0xf>(x);  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
    return x;
}

// LDS of a scan workgroup: the run's sums [frame][window][S1, S2], the list of frames for the literal loops, and - used
// by those only - a staged psd row and the double buffer of the variance chain's terms
struct ScanLds {
    double sums[kScanMaxFrames][2 * noise::kMaxWindows];
    double win_sums[noise::kMaxWindows];
    noise::Selection sel;
    int n_flagged;
    unsigned short flagged[kScanMaxFrames];
};
constexpr int scan_lds_bytes(int n) { return (int)((sizeof(ScanLds) + 15) / 16 * 16) + n * (int)sizeof(float) + 2 * kExactChunk * (int)sizeof(double); }

// The literal FindNoiseFloor (noise_cert.h: the oracle's loops) of one frame by a whole workgroup: the row is staged in
// LDS once, the ten window chains run side by side from there (a thread each), and the variance chain's terms -
// fl(fl(x - mean)^2), each its own rounding: any thread can form them - are produced chunk by chunk into an LDS double
// buffer by the other waves while thread 0 adds the previous chunk in order: the chain runs at the pace of a dependent
// float64 add with its operand on chip, 45 us for 12 000 terms.  (First version: a wave per frame straight from global
// memory, lane 0 walking 12 000 dependent loads - a millisecond per frame.)
__device__ void exact_frame_block(ScanLds &sh, float *row, double *terms, const float *__restrict__ src, sdr_frame_rec *rec, const noise::Geom &g)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < g.n; i += nt)
        row[i] = src[i];
    __syncthreads();
    auto x_at = [row](int i) { return (double)row[i]; };
    if (tid < g.n_windows)
        sh.win_sums[tid] = noise::window_sum(g, x_at, tid);
    __syncthreads();
    if (tid == 0)
        sh.sel = noise::select_window(g, x_at(0), sh.win_sums);
    __syncthreads();
    const noise::Selection sel = sh.sel;
    const int total = noise::result_to(g, sel.window) - g.edge + 1, n_chunks = (total + kExactChunk - 1) / kExactChunk;
    auto produce = [&](int c) {  // chunk c of the terms, by the threads of waves 1 ..
        double *dst = terms + (size_t)(c & 1) * kExactChunk;
        const int base = c * kExactChunk, len = min(kExactChunk, total - base);
        for (int i = tid - 64; i < len; i += nt - 64)
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
        rec->min_mean = r.min_mean;
        rec->variance = r.variance;
        rec->dev_in = r.dev_in;
        rec->nf_in = r.nf_in;
    }
    __syncthreads();  // (the next frame's staging overwrites the row)
}

template <int JMAX>
__global__ __launch_bounds__(64 * kScanMaxWaves) void k_psd_scan(const float *__restrict__ psd, sdr_frame_rec *__restrict__ recs,
                                                                  float *__restrict__ cum_out, float *__restrict__ cum_part,
                                                                  const BatchCursor *__restrict__ cur, ScanGeom g, double inv_n2,
                                                                  int force_exact)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ScanLds &sh = *reinterpret_cast<ScanLds *>(smem);
    int count0 = g.count0;
    if (cur)  // graph replay: this batch's cumulation phase comes from device memory
        count0 = cur->count0;
    // with bounds: blockIdx.x = slot * parts + part, a part = an equal share of the slot's frames (two parts where there are
    // too few slots for the chip: a workgroup walks its frames one after the other, 2.4 us each)
    const int slot = g.do_bound ? (int)blockIdx.x / g.parts : 0, part = g.do_bound ? (int)blockIdx.x % g.parts : 0, band = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n_waves = (int)blockDim.x >> 6;
    int f_begin, f_len;
    if (g.do_bound) {
        scan_slot_frames(slot, count0, &f_begin, &f_len);
    } else {
        // no bounds wanted (short batches: k_cumulate keeps every slot exact): the cumulation slots mean nothing here, a
        // workgroup takes g.fpw consecutive frames - a 2048-frame batch has 21 slots, and 21 workgroups walking a hundred
        // frames each took longer than the batch's FFT (config 3 at 2048 frames per batch: 105 GS/s instead of 160)
        f_begin = (int)blockIdx.x * g.fpw;
        f_len = g.fpw;
    }
    if (f_begin >= g.n_frames)
        return;  // (workgroup-uniform)
    const bool complete = f_begin + f_len <= g.n_frames;  // the batch completes this cumulation: its bound is wanted
    if (g.do_bound && g.parts > 1) {
        const int share = (f_len + g.parts - 1) / g.parts;
        f_begin += part * share;
        f_len = max(0, min(share, f_len - part * share));
    }
    const int f_end = min(f_begin + f_len, g.n_frames);
    const int edge_hi = g.edge + g.n_windows * g.window;  // first bin behind the last evaluated window
    const int n_left = (g.edge + g.piece - 1) / g.piece, n_right = (g.n - edge_hi + g.piece - 1) / g.piece;
    const int n_items = g.n_windows + ((g.do_bound && complete) ? n_left + n_right : 0);
    const float *__restrict__ rows = psd + ((size_t)band * g.stride + f_begin) * g.n;
    if (threadIdx.x == 0)
        sh.n_flagged = 0;
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
        // A lane's bins.  The segment's first 256 q4 values go by 16-BYTE loads - lane l holds bins b0 + 256 q + 4 l .. + 3 -
        // and the rest (fewer than 256) by dword loads, lane l bin b0 + 256 q4 + l + 64 r.  A CU's vector-memory pipe takes a
        // wave's load every 16 clocks or so whatever its width: nineteen dword loads per frame and window kept the kernel at
        // 27 GB/s per CU with the arithmetic units half idle.  All of them are BUFFER loads through descriptors that span
        // exactly the bytes they may touch (the 16-byte loads whole groups of 256 values, the dword loads the remainder): a
        // load beyond reads 0.0 without a compare, an exec mask or a branch, and the address is a scalar base plus one
        // per-lane offset.  (As global loads under per-lane predicates the loop was 38 instructions per value, five of them
        // the arithmetic.)  What such a value adds is harmless: nothing to S1 / S2, something to a unit count never stored.
        constexpr int Q4MAX = JMAX / 4, R1MAX = JMAX < 4 ? JMAX : 4, Q4A = Q4MAX ? Q4MAX : 1;
        const int q4 = len >> 8, rem = len - (q4 << 8);  // (wave-uniform; q4 <= Q4MAX as len <= 64 JMAX)
        unsigned u4[Q4A][4], u1[R1MAX], hw_max = 0u;
#pragma unroll
        for (int q = 0; q < Q4A; q++)
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++)
                u4[q][c4] = 0u;
#pragma unroll
        for (int r = 0; r < R1MAX; r++)
            u1[r] = 0u;
        const unsigned lane4 = (unsigned)lane * 4u, lane16 = (unsigned)lane * 16u;
        scan_u32x4 v4[SDR_SCAN_AHEAD + 1][Q4A];
        unsigned v1[SDR_SCAN_AHEAD + 1][R1MAX];
        auto fetch = [&](int f, int k) {
            const float *seg = rows + (size_t)(f - f_begin) * g.n + b0;
            const scan_rsrc_t wide = scan_make_rsrc(seg, (unsigned)q4 * 1024u), narrow = scan_make_rsrc(seg + (q4 << 8), (unsigned)rem * 4u);
#pragma unroll
            for (int q = 0; q < Q4MAX; q++)
                v4[k][q] = __builtin_amdgcn_raw_buffer_load_b128(wide, lane16 + 1024u * (unsigned)q, 0, SDR_SCAN_AUX);
#pragma unroll
            for (int r = 0; r < R1MAX; r++)
                v1[k][r] = __builtin_amdgcn_raw_buffer_load_b32(narrow, lane4 + 256u * (unsigned)r, 0, SDR_SCAN_AUX);
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
                if (w >= 0) {  // (wave-uniform)
                    double s1 = 0.0, s2 = 0.0;
                    auto add = [&](unsigned bits) {
                        const double xd = (double)__uint_as_float(bits);
                        s1 += xd;
                        s2 = __builtin_fma(xd, xd, s2);  // (a float32 squared has 48 bits: the product is exact either way)
                    };
#pragma unroll
                    for (int q = 0; q < Q4MAX; q++) {
                        add(v4[k][q].x);
                        add(v4[k][q].y);
                        add(v4[k][q].z);
                        add(v4[k][q].w);
                    }
#pragma unroll
                    for (int r = 0; r < R1MAX; r++)
                        add(v1[k][r]);
                    s1 = wave_sum_to_lane63(s1);
                    s2 = wave_sum_to_lane63(s2);
                    if (lane == 63) {
                        sh.sums[f - f_begin][2 * w] = s1;
                        sh.sums[f - f_begin][2 * w + 1] = s2;
                    }
                }
                if (bound) {  // (wave-uniform)
                    auto count = [&](unsigned bits, unsigned &units) {
                        const unsigned hw = bits >> 16;  // gomath::cum_bound_units without its + 1 (added below, per frame)
                        units += hw < 128u ? 128u : hw;
                        hw_max = hw > hw_max ? hw : hw_max;  // gomath::cum_bound_special: hw >= 0x7f80 - infinity, NaN or a sign bit
                    };
#pragma unroll
                    for (int q = 0; q < Q4MAX; q++) {
                        count(v4[k][q].x, u4[q][0]);
                        count(v4[k][q].y, u4[q][1]);
                        count(v4[k][q].z, u4[q][2]);
                        count(v4[k][q].w, u4[q][3]);
                    }
#pragma unroll
                    for (int r = 0; r < R1MAX; r++)
                        count(v1[k][r], u1[r]);
                }
            }
        }
        if (bound) {
            // a special value anywhere in a LANE's columns marks all of them (one maximum per value instead of a flag word per
            // column: the bound of such a column becomes +infinity, which is a bound, and its exact evaluation decides - a
            // psd row that holds an infinity or a NaN is not a case to be fast in)
            const bool sp = hw_max >= 0x7f80u;
            const unsigned n_run = (unsigned)(f_end - f_begin);  // units(psd) = max(hw, 128) + 1: the + 1 of every frame
            // The raw unit count goes out (all ones: a special value in the column), part 0's into the slot's row, part 1's
            // into the second buffer: k_bound_finish (k_peaks.hip) adds the parts and - slot 0 continues the cumulation
            // carried in from the previous batch, and that carry is produced on the peaks stream - the carry, and forms
            // the bound there.
            float *__restrict__ out = (part == 0 ? cum_out : cum_part) + ((size_t)band * g.max_chunks + slot) * g.n + b0;
#pragma unroll
            for (int q = 0; q < Q4MAX; q++) {
                if (q < q4) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; c4++)
                        out[256 * q + 4 * lane + c4] = __uint_as_float(sp ? 0xffffffffu : u4[q][c4] + n_run);
                }
            }
#pragma unroll
            for (int r = 0; r < R1MAX; r++)
                if (lane + 64 * r < rem)
                    out[(q4 << 8) + lane + 64 * r] = __uint_as_float(sp ? 0xffffffffu : u1[r] + n_run);
        }
    }
    __syncthreads();
    // one frame per thread: the reference's values where they can be had for certain, the frame onto the list otherwise
    // (force_exact, tests: 1 = every frame, k > 1 = every k-th takes the literal loops)
    const noise::Geom ng{g.n, g.edge, g.window, g.n_windows, inv_n2};
    for (int i = threadIdx.x; i < f_end - f_begin; i += blockDim.x) {
        const int f = f_begin + i;
        const size_t frame = (size_t)band * g.stride + f;
        const float *__restrict__ row = psd + frame * g.n;
        double s1[noise::kMaxWindows], s2[noise::kMaxWindows];
#pragma unroll
        for (int w = 0; w < noise::kMaxWindows; w++) {
            s1[w] = w < g.n_windows ? sh.sums[i][2 * w] : 0.0;
            s2[w] = w < g.n_windows ? sh.sums[i][2 * w + 1] : 0.0;
        }
        const noise::Result r = noise::certify(s1, s2, ng, [row](int k) { return (double)row[k]; });
        sdr_frame_rec rec;
        rec.min_mean = r.min_mean;
        rec.variance = r.variance;
        rec.dev_in = r.dev_in;
        rec.nf_in = r.nf_in;
        rec.noise_dev = rec.noise_floor = rec.peak_thr = rec.listen_thr = 0;
        rec.pad = r.ok ? 0.0f : (float)r.why;  // (diagnostic: why the frame went to the literal loops, noise_cert.h)
        recs[frame] = rec;
        if (!r.ok || force_exact == 1 || (force_exact > 1 && f % force_exact == 0))
            sh.flagged[atomicAdd(&sh.n_flagged, 1)] = (unsigned short)i;
    }
    __syncthreads();
    const int n_flagged = sh.n_flagged;  // (workgroup-uniform)
    if (n_flagged == 0)
        return;
    float *row_lds = reinterpret_cast<float *>(smem + (sizeof(ScanLds) + 15) / 16 * 16);
    double *terms = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(row_lds) + (size_t)g.n * sizeof(float));
    for (int k = 0; k < n_flagged; k++) {
        const size_t frame = (size_t)band * g.stride + f_begin + sh.flagged[k];
        exact_frame_block(sh, row_lds, terms, psd + frame * g.n, recs + frame, ng);
    }
}

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

// parts a slot's frames are dealt over (k_peaks.hip's k_bound_finish must add as many)
int scan_parts(int n_slots, int n_bands)
{
    static const int forced = getenv("SDR_SCAN_PARTS") ? atoi(getenv("SDR_SCAN_PARTS")) : 0;  // (experiments: 1 or 2)
    if (forced == 1 || forced == 2)
        return forced;
    // (two parts while the slots alone are fewer than a quarter of the CUs; from there on whole slots: with 16-byte loads a
    // workgroup walks a frame in 1.6 us, and 83 fat workgroups hold less CU time than 166 - config 3: 205.5 -> 209.7 GS/s)
    return (long)n_slots * n_bands < 64 ? 2 : 1;
}

static int scan_jmax(int window)
{
    const int j = (window + 63) / 64;
    return j <= 2 ? 2 : j <= 5 ? 5 : j <= 10 ? 10 : j <= 19 ? 19 : 26;
}

// One batch's noise floor: the FindNoiseFloor fields of every frame's record and (do_bound) the unit counts of every
// cumulation the batch completes.  force_exact: see k_psd_scan (tests).  (A stage event armed by the caller rides on
// the launch.)
hipError_t launch_psd_scan(const float *psd, sdr_frame_rec *recs, float *cum_out, float *cum_part, const BatchCursor *cur, NoiseGeom ng,
                           CumGeom cg, int n_slots, int n_bands, bool do_bound, int force_exact, hipStream_t stream)
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
    g.parts = scan_parts(n_slots, n_bands);
    // without bounds: as many frames per workgroup as leave the chip about two workgroups per CU, at least eight
    g.fpw = 0;
    if (!do_bound) {
        const long total = (long)cg.n_frames * n_bands;
        g.fpw = (int)(total / 512);
        g.fpw = g.fpw < 8 ? 8 : (g.fpw > kScanMaxFrames ? kScanMaxFrames : g.fpw);
    }
    const unsigned lds = (unsigned)scan_lds_bytes(ng.n);
    // more than 64 KB of dynamic LDS needs the attribute, once per device and instantiation
    static std::once_flag attr_once[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    if (dev < 0 || dev >= 64)
        return hipErrorInvalidDevice;
    hipError_t attr_err = hipSuccess;
    std::call_once(attr_once[dev], [&] {
        const int max_lds = scan_lds_bytes(16384);
        for (const void *k : {reinterpret_cast<const void *>(&k_psd_scan<2>), reinterpret_cast<const void *>(&k_psd_scan<5>),
                              reinterpret_cast<const void *>(&k_psd_scan<10>), reinterpret_cast<const void *>(&k_psd_scan<19>),
                              reinterpret_cast<const void *>(&k_psd_scan<26>)}) {
            const hipError_t ae = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
            if (ae != hipSuccess)
                attr_err = ae;
        }
    });
    if (attr_err != hipSuccess)
        return attr_err;
    // (graph mode: the grid must cover the slots of any cumulation phase - the kernel returns for slots beyond the batch)
    const dim3 grid(do_bound ? n_slots * g.parts : (cg.n_frames + g.fpw - 1) / g.fpw, n_bands), block(64 * waves);
    switch (jmax) {
    case 2: launch_kernel(k_psd_scan<2>, grid, block, lds, stream, psd, recs, cum_out, cum_part, cur, g, ng.inv_n2, force_exact); break;
    case 5: launch_kernel(k_psd_scan<5>, grid, block, lds, stream, psd, recs, cum_out, cum_part, cur, g, ng.inv_n2, force_exact); break;
    case 10: launch_kernel(k_psd_scan<10>, grid, block, lds, stream, psd, recs, cum_out, cum_part, cur, g, ng.inv_n2, force_exact); break;
    case 19: launch_kernel(k_psd_scan<19>, grid, block, lds, stream, psd, recs, cum_out, cum_part, cur, g, ng.inv_n2, force_exact); break;
    default: launch_kernel(k_psd_scan<26>, grid, block, lds, stream, psd, recs, cum_out, cum_part, cur, g, ng.inv_n2, force_exact); break;
    }
    return hipGetLastError();
}

}  // namespace sdr
