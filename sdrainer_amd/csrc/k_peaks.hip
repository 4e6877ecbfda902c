// k_peaks.hip — dB projection + ordered float32 cumulation over 100 frames, and the run-length peak scan of each
// completed cumulation.  Compiled with -ffp-contract=off (see gomath.h).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "fft_f64.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

// ---------------------------------------------------------------------------------------------
// The cumulation: cumulation[i] += spectrum[i] (rx/receiver.go:404-407), a float32 sum in frame order over 100 frames,
// where spectrum[i] = MagnitudeIndB(...) + dBmShift (dsp/fft.go:79-81, rx/receiver.go:377) is evaluated from the float32
// psd the FFT kernel stored (the projection is a pure function of that value).
//
// EXACT WHERE IT IS CONSUMED.  The only consumer of a completed cumulation on the hot path is FindPeaks (dsp/fft.go:
// 254-285): which bins have cumulation / 100 > threshold, and the values inside those runs and right beside their maxima.
// A noise bin sits 15 dB under that threshold.  Rounds 1-3 evaluated the certified logarithm for every bin of every
// frame - 33.5 M of them per 2048-frame batch, 15 % of all CU time on the chip - to learn, for 98 % of the bins, that
// they are not above the threshold.  Now:
//   k_cum_bound   an UPPER BOUND of every completed cumulation from the psd words' top halves alone - for psd = m 2^E,
//                 log2(psd) <= E + (m - 1) + 0.0861 and the float32 bits >> 16 ARE (E + 127) 128 + floor(128 (m - 1)):
//                 one shift, one max, one add per value, 16-byte loads, nothing but the psd stream;
//   k_cum_refine  compares bound / 100 with the threshold exactly as FindPeaks compares the cumulation (monotone
//                 float32 operations: a bin whose BOUND is not above is not above) and writes the literal ordered float32
//                 sum of certified dB values over the bound of the bins that are - and of their neighbours, which
//                 PeakCenterCorrection reads; k_find_peaks scans the row with those in place: the same runs, maxima and
//                 centre values, bit for bit;
//   k_cumulate    the exact kernel of rounds 1-3: the cumulation still open at the end of a batch (its carry into the
//                 next batch is exact for every bin: which bins will matter is not known yet), and whole rows on demand
//                 (sdr_read_cumulation, the scope tap).
// The bound (proof in DESIGN.md section 5): each term is fl32(fl32(10 log10(20 psd / N^2)) + 120) <= A U + K + 120 + 2e-5
// with U the linear bound of log2 above, A = 10 log10 2, K = 10 log10 20 - 20 log10 N; the ordered float32 sum of 100
// such terms (|partial sums| < 2^16 for any finite psd) exceeds the real sum by less than 0.2; the bound is evaluated in
// float64 with 0.05 to spare before it is rounded to float32.  Zero / subnormal psd count as 2^-126 (their dB is lower), infinities and
// NaNs give a huge bound (candidates: the exact evaluation decides, as the reference's would).
// ---------------------------------------------------------------------------------------------
// the literal Go algorithm, out of line: it is rare, and inlined its float64 temporaries set the kernel's register count
__device__ __attribute__((noinline)) float db_slow(float psd, double inv_n2) { return gomath::psd_value_in_db(psd, inv_n2); }

__device__ __forceinline__ float spectrum_value(float psd, gomath::DbTables t, double inv_n2)
{
    float db;
    if (!gomath::psd_value_in_db_fast(psd, t, &db))
        db = db_slow(psd, inv_n2);
    return db + (float)SDR_DBM_SHIFT;
}

#if !defined(SDR_CUM_THREADS)
#define SDR_CUM_THREADS 256
#endif
#if !defined(SDR_CUM_U)
#define SDR_CUM_U 4
#endif
#if !defined(SDR_REFINE_U)
#define SDR_REFINE_U 16
#endif
#if !defined(SDR_REFINE_Q)
#define SDR_REFINE_Q 4  // lanes a refined column's frames are dealt over (1: one lane per column, round 4's loop)
#endif
#if !defined(SDR_CUM_VGPR)
#define SDR_CUM_VGPR 32
#endif

// acc + the ordered float32 sum of spectrum(psd[base + off + k n]) for k = 0 .. count-1 (one column of the psd array,
// frame order): the certified table shortcut of gomath.h settles all but about three values in 10^5, the rest take
// the literal Go algorithm in a rare, divergent branch.  U loads are in flight while the previous U values are projected
// (independent float64 chains) and added in order.
template <int U = SDR_CUM_U>
__device__ __forceinline__ float cum_exact_column(const float *__restrict__ base, unsigned off, unsigned n, int count, float acc,
                                                  gomath::DbTables tab, double inv_n2)
{
    int f = 0;
    float v[U], nv[U];
    if (U <= count) {
#pragma unroll
        for (int k = 0; k < U; k++)
            nv[k] = __builtin_nontemporal_load(base + (off + (unsigned)k * n));
    }
    for (; f + U <= count; f += U) {
#pragma unroll
        for (int k = 0; k < U; k++)
            v[k] = nv[k];
        off += U * n;
        if (f + 2 * U <= count) {
#pragma unroll
            for (int k = 0; k < U; k++)
                nv[k] = __builtin_nontemporal_load(base + (off + (unsigned)k * n));
        }
        float db[U];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < U; k++)
            bad |= !gomath::psd_value_in_db_fast(v[k], tab, &db[k]);
        if (__builtin_amdgcn_ballot_w64(bad)) {
#pragma unroll 1
            for (int k = 0; k < U; k++) {
                float t;
                if (!gomath::psd_value_in_db_fast(v[k], tab, &t))
                    db[k] = db_slow(v[k], inv_n2);
            }
        }
#pragma unroll
        for (int k = 0; k < U; k++)
            acc += db[k] + (float)SDR_DBM_SHIFT;  // MagnitudeIndB + dBmShift (float32 add), then the ordered sum
    }
    for (; f < count; f++, off += n)
        acc += spectrum_value(__builtin_nontemporal_load(base + off), tab, inv_n2);
    return acc;
}

// frames [begin, begin + len) of slot `slot` of a batch that starts at cumulationCount count0: slot 0 takes the
// (100 - count0) frames that complete the cumulation carried in, later slots 100 each
__device__ __forceinline__ void cum_slot_frames(int slot, int count0, int *begin, int *len)
{
    const int first_len = SDR_CUMULATION_SIZE - count0;
    *begin = slot == 0 ? 0 : first_len + (slot - 1) * SDR_CUMULATION_SIZE;
    *len = slot == 0 ? first_len : SDR_CUMULATION_SIZE;
}

// k_cumulate - the exact cumulation, one thread per bin, lanes on neighbouring bins (coalesced 256-byte rows).
//   only_open = 1: just the cumulation the batch leaves open (the carry into the next batch); grid.y = 1
//   only_open = 0: slot = slot_base + blockIdx.y, complete slots to cum_out (or to row_out: sdr_read_cumulation, one band)
// 256-thread workgroups, 20 KB of LDS (the tables), at most SDR_CUM_VGPR VGPRs.
__global__ __launch_bounds__(SDR_CUM_THREADS) __attribute__((amdgpu_num_vgpr(SDR_CUM_VGPR))) void k_cumulate(const float *__restrict__ psd, const void *__restrict__ db_tab,
                                                   float *__restrict__ carry0, float *__restrict__ carry1, int carry_in_arg,
                                                   float *__restrict__ cum_out, float *__restrict__ row_out, const BatchCursor *__restrict__ cur, CumGeom g,
                                                   int only_open, int slot_base, double inv_n2)
{
    int carry_sel = carry_in_arg;
    if (cur) {  // graph replay: this batch's cumulation phase comes from device memory
        g.count0 = cur->count0;
        carry_sel = cur->carry_in;
    }
    const int first = SDR_CUMULATION_SIZE - g.count0;
    const int slots = g.n_frames <= first ? 1 : 1 + (g.n_frames - first + SDR_CUMULATION_SIZE - 1) / SDR_CUMULATION_SIZE;
    const int slot = only_open ? slots - 1 : slot_base + (int)blockIdx.y;
    if (slot >= slots)
        return;
    int begin, len;
    cum_slot_frames(slot, g.count0, &begin, &len);
    const bool complete = (begin + len) <= g.n_frames;
    if (only_open && complete)
        return;  // the batch ends on a boundary: nothing is carried
    const float *__restrict__ carry_in = carry_sel ? carry1 : carry0;
    float *__restrict__ carry_out = carry_sel ? carry0 : carry1;
    __shared__ __attribute__((aligned(16))) unsigned char s_tab[gomath::kDbTabBytes];
    {
        const uint4 *src = static_cast<const uint4 *>(db_tab);
        uint4 *dst = reinterpret_cast<uint4 *>(s_tab);
        for (int i = threadIdx.x; i < gomath::kDbTabBytes / 16; i += blockDim.x)
            dst[i] = src[i];
    }
    __syncthreads();
    const gomath::DbTables tab = gomath::db_tables(s_tab);
    const int bin = blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= g.n)
        return;
    const int band = blockIdx.z;
    const int end = min(begin + len, g.n_frames);
    float acc = 0.f;
    if (slot == 0 && g.count0 > 0)
        acc = carry_in[(size_t)band * g.n + bin];
    // (wave-uniform base in SGPRs + one 32-bit per-lane offset)
    const float *__restrict__ base = psd + (size_t)band * g.stride * g.n;
    acc = cum_exact_column(base, (unsigned)begin * (unsigned)g.n + (unsigned)bin, (unsigned)g.n, end - begin, acc, tab, inv_n2);
    if (row_out)
        row_out[bin] = acc;
    else if (complete)
        cum_out[((size_t)band * g.max_chunks + slot) * g.n + bin] = acc;  // completed chunk index == slot
    else
        carry_out[(size_t)band * g.n + bin] = acc;
}

// k_cum_bound - an upper bound of every cumulation the batch completes (see the head of this section).  The kernel is
// nothing but the psd stream - a shift, a max and an add per value - so what it costs the pipeline is the CU time it
// holds while the stream goes by, and k_fft_psd needs whole CUs: spread over the chip the way a grid of small workgroups
// is, it held all 256 CUs for as long as the exact kernel did (0.10 ms per 8192 frames, round 4's first version).  So it
// runs as FEW, FAT, PERSISTENT workgroups: 1024 threads x four neighbouring bins (16-byte loads, a 16 KB row segment per
// wave-instruction round) x U frames in flight = 128 KB per workgroup on its way, one workgroup per CU (its LDS
// reservation keeps a second one and the FFT's off that CU, and the dispatcher deals workgroups to different CUs), about
// a quarter of the chip - each of them at what one CU's memory pipeline delivers - while the other CUs stay with the FFT.
// Work item = (band, slot, block of 4096 bins); workgroup w takes items w, w + grid, ...
// a128 = 10 log10(2) / 128; per_frame = what every frame adds beyond its top-half sum (gomath.h cum_bound_constants).
#if !defined(SDR_BOUND_U)
#define SDR_BOUND_U 8
#endif
#if !defined(SDR_BOUND_WGS)
#define SDR_BOUND_WGS 64
#endif
constexpr int kBoundThreads = 1024;
constexpr int kBoundLdsBytes = 96 * 1024;  // (reserved, not used: one workgroup per CU)
typedef unsigned bound_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBoundThreads) void k_cum_bound(const float *__restrict__ psd, const float *__restrict__ carry0, const float *__restrict__ carry1,
                                                             int carry_in_arg, float *__restrict__ cum_out, const BatchCursor *__restrict__ cur, CumGeom g,
                                                             int n_slots, int n_bands, double a128, double per_frame)
{
    int carry_sel = carry_in_arg;
    if (cur) {
        g.count0 = cur->count0;
        carry_sel = cur->carry_in;
    }
    const float *__restrict__ carry_in = carry_sel ? carry1 : carry0;
    const int blocks = (g.n / 4 + (int)blockDim.x - 1) / (int)blockDim.x;  // column blocks per row
    const int items = n_bands * n_slots * blocks;
    constexpr int U = SDR_BOUND_U;
    const bound_u4 floor_hw = {128u, 128u, 128u, 128u};  // zero / subnormal psd count as 2^-126
    for (int item = blockIdx.x; item < items; item += gridDim.x) {  // (workgroup-uniform)
        const int cb = item % blocks, slot = (item / blocks) % n_slots, band = item / (blocks * n_slots);
        int begin, len;
        cum_slot_frames(slot, g.count0, &begin, &len);
        if (begin + len > g.n_frames)
            continue;  // not completed by this batch (k_cumulate carries it) or beyond the batch
        const int bin = (cb * (int)blockDim.x + (int)threadIdx.x) * 4;
        if (bin >= g.n)
            continue;
        const unsigned n4 = (unsigned)g.n >> 2;
        const bound_u4 *__restrict__ col = reinterpret_cast<const bound_u4 *>(psd + (size_t)band * g.stride * g.n) + ((unsigned)begin * n4 + ((unsigned)bin >> 2));
        bound_u4 acc = {0u, 0u, 0u, 0u}, special = {0u, 0u, 0u, 0u};
        auto add = [&](bound_u4 v) {  // gomath::cum_bound_units / cum_bound_special, four bins at a time
            v = v >> 16;
            acc += __builtin_elementwise_max(v, floor_hw) + 1u;
            special |= v + 0x8080u;  // bit 16 set  <=>  v >= 0x7f80 (v < 2^16): infinity, NaN or a sign bit
        };
        int f = 0;
        bound_u4 nv[U];
        if (U <= len) {
#pragma unroll
            for (int k = 0; k < U; k++)
                nv[k] = __builtin_nontemporal_load(col + (unsigned)k * n4);
        }
        for (; f + U <= len; f += U) {
            bound_u4 v[U];
#pragma unroll
            for (int k = 0; k < U; k++)
                v[k] = nv[k];
            col += U * n4;
            if (f + 2 * U <= len) {
#pragma unroll
                for (int k = 0; k < U; k++)
                    nv[k] = __builtin_nontemporal_load(col + (unsigned)k * n4);
            }
#pragma unroll
            for (int k = 0; k < U; k++)
                add(v[k]);
        }
        for (; f < len; f++, col += n4)
            add(__builtin_nontemporal_load(col));
        float4 out;
        float *o = &out.x;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double c0 = (slot == 0 && g.count0 > 0) ? (double)carry_in[(size_t)band * g.n + bin + i] : 0.0;
            o[i] = gomath::cum_bound(c0, acc[i], len, a128, per_frame, (special[i] >> 16) != 0u);
        }
        *reinterpret_cast<float4 *>(cum_out + ((size_t)band * g.max_chunks + slot) * g.n + bin) = out;
    }
}

// spectrum row of one frame from its psd row (parity reads and the scope tap: sdr_read_spectrum), literal algorithm
__global__ __launch_bounds__(256) void k_spectrum_row(const float *__restrict__ psd, float *__restrict__ out, int n,
                                                      double inv_n2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = gomath::psd_value_in_db(psd[i], inv_n2) + (float)SDR_DBM_SHIFT;
}

// ---------------------------------------------------------------------------------------------
// k_cum_refine — the exact cumulation where FindPeaks will look (see the head of the cumulation section): for one
// completed cumulation and one span of 4096 bins, the bins whose bound / 100 is above the threshold (the comparison
// FindPeaks makes, dsp/fft.go:259-260: a bin whose BOUND is not above is not above) and their two neighbours (the
// centre correction reads them) are listed - ballot words, a popcount prefix - and each gets the carry plus the ordered
// float32 sum of the certified dB values of its frames (cum_exact_column), written over its bound in the row.  Dense
// lanes: entry k of the list goes to thread k (neighbouring bins of a run sit in neighbouring lanes).  Four workgroups
// per row rather than the row's peak scan doing it itself: a cumulation's refinement is a latency chain (a hundred
// scattered 64-byte sectors per candidate cluster) and at 2048 frames per batch the peaks stream had become the longest
// of the four with it (134 instead of 150 GS/s).  (Whole rows x 1024 threads for batches with many rows hold less CU
// time - 6.4 against 10.8 CU-ms per 8192-frame step, stages one after the other - and measured no better in the pipeline:
// config 3 160.5 against 162.4 GS/s, config 5's share 182.9 against 180.8.  One shape.)
// A neighbour span may be rewriting the halo bin this workgroup classifies while it reads it: it then sees either the
// bound or the exact value, and both classify every bin that is above as above - the list can only differ in bins that
// need not have been refined.
// ---------------------------------------------------------------------------------------------
constexpr int kRefineSpan = 4096, kRefineThreads = 256;  // (the <SPAN, THREADS> of short batches; long ones: whole rows x 1024)
__device__ __forceinline__ int peak_words(int n) { return n >> 6; }

// exclusive prefix of per-word bit counts over `words` words (one wave: each lane takes `per` consecutive words);
// offs[words] = the total, also returned in every lane
__device__ __forceinline__ int word_prefix(const unsigned long long *bits, int *offs, int words, int lane)
{
    const int per = (words + 63) >> 6;
    int local = 0;
    for (int k = 0; k < per; k++) {
        const int w = lane * per + k;
        if (w < words)
            local += __popcll(bits[w]);
    }
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o)
            incl += up;
    }
    int run = incl - local;
    for (int k = 0; k < per; k++) {
        const int w = lane * per + k;
        if (w < words) {
            offs[w] = run;
            run += __popcll(bits[w]);
        }
    }
    const int total = __shfl(incl, 63);
    if (lane == 63)
        offs[words] = total;
    return total;
}

template <int kRefineSpan, int kRefineThreads>
__global__ __launch_bounds__(kRefineThreads) void k_cum_refine(float *__restrict__ cum, const float *__restrict__ psd, const void *__restrict__ db_tab,
                                                               const float *__restrict__ carry0, const float *__restrict__ carry1, int carry_in_arg,
                                                               const sdr_frame_rec *__restrict__ recs, const BatchCursor *__restrict__ cur, PeakGeom g,
                                                               int n_frames, double inv_n2, const float *__restrict__ tap_wide,
                                                               const int32_t *__restrict__ tap_used, int n_tap, int tap_stride)
{
    constexpr int SW = kRefineSpan / 64;  // words per span
    __shared__ __attribute__((aligned(16))) unsigned char s_tab[gomath::kDbTabBytes];
    __shared__ unsigned long long s_flags[SW + 2];  // [0] / [SW + 1]: the neighbours' edge words
    __shared__ unsigned long long s_rflags[SW];
    __shared__ int s_offs[SW + 1];
    __shared__ unsigned short s_list[kRefineSpan];
    __shared__ unsigned short s_tapidx[kRefineSpan];  // bin -> 3 slot + (0, 1, 2: left neighbour, the bin, right neighbour) of the wide tap, or none
    const int chunk = blockIdx.x, span = blockIdx.y, band = blockIdx.z, tid = threadIdx.x, lane = threadIdx.x & 63;
    int carry_sel = carry_in_arg;
    if (cur) {
        g.count0 = cur->count0;
        carry_sel = cur->carry_in;
        if (chunk >= chunks_completed(g.count0, n_frames))
            return;
    }
    const int n = g.n, T = blockDim.x;
    const int bin0 = span * kRefineSpan;
    const int span_bins = min(kRefineSpan, n - bin0), words = span_bins >> 6;  // (n is a multiple of 64)
#if defined(SDR_REFINE_CLOCK)  // (tools only: tools/build_abl.sh refclk "-DSDR_REFINE_CLOCK")
    unsigned long long ck[6] = {}, ck_last = __builtin_amdgcn_s_memtime();
    int ck_n = 0;
#define SDR_REFINE_TICK()                                             \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ck[ck_n++] = now_ - ck_last;                                  \
        ck_last = now_;                                               \
    } while (0)
#else
#define SDR_REFINE_TICK()
#endif
    {
        const uint4 *src = static_cast<const uint4 *>(db_tab);
        uint4 *dst = reinterpret_cast<uint4 *>(s_tab);
        for (int i = tid; i < gomath::kDbTabBytes / 16; i += T)
            dst[i] = src[i];
    }
    // The wide tap (k_fft_r32.hip): the FFT kernel left psd at bin - 1, bin, bin + 1 of every listener's bin, frame by frame,
    // in rows of four kilobytes (16 bytes per listener) - and the bins FindPeaks looks at are the signals' and their neighbours.  A candidate found
    // in this map reads its column there (contiguous, on chip after the first touch) instead of a hundred 64-byte sectors
    // of the psd array for a hundred 4-byte values; any other candidate (a signal nobody listens to yet) reads the psd.
    for (int i = tid; i < kRefineSpan; i += T)
        s_tapidx[i] = 0xffffu;
    __syncthreads();
    if (tap_wide) {
        for (int l = tid; l < n_tap; l += T) {
            const int b = tap_used[(size_t)band * tap_stride + l];
            if (b >= 0) {
#pragma unroll
                for (int c3 = 0; c3 < 3; c3++) {
                    const int nb = b + c3 - 1;
                    if (nb >= 0 && nb < n && nb >= bin0 && nb < bin0 + span_bins)
                        s_tapidx[nb - bin0] = (unsigned short)(3 * l + c3);  // (two listeners side by side: either source holds the same word)
                }
            }
        }
    }
    const int first_len = SDR_CUMULATION_SIZE - g.count0;
    const int end_frame = first_len + chunk * SDR_CUMULATION_SIZE - 1;  // frame that completes this chunk
    const float thr = recs[(size_t)band * g.stride + end_frame].peak_thr;
    float *c = cum + ((size_t)band * g.max_chunks + chunk) * n;
    const float size = (float)SDR_CUMULATION_SIZE;
    // flag words of the span and of the word on either side of it (zero beyond the row)
    for (int w = (tid >> 6); w < words + 2; w += (T >> 6)) {  // (one wave per word: 64 whole bins per wave instruction)
        const int b = bin0 + (w - 1) * 64 + lane;
        const bool above = b >= 0 && b < n && __fdiv_rn(c[b], size) > thr;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(above);
        if (lane == 0)
            s_flags[w] = m;
    }
    __syncthreads();
    SDR_REFINE_TICK();  // table copy, threshold, flags
    for (int w = tid; w < words; w += T) {
        const unsigned long long f = s_flags[w + 1];
        s_rflags[w] = f | (f << 1) | (f >> 1) | (s_flags[w] >> 63) | (s_flags[w + 2] << 63);
    }
    __syncthreads();
    if (tid < 64)
        word_prefix(s_rflags, s_offs, words, tid);
    __syncthreads();
    const int n_exact = s_offs[words];
    if (n_exact == 0)  // (workgroup-uniform)
        return;
    for (int w = tid; w < words; w += T) {
        unsigned long long r = s_rflags[w];
        int at = s_offs[w];
        while (r) {
            s_list[at++] = (unsigned short)((w << 6) + __builtin_ctzll(r));
            r &= r - 1;
        }
    }
    __syncthreads();
    SDR_REFINE_TICK();  // the list
    const gomath::DbTables tab = gomath::db_tables(s_tab);
    int begin, len;
    cum_slot_frames(chunk, g.count0, &begin, &len);
    const float *__restrict__ carry_in = carry_sel ? carry1 : carry0;
    const float *__restrict__ base = psd + (size_t)band * g.stride * n;
#if SDR_REFINE_Q == 1
    for (int k0 = 0; k0 < n_exact; k0 += T) {
        const int k = k0 + tid;
        const bool mine = k < n_exact;
        const int bin = bin0 + (int)s_list[mine ? k : n_exact - 1];
        float acc = 0.f;
        if (chunk == 0 && g.count0 > 0)
            acc = carry_in[(size_t)band * n + bin];
        // (every lane of a wave that has an entry runs the column loop - the rare literal-log branch inside it is a
        // wave-wide vote - lanes without one repeat the last entry and store nothing)
        if (k0 + (tid & ~63) < n_exact) {
            // (scattered columns, one 64-byte sector per cluster of candidates and frame: latency-bound, many loads in flight)
            acc = cum_exact_column<SDR_REFINE_U>(base, (unsigned)begin * (unsigned)n + (unsigned)bin, (unsigned)n, len, acc, tab, inv_n2);
            if (mine)
                c[bin] = acc;  // the row holds the exact cumulation wherever FindPeaks reads it
        }
    }
#else
    // A column is a latency chain only in its SUM: the hundred loads and the hundred projections are independent.  Q
    // neighbouring lanes take a quarter of a column's frames each - every load of the column in flight at once, one
    // round trip to memory instead of seven - project their values side by side, and the ordered float32 sum walks
    // through them lane after lane: carry + frame 0 + frame 1 + ..., the reference's order, one add at a time.
    constexpr int Q = SDR_REFINE_Q, FR = (SDR_CUMULATION_SIZE + Q - 1) / Q;
    static_assert(Q == 2 || Q == 4 || Q == 8, "lanes of a column share a wave");
    const int n_lanes = n_exact * Q;
    for (int k0 = 0; k0 < n_lanes; k0 += T) {
        if (k0 + (tid & ~63) >= n_lanes)  // (wave-uniform)
            break;
        const int idx = k0 + tid, cand = idx / Q, q = idx % Q;
        const bool mine = cand < n_exact;  // (lanes without a column repeat the last one and store nothing)
        const int bin = bin0 + (int)s_list[mine ? cand : n_exact - 1];
        const int f0 = q * FR, cnt = min(FR, len - f0);  // (may be <= 0 for the short first cumulation of a batch)
        // the column: frame stride and first element, in the wide tap if the bin is there, else in the psd array
        const unsigned ti = s_tapidx[bin - bin0];
        const bool tapped = ti != 0xffffu;
        const unsigned step = tapped ? 4u * (unsigned)tap_stride : (unsigned)n;
        const float *__restrict__ col = tapped ? tap_wide + ((size_t)band * g.stride + (size_t)(begin + f0)) * (4u * (unsigned)tap_stride) +
                                                     (size_t)(ti / 3u) * 4u + ti % 3u
                                               : base + ((size_t)(begin + f0) * (unsigned)n + (unsigned)bin);
        float v[FR], db[FR];
#pragma unroll
        for (int k = 0; k < FR; k++)
            v[k] = k < cnt ? col[(size_t)k * step] : 1.0f;
        float acc = 0.f;
        if (q == 0 && chunk == 0 && g.count0 > 0)
            acc = carry_in[(size_t)band * n + bin];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < FR; k++)
            bad |= !gomath::psd_value_in_db_fast(v[k], tab, &db[k]);
        if (__builtin_amdgcn_ballot_w64(bad)) {
#pragma unroll 1
            for (int k = 0; k < FR; k++) {
                // (dynamic index into registers: select through a rotate of the arrays would cost more than this rare loop
                // is worth - the values are re-read from memory instead)
                if (k < cnt) {
                    const float x = col[(size_t)k * step];
                    float t;
                    if (!gomath::psd_value_in_db_fast(x, tab, &t)) {
                        t = db_slow(x, inv_n2);
#pragma unroll
                        for (int j = 0; j < FR; j++)
                            if (j == k)
                                db[j] = t;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < FR; k++)
            db[k] += (float)SDR_DBM_SHIFT;  // MagnitudeIndB + dBmShift (float32 add), off the chain
#pragma unroll
        for (int s = 0; s < Q; s++) {
            if (q == s) {
#pragma unroll
                for (int k = 0; k < FR; k++)
                    if (k < cnt)
                        acc += db[k];  // the ordered sum
            }
            if (s + 1 < Q) {
                const float up = __shfl_up(acc, 1);
                if (q == s + 1)
                    acc = up;
            }
        }
        if (mine && q == Q - 1)
            c[bin] = acc;  // the row holds the exact cumulation wherever FindPeaks reads it
    }
#endif
    SDR_REFINE_TICK();  // the columns
#if defined(SDR_REFINE_CLOCK)
    if (chunk == 1 && span == 0 && band == 0 && (tid == 0 || tid == T - 64))
        printf("refine clocks tid %d: flags %llu list %llu columns %llu candidates %d\n", tid, ck[0], ck[1], ck[2], n_exact);
#endif
}

// ---------------------------------------------------------------------------------------------
// k_find_peaks — dsp.FindPeaks (dsp/fft.go:254-285) on one completed cumulation, one workgroup per cumulation.  The row
// holds the exact cumulation wherever a bin can be above the threshold and beside such bins (k_cum_refine), an upper
// bound elsewhere: every comparison, maximum and centre value below is the reference's.
//  1. the row is read once, coalesced (lane = bin), divided by 100 (float32, :259) and kept in LDS; the 64 comparison
//     results of a wave instruction are one ballot word of a bit array (bit b = `value[b] > threshold`, :260);
//  2. run starts are bit operations on that array (a set bit whose predecessor is clear), counted with popcount and
//     numbered by one wave's prefix sum over the words: the peak list comes out in bin order, as the reference's;
//  3. the thread that owns a word walks the runs starting in it: the run's end is found in the bit array (count of
//     trailing ones), its maximum in the LDS row, first maximum wins (strict `<`, :270); a run still open at the last
//     bin ends there (:276-282).
// ---------------------------------------------------------------------------------------------
constexpr int kPeakThreadsMax = 1024;

__global__ __launch_bounds__(kPeakThreadsMax) void k_find_peaks(const float *__restrict__ cum, const sdr_frame_rec *__restrict__ recs,
                                                                 DevPeak *__restrict__ peaks, int *__restrict__ counts,
                                                                 const BatchCursor *__restrict__ cur, PeakGeom g, int n_frames)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_peaks[];
    const int chunk = blockIdx.x, band = blockIdx.y, tid = threadIdx.x, lane = threadIdx.x & 63;
    if (cur) {
        g.count0 = cur->count0;
        if (chunk >= chunks_completed(g.count0, n_frames))
            return;
    }
    const int n = g.n, words = peak_words(n), T = blockDim.x;
    float *val = reinterpret_cast<float *>(smem_peaks);                                      // [n]
    unsigned long long *flags = reinterpret_cast<unsigned long long *>(smem_peaks + (size_t)n * 4);  // [words]
    unsigned long long *starts = flags + words;                                              // [words]
    int *offs = reinterpret_cast<int *>(starts + words);                                     // [words + 1]
    const int first_len = SDR_CUMULATION_SIZE - g.count0;
    const int end_frame = first_len + chunk * SDR_CUMULATION_SIZE - 1;  // frame that completes this chunk
    const float thr = recs[(size_t)band * g.stride + end_frame].peak_thr;
    const float *c = cum + ((size_t)band * g.max_chunks + chunk) * n;
    const float size = (float)SDR_CUMULATION_SIZE;
    // 1. values and flag words (n is a multiple of T, T a multiple of 64: every wave instruction covers 64 whole bins)
#pragma unroll 8
    for (int b = tid; b < n; b += T) {
        const float v = __fdiv_rn(c[b], size);
        val[b] = v;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(v > thr);
        if (lane == 0)
            flags[b >> 6] = m;
    }
    __syncthreads();
    // 2. run starts per word, numbered by a prefix sum
    for (int w = tid; w < words; w += T) {
        const unsigned long long f = flags[w];
        const unsigned long long before = (f << 1) | (w > 0 ? flags[w - 1] >> 63 : 0ull);
        starts[w] = f & ~before;
    }
    __syncthreads();
    if (tid < 64) {
        const int total = word_prefix(starts, offs, words, tid);
        if (tid == 63)
            counts[(size_t)band * g.max_chunks + chunk] = total;
    }
    __syncthreads();
    // 3. the runs that start in word w
    for (int w = tid; w < words; w += T) {
        unsigned long long st = starts[w];
        int idx = offs[w];
        while (st) {
            const int k = __builtin_ctzll(st);
            st &= st - 1;
            const int from = (w << 6) + k;
            // end of the run: first clear bit at or after `from`
            int to;
            {
                int ww = w;
                unsigned long long inv = ~flags[ww] >> k;  // bit j = bin from + j is NOT flagged (zeros shifted in at the top: handled below)
                int base = from;
                int room = 64 - k;  // valid bits in `inv`
                for (;;) {
                    const int z = inv ? __builtin_ctzll(inv) : 64;
                    if (z < room) {
                        to = base + z - 1;
                        break;
                    }
                    ww++;
                    base += room;
                    if (ww >= words) {
                        to = n - 1;
                        break;
                    }
                    inv = ~flags[ww];
                    room = 64;
                }
            }
            if (idx < g.max_peaks) {
                float best = val[from];
                int best_bin = from;
                for (int j = from + 1; j <= to; j++) {
                    const float vj = val[j];
                    if (best < vj) {
                        best = vj;
                        best_bin = j;
                    }
                }
                DevPeak p;
                p.from = from;
                p.to = to;
                p.signal_bin = best_bin;
                p.signal_value = best;
                p.y1 = best_bin > 0 ? c[best_bin - 1] : 0.f;  // (exact: neighbours of a flagged bin were refined)
                p.y2 = c[best_bin];
                p.y3 = best_bin < n - 1 ? c[best_bin + 1] : 0.f;
                peaks[((size_t)band * g.max_chunks + chunk) * g.max_peaks + idx] = p;
            }
            idx++;
        }
    }
}

// Bound-and-refine replaces the exact kernel's work with three launches on the peaks stream; on a short batch their fixed
// latencies (a refinement is a chain of a hundred scattered sector reads per candidate, whatever the batch) make that
// stream the longest of the four: config 3 at 2048 frames per batch 135-143 GS/s against 150 with every slot exact.  From
// 64 M samples per batch on it pays (config 3 at 8192 frames, config 5's share of 8 x 2048 x 8192).  SDR_CUM_BOUND=0 / 1
// forces one (development).
bool cum_bound_pays(int n_frames, int n_bands, int n)
{
    static const int force = getenv("SDR_CUM_BOUND") ? atoi(getenv("SDR_CUM_BOUND")) : -1;
    if (force >= 0)
        return force != 0;
    return (double)n_frames * (double)n_bands * (double)n >= 64.0 * 1024.0 * 1024.0;
}

// k_bound_finish - the bounds of the cumulations a batch completes from the unit counts k_psd_scan left in their rows
// (one or two partial counts per bin: a slot's frames may have been dealt over two workgroups) and, for the first one -
// slot 0 continues the cumulation carried in - the carry: gomath::cum_bound, as k_cum_bound forms it.
__global__ __launch_bounds__(256) void k_bound_finish(float *__restrict__ cum_out, const float *__restrict__ cum_part, int parts,
                                                      const float *__restrict__ carry0, const float *__restrict__ carry1, int carry_in_arg,
                                                      const BatchCursor *__restrict__ cur, CumGeom g, double a128, double per_frame)
{
    int carry_sel = carry_in_arg;
    if (cur) {
        g.count0 = cur->count0;
        carry_sel = cur->carry_in;
    }
    const int slot = blockIdx.y, band = blockIdx.z;
    int begin, len;
    cum_slot_frames(slot, g.count0, &begin, &len);
    if (begin + len > g.n_frames)
        return;  // the batch does not complete it
    const int bin = blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= g.n)
        return;
    const float *__restrict__ carry_in = carry_sel ? carry1 : carry0;
    const size_t at = ((size_t)band * g.max_chunks + slot) * g.n + bin;
    unsigned units = __float_as_uint(cum_out[at]);
    bool special = units == 0xffffffffu;
    if (parts > 1) {
        const unsigned u1 = __float_as_uint(cum_part[at]);
        special = special || u1 == 0xffffffffu;
        units += u1;
    }
    const double c0 = (slot == 0 && g.count0 > 0) ? (double)carry_in[(size_t)band * g.n + bin] : 0.0;
    cum_out[at] = gomath::cum_bound(c0, units, len, a128, per_frame, special);
}

// One batch's cumulation work, on `stream`: bounds of the cumulations it completes, the exact carry of the one it leaves
// open.  (A stage event armed by the caller rides on the last launch.)
hipError_t launch_cumulate(const float *psd, const void *db_tab, float *carry0, float *carry1, int carry_in, float *cum_out,
                           const float *cum_part, const BatchCursor *cur, CumGeom g, int n_slots, int n_bands, bool bound_done, hipStream_t stream)
{
    const double inv_n2 = 1.0 / ((double)g.n * (double)g.n);
    const int threads = g.n < SDR_CUM_THREADS ? g.n : SDR_CUM_THREADS;
    if (!cum_bound_pays(g.n_frames, n_bands, g.n)) {  // short batches: every slot exact, one launch (as rounds 1-3)
        launch_kernel(k_cumulate, dim3((g.n + threads - 1) / threads, n_slots, n_bands), dim3(threads), 0, stream, psd, db_tab, carry0, carry1,
                      carry_in, cum_out, static_cast<float *>(nullptr), cur, g, 0, 0, inv_n2);
        return hipGetLastError();
    }
    double a128, per_frame;
    gomath::cum_bound_constants(g.n, &a128, &per_frame);
    const hipEvent_t done = t_done_event;
    t_done_event = nullptr;
    if (bound_done) {
        hipLaunchKernelGGL(k_bound_finish, dim3((g.n + 255) / 256, n_slots, n_bands), dim3(256), 0, stream, cum_out, cum_part,
                           scan_parts(n_slots, n_bands), carry0, carry1, carry_in, cur, g, a128, per_frame);
    } else {
        static std::once_flag attr_once[64];
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess)
            return e;
        if (dev < 0 || dev >= 64)
            return hipErrorInvalidDevice;
        hipError_t attr_err = hipSuccess;
        std::call_once(attr_once[dev], [&] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cum_bound), hipFuncAttributeMaxDynamicSharedMemorySize, kBoundLdsBytes);
        });
        if (attr_err != hipSuccess)
            return attr_err;
        static const int wgs_env = getenv("SDR_BOUND_WGS") ? atoi(getenv("SDR_BOUND_WGS")) : SDR_BOUND_WGS;  // (development)
        const int threads = (g.n / 4) < kBoundThreads ? g.n / 4 : kBoundThreads;
        const int blocks = (g.n / 4 + threads - 1) / threads, items = n_bands * n_slots * blocks;
        // as few workgroups as keep the kernel a fraction of the batch's FFT time (the peaks stream must not become the
        // longest of the four): one item per workgroup for a 2048 x 16384 batch (84 items, round 4: 64 workgroups took two
        // rounds there and bound the step - 127 instead of 150 GS/s), about four for four times the samples
        const double batch_samples = (double)g.n_frames * (double)n_bands * (double)g.n;
        int rounds = (int)(batch_samples / (2048.0 * 16384.0) + 0.5);
        rounds = rounds < 1 ? 1 : (rounds > 5 ? 5 : rounds);
        int wgs = (items + rounds - 1) / rounds;
        if (getenv("SDR_BOUND_WGS"))
            wgs = wgs_env;
        wgs = wgs < 1 ? 1 : (wgs > items ? items : wgs);
        hipLaunchKernelGGL(k_cum_bound, dim3(wgs), dim3(threads), kBoundLdsBytes, stream, psd, carry0, carry1, carry_in, cum_out, cur, g, n_slots,
                           n_bands, a128, per_frame);
    }
    t_done_event = done;
    launch_kernel(k_cumulate, dim3((g.n + threads - 1) / threads, 1, n_bands), dim3(threads), 0, stream, psd, db_tab, carry0, carry1, carry_in,
                  cum_out, static_cast<float *>(nullptr), cur, g, 1, 0, inv_n2);
    return hipGetLastError();
}

// the exact cumulation `slot` of one band, on demand (sdr_read_cumulation, the scope tap): psd / carry_in point at the band
hipError_t launch_cumulation_row(const float *psd_band, const void *db_tab, const float *carry_in_band, float *row_out, CumGeom g, int slot,
                                 hipStream_t stream)
{
    const double inv_n2 = 1.0 / ((double)g.n * (double)g.n);
    const int threads = g.n < SDR_CUM_THREADS ? g.n : SDR_CUM_THREADS;
    float *ci = const_cast<float *>(carry_in_band);
    hipLaunchKernelGGL(k_cumulate, dim3((g.n + threads - 1) / threads, 1, 1), dim3(threads), 0, stream, psd_band, db_tab, ci, ci, 0,
                       static_cast<float *>(nullptr), row_out, static_cast<const BatchCursor *>(nullptr), g, 0, slot, inv_n2);
    return hipGetLastError();
}

hipError_t launch_spectrum_row(const float *psd_row, float *out, int n, hipStream_t stream)
{
    const double inv_n2 = 1.0 / ((double)n * (double)n);
    hipLaunchKernelGGL(k_spectrum_row, dim3((n + 255) / 256), dim3(256), 0, stream, psd_row, out, n, inv_n2);
    return hipGetLastError();
}

hipError_t launch_find_peaks(float *cum, const float *psd, const void *db_tab, const float *carry0, const float *carry1, int carry_in,
                             const sdr_frame_rec *recs, DevPeak *peaks, int *counts, const BatchCursor *cur, PeakGeom g, int n_frames,
                             int n_chunks, int n_bands, FftTap tap, hipStream_t stream)
{
    if (n_chunks == 0)
        return hipSuccess;
    const int words = g.n >> 6;
    const unsigned lds = (unsigned)((size_t)g.n * 4 + (size_t)words * 16 + (size_t)(words + 1) * 4);
    // more than 64 KB of dynamic LDS needs the attribute, once per device
    static std::once_flag attr_once[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    if (dev < 0 || dev >= 64)
        return hipErrorInvalidDevice;
    hipError_t attr_err = hipSuccess;
    std::call_once(attr_once[dev], [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_find_peaks), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       16384 * 4 + 256 * 16 + 257 * 4);
    });
    if (attr_err != hipSuccess)
        return attr_err;
    // the exact cumulation where the scan will look, then the scan (a stage event armed by the caller rides on the scan)
    const hipEvent_t done = t_done_event;
    t_done_event = nullptr;
    const double inv_n2 = 1.0 / ((double)g.n * (double)g.n);
    if (cum_bound_pays(n_frames, n_bands, g.n))  // (otherwise k_cumulate left every row exact)
    {
        // Spans of 4096 bins x 256 threads keep a cumulation's refinement - a latency chain of a hundred scattered sector
        // reads - short where the peaks stream's length bounds the step (few cumulations per batch).  With many
        // cumulations per batch what counts is the CU time the kernel HOLDS: four waves of a small workgroup hold a whole
        // CU against the FFT's workgroups just as sixteen do, so a workgroup takes the whole row (10.9 -> CU-ms per
        // 8192-frame step; SDR_REFINE_WIDE = 0 / 1 forces one).
        static const int wide_env = getenv("SDR_REFINE_WIDE") ? atoi(getenv("SDR_REFINE_WIDE")) : -1;
        const bool wide = wide_env >= 0 ? wide_env != 0 : (g.n >= 4096 && (long)n_chunks * n_bands >= 64);
        if (wide)
            hipLaunchKernelGGL((k_cum_refine<16384, 1024>), dim3(n_chunks, (g.n + 16383) / 16384, n_bands), dim3(1024), 0, stream, cum, psd, db_tab,
                               carry0, carry1, carry_in, recs, cur, g, n_frames, inv_n2, tap.wide, tap.used, tap.n, tap.stride);
        else
            hipLaunchKernelGGL((k_cum_refine<kRefineSpan, kRefineThreads>), dim3(n_chunks, (g.n + kRefineSpan - 1) / kRefineSpan, n_bands),
                               dim3(kRefineThreads), 0, stream, cum, psd, db_tab, carry0, carry1, carry_in, recs, cur, g, n_frames, inv_n2, tap.wide,
                               tap.used, tap.n, tap.stride);
    }
    t_done_event = done;
    const int threads = g.n < kPeakThreadsMax ? g.n : kPeakThreadsMax;
    launch_kernel(k_find_peaks, dim3(n_chunks, n_bands), dim3(threads), lds, stream, static_cast<const float *>(cum), recs, peaks, counts, cur, g,
                  n_frames);
    return hipGetLastError();
}

}  // namespace sdr
