// k_peaks.hip — dB projection + ordered float32 cumulation over 100 frames, and the run-length peak scan of each
// completed cumulation.  Compiled with -ffp-contract=off (see gomath.h).
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "fft_f64.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

// ---------------------------------------------------------------------------------------------
// k_cumulate — cumulation[i] += spectrum[i] (rx/receiver.go:404-407): a float32 sum in frame order, where
// spectrum[i] = MagnitudeIndB(...) + dBmShift (dsp/fft.go:79-81, rx/receiver.go:377) is evaluated here from the
// float32 psd the FFT kernel stored: the projection is a pure function of that value, and this is the only
// place that needs it for every bin.  One thread per bin, lanes on neighbouring bins (coalesced 256-byte rows),
// ten loads in flight, then ten ORDERED adds.  The certified table shortcut of gomath.h settles all but about
// three values in 10^5; the rest take the literal Go algorithm in a (rare, divergent) branch.
// Slot 0 continues the cumulation carried over from the previous batch; a slot that reaches 100
// frames is written out for the peak scan, an incomplete last slot becomes the next carry.
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

// 256-thread workgroups, 20 KB of LDS (the tables), at most 64 VGPRs: eight waves per SIMD.  (A 32-VGPR build -
// what a CU has left beside a resident k_fft_psd workgroup, so that the two could share a CU - measured the same
// step, so the registers go to a software pipeline instead.)
#if !defined(SDR_CUM_THREADS)
#define SDR_CUM_THREADS 256
#endif
#if !defined(SDR_CUM_U)
#define SDR_CUM_U 4
#endif
#if !defined(SDR_CUM_VGPR)
#define SDR_CUM_VGPR 32
#endif
__global__ __launch_bounds__(SDR_CUM_THREADS) __attribute__((amdgpu_num_vgpr(SDR_CUM_VGPR))) void k_cumulate(const float *__restrict__ psd, const void *__restrict__ db_tab,
                                                   float *__restrict__ carry0, float *__restrict__ carry1, int carry_in_arg,
                                                   float *__restrict__ cum_out, const BatchCursor *__restrict__ cur, CumGeom g,
                                                   double inv_n2)
{
    int carry_sel = carry_in_arg;
    if (cur) {  // graph replay: this batch's cumulation phase comes from device memory; the grid covers every slot a
        g.count0 = cur->count0;  // batch of this length can have, surplus workgroups leave
        carry_sel = cur->carry_in;
        const int first = SDR_CUMULATION_SIZE - g.count0;
        const int slots = g.n_frames <= first ? 1 : 1 + (g.n_frames - first + SDR_CUMULATION_SIZE - 1) / SDR_CUMULATION_SIZE;
        if ((int)blockIdx.y >= slots)
            return;
    }
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
    const int slot = blockIdx.y, band = blockIdx.z;
    // frames of this slot: slot 0 takes (100 - count0) frames, later slots 100 each
    const int first_len = SDR_CUMULATION_SIZE - g.count0;
    const int begin = slot == 0 ? 0 : first_len + (slot - 1) * SDR_CUMULATION_SIZE;
    const int len = slot == 0 ? first_len : SDR_CUMULATION_SIZE;
    const int end = min(begin + len, g.n_frames);
    float acc = 0.f;
    if (slot == 0 && g.count0 > 0)
        acc = carry_in[(size_t)band * g.n + bin];
    // (wave-uniform base in SGPRs + one 32-bit per-lane offset)
    const float *__restrict__ base = psd + (size_t)band * g.stride * g.n;
    const unsigned n = (unsigned)g.n;
    unsigned off = (unsigned)begin * n + (unsigned)bin;
    constexpr int U = SDR_CUM_U;  // loads in flight, and independent dB evaluations between two ordered adds
    int f = begin;
    // software pipeline: the next U values are on their way while these U are projected and added
    float v[U], nv[U];
    const bool any = f + U <= end;
    if (any) {
#pragma unroll
        for (int k = 0; k < U; k++)
            nv[k] = __builtin_nontemporal_load(base + (off + (unsigned)k * n));
    }
    for (; f + U <= end; f += U) {
#pragma unroll
        for (int k = 0; k < U; k++)
            v[k] = nv[k];
        off += U * n;
        if (f + 2 * U <= end) {
#pragma unroll
            for (int k = 0; k < U; k++)
                nv[k] = __builtin_nontemporal_load(base + (off + (unsigned)k * n));
        }
        float db[U];
        // the shortcut for all of them, straight-line (independent float64 chains); the literal algorithm only
        // where the certificate failed - about one iteration in a few hundred has such a lane
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
    for (; f < end; f++, off += n)
        acc += spectrum_value(__builtin_nontemporal_load(base + off), tab, inv_n2);
    const bool complete = (begin + len) <= g.n_frames;
    if (complete) {
        // completed chunk index == slot (slot 0 completes first if it completes at all)
        cum_out[((size_t)band * g.max_chunks + slot) * g.n + bin] = acc;
    } else {
        carry_out[(size_t)band * g.n + bin] = acc;
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
// k_find_peaks — dsp.FindPeaks (dsp/fft.go:254-285) on one completed cumulation, one workgroup per cumulation.
//  1. the row is read once, coalesced (lane = bin), divided by 100 (float32, :259) and kept in LDS; the 64 comparison
//     results of a wave instruction are one ballot word of a bit array (bit b = `value[b] > threshold`, :260);
//  2. run starts are bit operations on that array (a set bit whose predecessor is clear), counted with popcount and
//     numbered by one wave's prefix sum over the words: the peak list comes out in bin order, as the reference's;
//  3. the thread that owns a word walks the runs starting in it: the run's end is found in the bit array (count of
//     trailing ones), its maximum in the LDS row, first maximum wins (strict `<`, :270); a run still open at the last
//     bin ends there (:276-282).
// (Round 2's kernel gave each thread 64 consecutive bins - lanes 256 bytes apart - scanned the starts through 16
// barriers and walked runs in global memory: 0.040 ms standalone, 0.08 inside the pipeline.)
// ---------------------------------------------------------------------------------------------
constexpr int kPeakThreadsMax = 1024;
__device__ __forceinline__ int peak_words(int n) { return n >> 6; }

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
    // 2. run starts per word, numbered by a prefix sum (wave 0: each lane takes `per` consecutive words)
    if (tid < 64) {
        const int per = (words + 63) >> 6;
        int local = 0;
        for (int k = 0; k < per; k++) {
            const int w = tid * per + k;
            if (w < words) {
                const unsigned long long f = flags[w];
                const unsigned long long before = (f << 1) | (w > 0 ? flags[w - 1] >> 63 : 0ull);
                const unsigned long long st = f & ~before;
                starts[w] = st;
                local += __popcll(st);
            }
        }
        int incl = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (tid >= o)
                incl += up;
        }
        int run = incl - local;
        for (int k = 0; k < per; k++) {
            const int w = tid * per + k;
            if (w < words) {
                offs[w] = run;
                run += __popcll(starts[w]);
            }
        }
        if (tid == 63) {
            offs[words] = incl;
            counts[(size_t)band * g.max_chunks + chunk] = incl;
        }
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
                p.y1 = best_bin > 0 ? c[best_bin - 1] : 0.f;
                p.y2 = c[best_bin];
                p.y3 = best_bin < n - 1 ? c[best_bin + 1] : 0.f;
                peaks[((size_t)band * g.max_chunks + chunk) * g.max_peaks + idx] = p;
            }
            idx++;
        }
    }
}

hipError_t launch_cumulate(const float *psd, const void *db_tab, float *carry0, float *carry1, int carry_in, float *cum_out,
                           const BatchCursor *cur, CumGeom g, int n_slots, int n_bands, hipStream_t stream)
{
    const int threads = g.n < SDR_CUM_THREADS ? g.n : SDR_CUM_THREADS;
    const double inv_n2 = 1.0 / ((double)g.n * (double)g.n);
    launch_kernel(k_cumulate, dim3((g.n + threads - 1) / threads, n_slots, n_bands), dim3(threads), 0, stream, psd, db_tab,
                       carry0, carry1, carry_in, cum_out, cur, g, inv_n2);
    return hipGetLastError();
}

hipError_t launch_spectrum_row(const float *psd_row, float *out, int n, hipStream_t stream)
{
    const double inv_n2 = 1.0 / ((double)n * (double)n);
    hipLaunchKernelGGL(k_spectrum_row, dim3((n + 255) / 256), dim3(256), 0, stream, psd_row, out, n, inv_n2);
    return hipGetLastError();
}

hipError_t launch_find_peaks(const float *cum, const sdr_frame_rec *recs, DevPeak *peaks, int *counts, const BatchCursor *cur,
                             PeakGeom g, int n_frames, int n_chunks, int n_bands, hipStream_t stream)
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
    const int threads = g.n < kPeakThreadsMax ? g.n : kPeakThreadsMax;
    launch_kernel(k_find_peaks, dim3(n_chunks, n_bands), dim3(threads), lds, stream, cum, recs, peaks, counts, cur, g, n_frames);
    return hipGetLastError();
}

}  // namespace sdr
