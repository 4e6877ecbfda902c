// fft_f64.h — register-tiled radix-2 DIT FFT in float64 whose arithmetic is *bit-identical* to the
// stage-by-stage radix-2 algorithm the reference calls (go-dsp fft.FFT, dsp/fft.go:26).
//
// Idea: a radix-2 decimation-in-time FFT is a fixed dataflow graph of butterflies
//     t = r[i+h] * W[(N/2h) * (i mod h)];  r[i] = r[i] + t;  r[i+h] = r[i] - t
// over a bit-reversed copy of the input.  Any schedule that evaluates that same graph with the same
// IEEE operations (complex multiply as (ac-bd, ad+bc), no FMA contraction) and the same twiddle
// VALUES produces the same bits.  So each thread keeps R = 16 (8 for N=512) points in VGPRs and runs
// LOGR consecutive stages on them (a "pass"), then the workgroup transposes through LDS for the next
// pass.  Twiddles are never recomputed on the device: the host builds go-dsp's table (radix2 factors
// via math.Sincos, even entries copied from the half-size table) and uploads it re-laid-out per pass
// so a wave reads them with coalesced 16-byte loads.
//
// Everything here is `SDR_HD` and free of HIP intrinsics so tests/emu can run the very same phase
// functions thread-by-thread on the CPU to validate the index math without a GPU.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define SDR_HD __host__ __device__
#else
#define SDR_HD
#endif

namespace fft64 {

struct cplx {
    double x, y;
};

template <int LOGN>
struct Plan {
    static constexpr int N = 1 << LOGN;
    // points per thread: 16 for N >= 1024 (radix-16 register passes), 8 at N = 512.  (32 points at
    // N = 16384 would save an exchange, but its 31 twiddles per pass plus a prefetched frame spill.)
    static constexpr int LOGR = (LOGN >= 10) ? 4 : 3;
    static constexpr int R = 1 << LOGR;                       // points per thread
    static constexpr int T = N / R;                           // threads per frame
    static constexpr int NPASS = (LOGN + LOGR - 1) / LOGR;    // register passes
    static constexpr int LAST_LOG = LOGN - (NPASS - 1) * LOGR;
    static constexpr bool SPLIT = (N * 16 > 65536);           // exchange re / im separately through LDS
    static constexpr int LDS_BYTES = SPLIT ? N * 8 : N * 16;

    SDR_HD static constexpr int pass_log(int p) { return p < NPASS - 1 ? LOGR : LAST_LOG; }
    // offset (in entries) of pass p's twiddle block: block p holds (2^pass_log(p) - 1) * R^p entries
    SDR_HD static constexpr int tw_offset(int p)
    {
        int o = 0;
        for (int k = 0; k < p; k++)
            o += ((1 << pass_log(k)) - 1) << (k * LOGR);
        return o;
    }
    static constexpr int TW_TOTAL = tw_offset(NPASS);
};

SDR_HD inline unsigned brev_bits(unsigned v, int bits)
{
    unsigned r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

// Index (in the bit-reversed work array r[]) of register slot (u, m) of thread t during pass P.
template <int LOGN, int P>
SDR_HD inline int elem_index(int t, int u, int m)
{
    using PL = Plan<LOGN>;
    if (P == 0)
        return (int)(brev_bits((unsigned)t, LOGN - PL::LOGR) << PL::LOGR) + m;
    constexpr int SH = P * PL::LOGR;
    constexpr int PLOG = PL::pass_log(P);
    const int g = t + PL::T * u;
    const int hi = g >> SH;
    const int lo = g & ((1 << SH) - 1);
    return (hi << (SH + PLOG)) + (m << SH) + lo;
}

// LDS swizzle of exchange E (between pass E and E+1).  Only the first exchange needs one: there a
// thread writes 16 consecutive doubles (stride-16 across lanes) and consecutive lanes own groups that
// differ in their TOP bits (g = bitrev(t)), so the low LOGR bits are XORed with the top LOGR bits.
template <int LOGN, int E>
SDR_HD inline int swz(int i)
{
    return E == 0 ? (i ^ (i >> (LOGN - Plan<LOGN>::LOGR))) : i;
}

// Pass 0 input: slot m <- x[bitrev_LOGR(m) * T + t]  (coalesced across t), widened to float64
// (dsp/fft.go:59-69 setSamplesFromIQ).
template <int LOGN>
SDR_HD inline void load_input(const float *iq, int t, double *xr, double *xi)
{
    using PL = Plan<LOGN>;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int m = 0; m < PL::R; m++) {
        const int k = (int)brev_bits((unsigned)m, PL::LOGR);
        const int n = k * PL::T + t;
        xr[m] = (double)iq[2 * n];
        xi[m] = (double)iq[2 * n + 1];
    }
}

// Pass P: pass_log(P) radix-2 stages on the thread's registers.
template <int LOGN, int P>
SDR_HD inline void butterfly_pass(double *xr, double *xi, int t, const cplx *tw)
{
    using PL = Plan<LOGN>;
    constexpr int PLOG = PL::pass_log(P);
    constexpr int RP = 1 << PLOG;
    constexpr int G = PL::R / RP;
    constexpr int SH = P * PL::LOGR;
    constexpr int S = 1 << SH;
    constexpr int OFF = PL::tw_offset(P);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int u = 0; u < G; u++) {
        const int lo = (P == 0) ? 0 : ((t + PL::T * u) & (S - 1));
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int q = 0; q < PLOG; q++) {
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int mm = 0; mm < (1 << q); mm++) {
                // pass 0 has thread-independent twiddles; W[0] = 1 and W[N/4] = -i are the literal
                // entries of go-dsp's size-4 table, multiplying by them is exact (up to the sign of a
                // zero, which cannot reach |X|^2), so the multiply is skipped.
                const bool one = (P == 0 && mm == 0);
                const bool minus_i = (P == 0 && q >= 1 && mm == (1 << (q - 1)));
                double wr = 1.0, wi = 0.0;
                if (!one && !minus_i) {
#if defined(SDR_ABLATE) && (SDR_ABLATE == 3)
                    wr = 0.5 + 1e-9 * lo;  // timing-only build: no twiddle loads
                    wi = 0.25;
#else
                    const cplx w = tw[OFF + ((1 << q) - 1 + mm) * S + lo];
                    wr = w.x;
                    wi = w.y;
#endif
                }
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (int k = 0; k < (RP >> (q + 1)); k++) {
                    const int a = u * RP + mm + (k << (q + 1));
                    const int b = a + (1 << q);
                    double tr, ti;
                    if (one) {
                        tr = xr[b];
                        ti = xi[b];
                    } else if (minus_i) {
                        tr = xi[b];
                        ti = -xr[b];
                    } else {
                        tr = xr[b] * wr - xi[b] * wi;  // Go complex128 multiply, amd64: no FMA
                        ti = xr[b] * wi + xi[b] * wr;
                    }
                    const double ar = xr[a], ai = xi[a];
                    xr[a] = ar + tr;
                    xi[a] = ai + ti;
                    xr[b] = ar - tr;
                    xi[b] = ai - ti;
                }
            }
        }
    }
}

// Exchange E, write side: scatter the thread's slots (pass E layout) into LDS.
template <int LOGN, int E>
SDR_HD inline void exchange_write(const double *x, int t, double *lds)
{
    using PL = Plan<LOGN>;
    constexpr int RP = 1 << PL::pass_log(E);
    constexpr int G = PL::R / RP;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int u = 0; u < G; u++)
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int m = 0; m < RP; m++)
            lds[swz<LOGN, E>(elem_index<LOGN, E>(t, u, m))] = x[u * RP + m];
}

// Exchange E, read side: gather the slots of pass E+1.
template <int LOGN, int E>
SDR_HD inline void exchange_read(double *x, int t, const double *lds)
{
    using PL = Plan<LOGN>;
    constexpr int RP = 1 << PL::pass_log(E + 1);
    constexpr int G = PL::R / RP;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int u = 0; u < G; u++)
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int m = 0; m < RP; m++)
            x[u * RP + m] = lds[swz<LOGN, E>(elem_index<LOGN, E + 1>(t, u, m))];
}

// Natural-order DFT bin held in register slot s = u*RP+m after the last pass.
template <int LOGN>
SDR_HD inline int output_bin(int t, int s)
{
    using PL = Plan<LOGN>;
    constexpr int RP = 1 << PL::LAST_LOG;
    return elem_index<LOGN, PL::NPASS - 1>(t, s / RP, s % RP);
}

// (host only)
// Host: lay go-dsp's factor table W[k] = e^{-2 pi i k / N} (k < N) out per pass:
// entry OFF_p + ((2^q - 1) + mm) * S_p + lo  =  W[(N / (2 * S_p * 2^q)) * (mm * S_p + lo)].
template <int LOGN>
inline void build_pass_twiddles(const double *wre, const double *wim, cplx *out)
{
    using PL = Plan<LOGN>;
    for (int p = 0; p < PL::NPASS; p++) {
        const int S = 1 << (p * PL::LOGR);
        const int off = PL::tw_offset(p);
        for (int q = 0; q < PL::pass_log(p); q++)
            for (int mm = 0; mm < (1 << q); mm++)
                for (int lo = 0; lo < S; lo++) {
                    const int h = S << q;
                    const int k = (PL::N / (2 * h)) * (mm * S + lo);
                    out[off + ((1 << q) - 1 + mm) * S + lo] = cplx{wre[k], wim[k]};
                }
    }
}

}  // namespace fft64
