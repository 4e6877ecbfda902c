// fft_f64.h — register-tiled radix-2 DIT FFT in float64 whose arithmetic is *bit-identical* to the
// stage-by-stage radix-2 algorithm the reference calls (go-dsp fft.FFT, dsp/fft.go:26).
//
// Idea: a radix-2 decimation-in-time FFT is a fixed dataflow graph of butterflies
//     t = r[i+h] * W[(N/2h) * (i mod h)];  r[i] = r[i] + t;  r[i+h] = r[i] - t
// over a bit-reversed copy of the input.  Any schedule that evaluates that same graph with the same
// IEEE operations (complex multiply as (ac-bd, ad+bc), no FMA contraction) and the same twiddle
// VALUES produces the same bits.  So each thread keeps R = 16 (8 for N=512) points in VGPRs and runs
// LOGR consecutive stages on them (a "pass"), then the data is exchanged for the next pass - inside a
// wave's own block of LDS, across waves through LDS (once), or in registers (make_layout, make_swap_plan).  Twiddles are never recomputed on the device: the host builds go-dsp's table (radix2 factors
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
    static constexpr int LOGT = LOGN - LOGR;
    static constexpr int LOGL = 6;                            // lane bits of a thread id (64-wide waves)
    static constexpr int WB = LOGT - LOGL;                    // wave bits of a thread id
    static constexpr int NPASS = (LOGN + LOGR - 1) / LOGR;    // register passes
    static constexpr int LAST_LOG = LOGN - (NPASS - 1) * LOGR;
    static constexpr bool SPLIT = (N * 16 > 65536);           // exchange re / im separately through LDS
    static constexpr int LDS_BYTES = SPLIT ? N * 8 : N * 16;
    static_assert(LOGT >= LOGL, "a frame needs at least one full wave");

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
    // Exchange e (between pass e and e+1) moves data between waves only when the set of index bits
    // held in the wave id changes; that happens once, after pass 1 (see make_layout).  Every other
    // exchange stays inside a wave and needs no workgroup barrier.
    SDR_HD static constexpr bool cross_wave(int e) { return WB > 0 && e == 1; }
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

// ---------------------------------------------------------------------------------------------
// Who holds what.  During pass P the LOGN bits of an element's index i (in the bit-reversed work
// array) are spread over the register slot (LOGR bits), the lane (6 bits) and the wave (WB bits):
//
//   slot : the bits the pass's stages pair up, [4P, 4P+pass_log); a short last pass fills the slot
//          with the bits just below them
//   wave : passes 0,1: bits [8, 8+WB); later passes: the WB bits below the last pass's slot bits,
//          i.e. [8-WB, 8) (one lower when the last pass borrowed bit 7)
//   lane : everything else.  Pass 0 orders them downwards (lane bit 0 = top index bit = lowest
//          sample-number bit, so neighbouring lanes load neighbouring samples); later passes
//          upwards (lane bits 0-3 = index bits 0-3: coalesced twiddle loads and output stores).
//
// With that choice passes 0 and 1 share their wave bits, and so do passes 2 and 3: only the exchange
// after pass 1 crosses waves.
// ---------------------------------------------------------------------------------------------
struct Layout {
    int sbit[4];   // index bit held by slot bit j
    int tbit[10];  // index bit held by thread-id bit j (lanes first, then waves)
};

template <int LOGN>
SDR_HD constexpr Layout make_layout(int P)
{
    using PL = Plan<LOGN>;
    Layout L{};
    bool used[16] = {};
    const int plog = PL::pass_log(P);
    const int base = P * PL::LOGR;
    for (int j = 0; j < PL::LOGR; j++) {
        L.sbit[j] = j < plog ? base + j : base - (PL::LOGR - plog) + (j - plog);
        used[L.sbit[j]] = true;
    }
    const int wz_hi = (PL::NPASS == 3) ? 2 * PL::LOGR - (PL::LOGR - PL::LAST_LOG) : 2 * PL::LOGR;
    const int wlo = (P <= 1) ? 2 * PL::LOGR : wz_hi - PL::WB;
    for (int j = 0; j < PL::WB; j++) {
        L.tbit[PL::LOGL + j] = wlo + j;
        used[wlo + j] = true;
    }
    int k = 0;
    if (P == 0) {
        for (int b = LOGN - 1; b >= 0; b--)
            if (!used[b])
                L.tbit[k++] = b;
    } else {
        for (int b = 0; b < LOGN; b++)
            if (!used[b])
                L.tbit[k++] = b;
    }
    return L;
}

// index bits contributed by the thread id / by the register slot during pass P
template <int LOGN, int P>
SDR_HD inline int thread_part(int t)
{
    constexpr Layout L = make_layout<LOGN>(P);
    int r = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < Plan<LOGN>::LOGT; j++)
        r |= ((t >> j) & 1) << L.tbit[j];
    return r;
}
template <int LOGN, int P>
SDR_HD constexpr int slot_part(int s)
{
    constexpr Layout L = make_layout<LOGN>(P);
    int r = 0;
    for (int j = 0; j < Plan<LOGN>::LOGR; j++)
        r |= ((s >> j) & 1) << L.sbit[j];
    return r;
}

// Index (in the bit-reversed work array r[]) of register slot (u, m) of thread t during pass P.
template <int LOGN, int P>
SDR_HD inline int elem_index(int t, int u, int m)
{
    return thread_part<LOGN, P>(t) | slot_part<LOGN, P>((u << Plan<LOGN>::pass_log(P)) + m);
}

// ---------------------------------------------------------------------------------------------
// LDS address (in doubles) of index i during exchange E: a bit permutation of i with a 4-bit XOR
// swizzle, i.e. linear over GF(2), so addr(thread_part | slot_part) = addr(thread_part) ^ addr(slot_part)
// and the slot half is a compile-time constant.  Chosen per exchange so that neither side has bank
// conflicts (MI355X: a ds_write_b64 is served in groups of 16 consecutive lanes over 32 banks, a
// ds_read_b64 in groups of 32 lanes over 64 banks):
//   address bits 0-3 <- the index bits in the reader's lane bits 0-3, XORed with the index bits in the
//                       writer's lane bits 0-3 where those differ (exchange 0 is a transpose: the reader's
//                       low lane bits are the writer's slot bits)
//   address bit 4    <- the index bit in the reader's lane bit 4
//   the rest upwards.
// ---------------------------------------------------------------------------------------------
struct AddrMap {
    int abit[14];  // address bit k <- index bit abit[k]
    int xbit[4];   // address bit k (k < 4) additionally XORs index bit xbit[k] (-1: none)
};

// A wave-local exchange also puts the wave's bits at the top of the address: wave w then owns the
// words [w * 2^(LOGN-WB), (w+1) * 2^(LOGN-WB)) in every wave-local exchange, whichever index bits its
// id stands for at the time, so consecutive wave-local exchanges (also across frames of a persistent
// workgroup) need no barrier between them.
template <int LOGN>
SDR_HD constexpr AddrMap make_addr(int E)
{
    using PL = Plan<LOGN>;
    AddrMap A{};
    const Layout W = make_layout<LOGN>(E), Rd = make_layout<LOGN>(E + 1);
    bool used[16] = {};
    for (int k = 0; k < 5; k++) {
        A.abit[k] = Rd.tbit[k];
        used[Rd.tbit[k]] = true;
    }
    for (int k = 0; k < 4; k++) {
        bool in_reader_low = false;
        for (int j = 0; j < 4; j++)
            in_reader_low = in_reader_low || (W.tbit[k] == Rd.tbit[j]);
        A.xbit[k] = in_reader_low ? -1 : W.tbit[k];
    }
    if (!PL::cross_wave(E))
        for (int j = 0; j < PL::WB; j++) {
            A.abit[LOGN - PL::WB + j] = W.tbit[PL::LOGL + j];
            used[W.tbit[PL::LOGL + j]] = true;
        }
    int k = 5;
    for (int b = 0; b < LOGN; b++)
        if (!used[b])
            A.abit[k++] = b;
    return A;
}

template <int LOGN, int E>
SDR_HD constexpr int lds_addr(int i)
{
    constexpr AddrMap A = make_addr<LOGN>(E);
    int r = 0;
    for (int k = 0; k < LOGN; k++)
        r |= ((i >> A.abit[k]) & 1) << k;
    for (int k = 0; k < 4; k++)
        if (A.xbit[k] >= 0)
            r ^= ((i >> A.xbit[k]) & 1) << k;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Register exchange.  When exchange E only trades slot bits for the index bits held in lane bits 4
// and 5 (everything else stays where it is), it needs no LDS at all: gfx950's v_permlane16_swap /
// v_permlane32_swap exchange the upper half-rows / upper half of one register with the lower ones of
// another, which IS the swap of a register-index bit with lane bit 4 / 5.  64 VALU instructions per
// wave for the last exchange at N = 16384 instead of 64 LDS instructions, four dependent LDS round
// trips and a barrier.
// ---------------------------------------------------------------------------------------------
struct SwapPlan {
    bool ok;
    int slot_bit_lane4;  // slot bit that trades places with lane bit 4 (-1: lane bit 4 keeps its index bit)
    int slot_bit_lane5;
};

template <int LOGN>
SDR_HD constexpr SwapPlan make_swap_plan(int E)
{
    using PL = Plan<LOGN>;
    const Layout A = make_layout<LOGN>(E), B = make_layout<LOGN>(E + 1);
    SwapPlan p{true, -1, -1};
    for (int j = 0; j < PL::LOGT; j++)
        if (j != 4 && j != 5 && A.tbit[j] != B.tbit[j])
            p.ok = false;
    for (int j = 0; j < PL::LOGR; j++) {
        if (A.sbit[j] == B.sbit[j])
            continue;
        if (B.sbit[j] == A.tbit[4] && A.sbit[j] == B.tbit[4] && p.slot_bit_lane4 < 0)
            p.slot_bit_lane4 = j;
        else if (B.sbit[j] == A.tbit[5] && A.sbit[j] == B.tbit[5] && p.slot_bit_lane5 < 0)
            p.slot_bit_lane5 = j;
        else
            p.ok = false;
    }
    if (p.slot_bit_lane4 < 0 && A.tbit[4] != B.tbit[4])
        p.ok = false;
    if (p.slot_bit_lane5 < 0 && A.tbit[5] != B.tbit[5])
        p.ok = false;
    if (p.slot_bit_lane4 < 0 && p.slot_bit_lane5 < 0)
        p.ok = false;
#if defined(SDR_FFT_NO_SWAP)
    p.ok = false;  // diagnostic builds: every exchange through LDS
#endif
    return p;
}

// Host-side model of the same exchange for one wave (x[lane][slot]); tests/emu uses it.
template <int LOGN, int E>
inline void exchange_swap_wave(double (*x)[Plan<LOGN>::R])
{
    constexpr SwapPlan P = make_swap_plan<LOGN>(E);
    static_assert(P.ok, "exchange is not a register swap");
    for (int round = 0; round < 2; round++) {
        const int sb = round == 0 ? P.slot_bit_lane4 : P.slot_bit_lane5;
        const int half = round == 0 ? 16 : 32;
        if (sb < 0)
            continue;
        for (int a = 0; a < Plan<LOGN>::R; a++) {
            if ((a >> sb) & 1)
                continue;
            const int b = a | (1 << sb);
            // v_permlaneNN_swap A, B: A's lanes with the bit set <-> B's lanes with the bit clear
            for (int l = 0; l < 64; l++)
                if (!(l & half)) {
                    const double t = x[l + half][a];
                    x[l + half][a] = x[l][b];
                    x[l][b] = t;
                }
        }
    }
}

#if defined(__HIPCC__)
__device__ __forceinline__ void swap_halves(double &a, double &b, bool rows16)
{
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if (rows16) {
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        alo = lo[0], blo = lo[1], ahi = hi[0], bhi = hi[1];
    } else {
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        alo = lo[0], blo = lo[1], ahi = hi[0], bhi = hi[1];
    }
    a = __hiloint2double((int)ahi, (int)alo);
    b = __hiloint2double((int)bhi, (int)blo);
}

template <int LOGN, int E>
__device__ __forceinline__ void exchange_swap(double (&x)[Plan<LOGN>::R])
{
    constexpr SwapPlan P = make_swap_plan<LOGN>(E);
    static_assert(P.ok, "exchange is not a register swap");
#pragma unroll
    for (int a = 0; a < Plan<LOGN>::R; a++)
        if (P.slot_bit_lane4 >= 0 && !((a >> P.slot_bit_lane4) & 1))
            swap_halves(x[a], x[a | (1 << P.slot_bit_lane4)], true);
#pragma unroll
    for (int a = 0; a < Plan<LOGN>::R; a++)
        if (P.slot_bit_lane5 >= 0 && !((a >> P.slot_bit_lane5) & 1))
            swap_halves(x[a], x[a | (1 << P.slot_bit_lane5)], false);
}
#endif

// Sample number (natural order) held in slot m of thread t at the start of pass 0.
template <int LOGN>
SDR_HD inline int input_sample(int t, int m)
{
    return (int)brev_bits((unsigned)elem_index<LOGN, 0>(t, 0, m), LOGN);
}

// ---------------------------------------------------------------------------------------------
// Input staging through LDS.  Pass 0 wants lane neighbours 32 bytes apart in memory at best (make_layout),
// which costs the texture unit four times the cache-line look-ups of a contiguous read.  So the frame is
// first copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, each wave instruction one
// contiguous 1 KB row of 128 samples) and then read from LDS in the pass-0 layout.  LDS-DMA writes lane l's
// 16 bytes at row base + 16*l, so the image is shaped through the SOURCE address: lane p of row r fetches
// granule g = in_granule(p, r) (a granule = 2 samples = 16 bytes); the XOR swizzle spreads the rows and
// row halves that the 32 lanes of a ds_read_b64 group touch over the LDS banks.
//   sample n  ->  row r = n >> 7, granule g = (n >> 1) & 63, byte address r*1024 + in_pos(g, r)*16 + (n & 1)*8
// ---------------------------------------------------------------------------------------------
struct InSwz {
    int src[3];  // position bit 1+k is XORed with: -1 nothing, 0..5 granule bit, 8+j row bit j
};

template <int LOGN>
SDR_HD constexpr InSwz make_in_swz()
{
    const Layout L0 = make_layout<LOGN>(0);
    InSwz z{{-1, -1, -1}};
    for (int k = 0; k < 3; k++) {
        const int nb = LOGN - 1 - L0.tbit[2 + k];  // sample-number bit behind lane bit 2+k
        if (nb >= 7)
            z.src[k] = 8 + (nb - 7);
        else if (nb - 1 != 1 + k)
            z.src[k] = nb - 1;
    }
    return z;
}

template <int LOGN>
SDR_HD constexpr bool in_swz_solvable()
{
    const InSwz z = make_in_swz<LOGN>();
    for (int k = 0; k < 3; k++)
        if (z.src[k] >= 0 && z.src[k] < 8 && z.src[k] <= 1 + k)
            return false;  // a position bit may only depend on higher granule bits (in_granule solves downwards)
    return true;
}

template <int LOGN>
SDR_HD inline int in_pos(int g, int r)
{
    constexpr InSwz Z = make_in_swz<LOGN>();
    int p = g;
    for (int k = 0; k < 3; k++) {
        if (Z.src[k] >= 8)
            p ^= ((r >> (Z.src[k] - 8)) & 1) << (1 + k);
        else if (Z.src[k] >= 0)
            p ^= ((g >> Z.src[k]) & 1) << (1 + k);
    }
    return p;
}

template <int LOGN>
SDR_HD inline int in_granule(int p, int r)
{
    static_assert(in_swz_solvable<LOGN>(), "input swizzle is not triangular");
    constexpr InSwz Z = make_in_swz<LOGN>();
    int g = p;
    for (int k = 2; k >= 0; k--) {
        if (Z.src[k] >= 8)
            g ^= ((r >> (Z.src[k] - 8)) & 1) << (1 + k);
        else if (Z.src[k] >= 0)
            g ^= ((g >> Z.src[k]) & 1) << (1 + k);  // bit src[k] > 1+k is final already
    }
    return g;
}

// byte address in the staging image of sample n
template <int LOGN>
SDR_HD inline int in_lds_byte(int n)
{
    const int r = n >> 7, g = (n >> 1) & 63;
    return r * 1024 + in_pos<LOGN>(g, r) * 16 + (n & 1) * 8;
}

// Pass 0 input: slot m <- x[input_sample(t, m)], widened to float64
// (dsp/fft.go:59-69 setSamplesFromIQ).
template <int LOGN>
SDR_HD inline void load_input(const float *iq, int t, double *xr, double *xi)
{
    using PL = Plan<LOGN>;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int m = 0; m < PL::R; m++) {
        const int n = input_sample<LOGN>(t, m);
        xr[m] = (double)iq[2 * n];
        xi[m] = (double)iq[2 * n + 1];
    }
}

// Pass P: pass_log(P) radix-2 stages on the thread's registers.  `tw_at(c, lo)` returns twiddle table
// entry c + lo, where c is a compile-time constant and lo the only per-thread part (the kernel turns that
// split into a buffer load with c in the scalar offset; the emulator indexes an array).
template <int LOGN, int P, class TW>
SDR_HD inline void butterfly_pass(double *xr, double *xi, int t, TW tw_at)
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
        const int lo = (P == 0) ? 0 : (elem_index<LOGN, P>(t, u, 0) & (S - 1));
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
                    const cplx w = tw_at(OFF + ((1 << q) - 1 + mm) * S, lo);
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
    const int base = lds_addr<LOGN, E>(thread_part<LOGN, E>(t));
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int s = 0; s < PL::R; s++)
        lds[base ^ lds_addr<LOGN, E>(slot_part<LOGN, E>(s))] = x[s];
}

// Exchange E, read side: gather the slots of pass E+1.
template <int LOGN, int E>
SDR_HD inline void exchange_read(double *x, int t, const double *lds)
{
    using PL = Plan<LOGN>;
    const int base = lds_addr<LOGN, E>(thread_part<LOGN, E + 1>(t));
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int s = 0; s < PL::R; s++)
        x[s] = lds[base ^ lds_addr<LOGN, E>(slot_part<LOGN, E + 1>(s))];
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
