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

// Two assignments of index bits to (register slot, lane, wave) - make_layout below:
//  A  passes 0,1 share their wave bits, the cross-wave exchange sits after pass 1; the frame is staged through LDS
//     (LDS-DMA) and read in the pass-0 layout.  All sizes below SDR_FFT_LAYOUT_B_FROM.
//  B  the cross-wave exchange sits after pass 0; from then on wave w owns the sub-problem "index bits 0-3 = w".
//     Pass 0's lanes are the six lowest sample-number bits, so a wave loads 512 contiguous bytes per register slot
//     straight from memory - no LDS staging, the next frame is prefetched into registers - and pass 1's twiddles depend
//     on the wave only (scalar loads).  Needs the wave id to cover a whole pass's worth of bits (WB == LOGR): N = 16384.
//     MEASURED (round 3, tools/fft_bench, 2048 frames of 16384): bit-identical, but no faster - 0.168 ms per launch
//     against layout A's 0.165.  What it gains (the next frame's LDS-DMA under passes 1-2: A loses 0.050 ms to the input
//     wait, B 0.032) it pays back: the register-only exchanges (+128 v_permlane, 128 ds_bpermute per wave), a
//     transposed epilogue through LDS (wave w holds the bins = w mod 16; stored from registers they cost 0.087 ms)
//     and waves that drift apart over the long barrier-free stretch.  And its workgroups must take several frames
//     each, which starves the tail stages of CUs inside the pipeline (0.273 ms per step against 0.216).  So it is
//     OFF by default (SDR_FFT_LAYOUT_B_FROM = 15 selects no size); -DSDR_FFT_LAYOUT_B_FROM=14 builds it, tests/emu
//     checks both layouts on the CPU.
#if !defined(SDR_FFT_LAYOUT_B_FROM)
#define SDR_FFT_LAYOUT_B_FROM 15
#endif
SDR_HD constexpr bool layout_b(int logn) { return logn >= SDR_FFT_LAYOUT_B_FROM; }

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
    // (bytes of LDS a frame needs: fft64::lds_bytes<LOGN>() - the exchange maps are padded, see make_addr)
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
    static constexpr bool LB = layout_b(LOGN);
    static_assert(!LB || WB == LOGR, "layout B needs the wave id to hold exactly one pass's bits");
    // Exchange e (between pass e and e+1) moves data between waves only when the set of index bits
    // held in the wave id changes; that happens once: after pass 1 in layout A, after pass 0 in layout B
    // (see make_layout).  Every other exchange stays inside a wave and needs no workgroup barrier.
    SDR_HD static constexpr bool cross_wave(int e) { return WB > 0 && e == (LB ? 0 : 1); }
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
    // layout B: pass 0 keeps pass 1's slot bits in the wave id, every later pass the bits of pass 0's slots
    const int wlo = PL::LB ? (P == 0 ? PL::LOGR : 0) : (P <= 1) ? 2 * PL::LOGR : wz_hi - PL::WB;
    for (int j = 0; j < PL::WB; j++) {
        L.tbit[PL::LOGL + j] = wlo + j;
        used[wlo + j] = true;
    }
    int k = 0;
    if (PL::LB && P >= 1) {
        // Layout B keeps LDS for the one cross-wave exchange (and the next frame's samples): behind it index bits
        // move between slots and lanes in registers only - v_permlane16/32_swap trade a slot bit for lane bit 4 / 5,
        // ds_bpermute (the LDS crossbar, no LDS memory) rotates other lane bits into those two positions
        // (make_reg_plan).  The lane orders below are the ones those steps produce.
        constexpr int kLanesB[4][6] = {{0, 0, 0, 0, 0, 0}, {10, 11, 12, 13, 8, 9}, {4, 5, 12, 13, 6, 7}, {4, 5, 6, 7, 8, 9}};
        for (int j = 0; j < PL::LOGL; j++)
            L.tbit[j] = kLanesB[P][j];
    } else if (P == 0) {
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

// ---------------------------------------------------------------------------------------------
// Where a pass's twiddles sit in its table.  Pass P's twiddle for a butterfly depends on the index bits below
// the pass's own (lo = i mod 2^(P*LOGR)) and on the row (stage and position inside the register group).  Within a
// row the entries are stored in the order the THREADS want them: position bit 0 is the lowest of those index bits
// that sits in a lane, and so on through lanes, waves and (short last pass) the slot's spare bits - so that
// neighbouring lanes read neighbouring entries whatever the layout.
// ---------------------------------------------------------------------------------------------
struct TwPerm {
    int pos[16];  // index bit b (< P*LOGR) -> position bit
};

template <int LOGN>
SDR_HD constexpr TwPerm make_tw_perm(int P)
{
    using PL = Plan<LOGN>;
    const Layout L = make_layout<LOGN>(P);
    const int SH = P * PL::LOGR;
    TwPerm T{};
    int k = 0;
    for (int j = 0; j < PL::LOGT; j++)
        if (L.tbit[j] < SH)
            T.pos[L.tbit[j]] = k++;
    for (int j = 0; j < PL::LOGR; j++)
        if (L.sbit[j] < SH)
            T.pos[L.sbit[j]] = k++;
    return T;
}

// position of the entry for lo (host side, table construction)
template <int LOGN, int P>
SDR_HD constexpr int tw_pos_of_lo(int lo)
{
    constexpr TwPerm T = make_tw_perm<LOGN>(P);
    int r = 0;
    for (int b = 0; b < P * Plan<LOGN>::LOGR; b++)
        r |= ((lo >> b) & 1) << T.pos[b];
    return r;
}

// Index (in the bit-reversed work array r[]) of register slot (u, m) of thread t during pass P.
template <int LOGN, int P>
SDR_HD inline int elem_index(int t, int u, int m)
{
    return thread_part<LOGN, P>(t) | slot_part<LOGN, P>((u << Plan<LOGN>::pass_log(P)) + m);
}

// position (inside a row of pass P's twiddle table) of the entry thread t needs for register group u
template <int LOGN, int P>
SDR_HD inline int tw_pos(int t, int u)
{
    using PL = Plan<LOGN>;
    constexpr Layout L = make_layout<LOGN>(P);
    constexpr TwPerm T = make_tw_perm<LOGN>(P);
    constexpr int SH = P * PL::LOGR;
    int r = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < PL::LOGT; j++)
        if (L.tbit[j] < SH)
            r |= ((t >> j) & 1) << T.pos[L.tbit[j]];
    const int s = u << PL::pass_log(P);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < PL::LOGR; j++)
        if (L.sbit[j] < SH)
            r |= ((s >> j) & 1) << T.pos[L.sbit[j]];
    return r;
}

// ---------------------------------------------------------------------------------------------
// LDS address (in doubles) of index i during exchange E: a WEIGHTED SUM of the index bits, so that
// addr(thread_part | slot_part) = addr(thread_part) + addr(slot_part) and the slot half is a compile-time constant
// that goes into the DS instruction's offset field - one address register per side of an exchange (two where the
// constants pass 64 KB) instead of one per register slot (the XOR-swizzled map of rounds 1-2 cost the kernel about
// thirty address registers at the moment its register demand peaks).  The weights are chosen per exchange so that
// neither side has bank conflicts (MI355X: a ds_write_b64 is served in groups of 16 consecutive lanes over 32 banks,
// a ds_read_b64 in groups of 32 lanes over 64 banks), by padding rather than by swizzling:
//   the index bits in the reader's lane bits 0-4 get the weights 1, 2, 4, 8, 16 (32 lanes: 32 different words mod 32),
//     those that also sit in the writer's lane bits 0-3 first;
//   the remaining index bits of the writer's lane bits 0-3 get 32 * 2^j + (an unused one of 1, 2, 4, 8), so the
//     writer's 16 lanes fall on 16 different words mod 16 as well;
//   everything else is a multiple of the range those bits span, in this order: the writer's slot bits, the reader's
//     slot bits (small constants in the offset fields), the rest - and, in a wave-local exchange, the wave's bits last:
//     wave w then owns one contiguous block of words in every wave-local exchange, whichever index bits its id stands
//     for at the time, so consecutive wave-local exchanges need no barrier between them.
// ---------------------------------------------------------------------------------------------
struct AddrMap {
    int w[16];   // weight of index bit b, in doubles
    int size;    // words the exchange area spans
    int block;   // wave-local exchange: words per wave (0: cross-wave)
};

template <int LOGN>
SDR_HD constexpr AddrMap make_addr(int E)
{
    using PL = Plan<LOGN>;
    AddrMap A{};
    const Layout W = make_layout<LOGN>(E), Rd = make_layout<LOGN>(E + 1);
    bool used[16] = {};
    constexpr int NR = PL::LOGT < 5 ? PL::LOGT : 5;  // reader lane bits that share a ds_read_b64 group
    constexpr int NW = 4;                            // writer lane bits that share a ds_write_b64 group
    // 1. reader's low lane bits: powers of two, the ones shared with the writer's low lane bits first
    int k = 0;
    bool low_res_used[NW] = {};  // which of the residues 1, 2, 4, 8 (mod 16) the writer's lanes already have
    for (int pass = 0; pass < 2; pass++)
        for (int j = 0; j < NR; j++) {
            const int b = Rd.tbit[j];
            bool shared = false;
            for (int i = 0; i < NW; i++)
                shared = shared || (W.tbit[i] == b);
            if (used[b] || (pass == 0) != shared)
                continue;
            A.w[b] = 1 << k;
            if (shared && k < NW)
                low_res_used[k] = true;
            used[b] = true;
            k++;
        }
    int range = 1 << k;
    // 2. the writer's other low lane bits: 32 * 2^j plus an unused residue
    {
        int j = 0;
        for (int i = 0; i < NW; i++) {
            const int b = W.tbit[i];
            if (used[b])
                continue;
            int r = 0;
            while (r < NW && low_res_used[r])
                r++;
            low_res_used[r < NW ? r : NW - 1] = true;
            A.w[b] = ((1 << k) << j) + (1 << r);
            range = A.w[b] + range;  // (weights grow: the span so far is the sum of all weights + 1)
            used[b] = true;
            j++;
        }
    }
    // span of the bits placed so far = 1 + sum of their weights
    {
        int sum = 0;
        for (int b = 0; b < LOGN; b++)
            if (used[b])
                sum += A.w[b];
        range = sum + 1;
    }
    // 3. everything else: multiples of that range; writer's slot bits, reader's slot bits, the rest, wave bits last
    int mult = 1;
    auto place = [&](int b) {
        if (b < 0 || b >= LOGN || used[b])
            return;
        A.w[b] = range * mult;
        mult <<= 1;
        used[b] = true;
    };
    const bool local = !PL::cross_wave(E);
    bool is_wave[16] = {};
    if (local)
        for (int j = 0; j < PL::WB; j++)
            is_wave[W.tbit[PL::LOGL + j]] = true;
    for (int j = 0; j < PL::LOGR; j++)
        if (!is_wave[W.sbit[j]])
            place(W.sbit[j]);
    for (int j = 0; j < PL::LOGR; j++)
        if (!is_wave[Rd.sbit[j]])
            place(Rd.sbit[j]);
    for (int b = 0; b < LOGN; b++)
        if (!is_wave[b])
            place(b);
    A.block = local ? range * mult : 0;
    for (int j = 0; j < PL::WB; j++)
        if (local)
            place(W.tbit[PL::LOGL + j]);
    A.size = range * mult;
    return A;
}

template <int LOGN, int E>
SDR_HD constexpr int lds_addr(int i)
{
    constexpr AddrMap A = make_addr<LOGN>(E);
    int r = 0;
    for (int b = 0; b < LOGN; b++)
        r += ((i >> b) & 1) * A.w[b];
    return r;
}

// the same for the thread's part of an index during pass P (run-time argument: unrolled over the thread-id bits, the
// weights become literals)
template <int LOGN, int E, int P>
SDR_HD inline int lds_addr_thread(int t)
{
    constexpr AddrMap A = make_addr<LOGN>(E);
    constexpr Layout L = make_layout<LOGN>(P);
    int r = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < Plan<LOGN>::LOGT; j++)
        r += ((t >> j) & 1) * A.w[L.tbit[j]];
    return r;
}

// words the exchanges of a frame need (the largest of the LDS exchanges' spans)
template <int LOGN>
SDR_HD constexpr int exchange_words()
{
    using PL = Plan<LOGN>;
    int m = PL::N;
    for (int e = 0; e < PL::NPASS - 1; e++) {
        const int sz = make_addr<LOGN>(e).size;
        m = sz > m ? sz : m;
    }
    return m;
}

// (as variables: a constexpr FUNCTION called where a constant is not required may be compiled and called at run time)
template <int LOGN>
inline constexpr int kExchangeWords = exchange_words<LOGN>();
template <int LOGN>
inline constexpr int kLdsBytes = (Plan<LOGN>::SPLIT ? 1 : 2) * kExchangeWords<LOGN> * 8;
template <int LOGN>
SDR_HD constexpr int lds_bytes()
{
    return kLdsBytes<LOGN>;
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

// ---------------------------------------------------------------------------------------------
// Register exchanges in general: a short sequence of steps on a wave's x[lane][slot],
//   SWAP(a, b)  slot bit a <-> lane bit 4 and slot bit b <-> lane bit 5 (either may be -1: left alone)
//   ROT(p)      lane positions (p, p+1) <-> (4, 5): every register moves to the lane with those bits exchanged
// make_reg_plan(E).n == 0: exchange E goes through LDS.
// ---------------------------------------------------------------------------------------------
struct RegStep {
    int rot;   // 0: SWAP, 1: ROT
    int a, b;  // SWAP: slot bits for lane bits 4 / 5; ROT: a = p
};
struct RegPlan {
    int n;
    RegStep st[3];
};

template <int LOGN>
SDR_HD constexpr RegPlan make_reg_plan(int E)
{
    using PL = Plan<LOGN>;
    RegPlan R{0, {}};
#if defined(SDR_FFT_B_LDS_EXCH)
    // (experiment, round 4: layout B with its two wave-local exchanges through the wave's own LDS block instead of
    // v_permlane / ds_bpermute.  VALID FOR ONE-FRAME WORKGROUPS ONLY (SDR_FFT_FPW=1): the blocks overlap the next frame's
    // staging image.  Measured, 2048 x 16384 standalone: 0.169 ms (registers: 0.185 at one frame per workgroup, 0.168 at
    // eight) against layout A's 0.162 - the cross-wave exchange right behind pass 0 still waits for the last wave's
    // samples, and the transposed epilogue costs what the barrier-free second half gains.  With the next frame's DMA
    // issued as if LDS were free (wrong results, timing only) 0.155: the price of that LDS is what separates the two.)
    if (PL::LB)
        return R;
#endif
    if (PL::LB) {
        if (E == 1) {
            R.n = 3;
            R.st[0] = RegStep{0, 0, 1};
            R.st[1] = RegStep{1, 0, 0};
            R.st[2] = RegStep{0, 2, 3};
        } else if (E == 2) {
            R.n = 2;
            R.st[0] = RegStep{1, 2, 0};
            R.st[1] = RegStep{0, 0, 1};
        }
        return R;
    }
    const SwapPlan S = make_swap_plan<LOGN>(E);
    if (S.ok) {
        R.n = 1;
        R.st[0] = RegStep{0, S.slot_bit_lane4, S.slot_bit_lane5};
    }
    return R;
}

// does the plan turn pass E's layout into pass E+1's?  (bit bookkeeping only)
template <int LOGN>
SDR_HD constexpr bool reg_plan_valid(int E)
{
    using PL = Plan<LOGN>;
    const RegPlan R = make_reg_plan<LOGN>(E);
    if (R.n == 0)
        return true;
    Layout A = make_layout<LOGN>(E);
    const Layout B = make_layout<LOGN>(E + 1);
    for (int i = 0; i < R.n; i++) {
        const RegStep &st = R.st[i];
        if (st.rot) {
            for (int j = 0; j < 2; j++) {
                const int t = A.tbit[st.a + j];
                A.tbit[st.a + j] = A.tbit[4 + j];
                A.tbit[4 + j] = t;
            }
        } else {
            if (st.a >= 0) {
                const int t = A.sbit[st.a];
                A.sbit[st.a] = A.tbit[4];
                A.tbit[4] = t;
            }
            if (st.b >= 0) {
                const int t = A.sbit[st.b];
                A.sbit[st.b] = A.tbit[5];
                A.tbit[5] = t;
            }
        }
    }
    for (int j = 0; j < PL::LOGR; j++)
        if (A.sbit[j] != B.sbit[j])
            return false;
    for (int j = 0; j < PL::LOGT; j++)
        if (A.tbit[j] != B.tbit[j])
            return false;
    return true;
}

// lane a register comes from in ROT(p): the lane number with bit positions (p, p+1) and (4, 5) exchanged
SDR_HD constexpr int rot_source_lane(int lane, int p)
{
    const int lo = (lane >> p) & 3, hi = (lane >> 4) & 3;
    return (lane & ~((3 << p) | (3 << 4))) | (hi << p) | (lo << 4);
}

// Host-side model of a register exchange for one wave (x[lane][slot]); tests/emu uses it.
template <int LOGN, int E>
inline void exchange_regs_wave(double (*x)[Plan<LOGN>::R])
{
    constexpr RegPlan P = make_reg_plan<LOGN>(E);
    static_assert(P.n > 0 && reg_plan_valid<LOGN>(E), "exchange is not a valid register plan");
    constexpr int R = Plan<LOGN>::R;
    for (int i = 0; i < P.n; i++) {
        const RegStep st = P.st[i];
        if (st.rot) {
            double tmp[64][R];
            for (int l = 0; l < 64; l++)
                for (int s = 0; s < R; s++)
                    tmp[l][s] = x[rot_source_lane(l, st.a)][s];
            for (int l = 0; l < 64; l++)
                for (int s = 0; s < R; s++)
                    x[l][s] = tmp[l][s];
            continue;
        }
        for (int round = 0; round < 2; round++) {
            const int sb = round == 0 ? st.a : st.b;
            const int half = round == 0 ? 16 : 32;
            if (sb < 0)
                continue;
            for (int a = 0; a < R; a++) {
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

#if defined(__HIPCC__)
// Device: the register exchange of plan E on one array of R doubles (called once for the real, once for the
// imaginary parts).
template <int LOGN, int E>
__device__ __forceinline__ void exchange_regs(double (&x)[Plan<LOGN>::R])
{
    constexpr RegPlan P = make_reg_plan<LOGN>(E);
    static_assert(P.n > 0 && reg_plan_valid<LOGN>(E), "exchange is not a valid register plan");
    constexpr int R = Plan<LOGN>::R;
#pragma unroll
    for (int i = 0; i < P.n; i++) {
        if (P.st[i].rot) {
            const int lane = (int)(threadIdx.x & 63);
            const int src = rot_source_lane(lane, P.st[i].a) * 4;
#pragma unroll
            for (int a = 0; a < R; a++) {
                const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(x[a]));
                const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(x[a]));
                x[a] = __hiloint2double(hi, lo);
            }
        } else {
#pragma unroll
            for (int a = 0; a < R; a++)
                if (P.st[i].a >= 0 && !((a >> P.st[i].a) & 1))
                    swap_halves(x[a], x[a | (1 << P.st[i].a)], true);
#pragma unroll
            for (int a = 0; a < R; a++)
                if (P.st[i].b >= 0 && !((a >> P.st[i].b) & 1))
                    swap_halves(x[a], x[a | (1 << P.st[i].b)], false);
        }
    }
}
#endif

// sample-number part contributed by register slot m at the start of pass 0 (the thread's part is input_sample(t, 0))
template <int LOGN>
SDR_HD constexpr int input_slot_sample(int m)
{
    const int i = slot_part<LOGN, 0>(m);
    int r = 0;
    for (int b = 0; b < LOGN; b++)
        r |= ((i >> b) & 1) << (LOGN - 1 - b);
    return r;
}

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
// entry c + lo, where c (the row) is a compile-time constant and lo = tw_pos(t, u) the only per-thread part (the
// kernel turns that split into a buffer load with c in the scalar offset; the emulator indexes an array).
// CHUNK > 0 (device, layout B): at most CHUNK twiddle rows are requested at a time and a scheduling fence behind
// their butterflies keeps the compiler from pulling the next rows' loads forward - the register-staged kernel has 16
// registers for twiddles (the next frame's samples occupy 32), not the 60 a whole pass's rows take when they are all
// in flight at once.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define SDR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SDR_SCHED_FENCE() \
    do {                  \
    } while (0)
#endif
// stage Q of pass P on register group u (compile-time loop over the stages: every bound below is a constant
// expression, so the twiddle arrays and the data stay in registers)
template <int LOGN, int P, int CHUNK, int Q, class TW>
SDR_HD inline void butterfly_stages(double *xr, double *xi, int u, int lo, TW &tw_at)
{
    using PL = Plan<LOGN>;
    constexpr int PLOG = PL::pass_log(P);
    if constexpr (Q < PLOG) {
        constexpr int RP = 1 << PLOG;
        constexpr int S = 1 << (P * PL::LOGR);
        constexpr int OFF = PL::tw_offset(P);
        constexpr int ROWS = 1 << Q;
        constexpr int CH = (CHUNK > 0 && CHUNK < ROWS) ? CHUNK : ROWS;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int c = 0; c < ROWS / CH; c++) {
            double wr[CH], wi[CH];
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < CH; j++) {
                const int mm = c * CH + j;
                // pass 0 has thread-independent twiddles; W[0] = 1 and W[N/4] = -i are the literal
                // entries of go-dsp's size-4 table, multiplying by them is exact (up to the sign of a
                // zero, which cannot reach |X|^2), so the multiply is skipped.
                const bool one = (P == 0 && mm == 0);
                const bool minus_i = (P == 0 && Q >= 1 && mm == (ROWS >> 1));
                wr[j] = 1.0;
                wi[j] = 0.0;
                if (!one && !minus_i) {
#if defined(SDR_ABLATE) && (SDR_ABLATE == 3)
                    wr[j] = 0.5 + 1e-9 * lo;  // timing-only build: no twiddle loads
                    wi[j] = 0.25;
#else
                    const cplx w = tw_at(OFF + (ROWS - 1 + mm) * S, lo);
                    wr[j] = w.x;
                    wi[j] = w.y;
#endif
                }
            }
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < CH; j++) {
                const int mm = c * CH + j;
                const bool one = (P == 0 && mm == 0);
                const bool minus_i = (P == 0 && Q >= 1 && mm == (ROWS >> 1));
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (int k = 0; k < (RP >> (Q + 1)); k++) {
                    const int a = u * RP + mm + (k << (Q + 1));
                    const int b = a + ROWS;
                    double tr, ti;
                    if (one) {
                        tr = xr[b];
                        ti = xi[b];
                    } else if (minus_i) {
                        tr = xi[b];
                        ti = -xr[b];
                    } else {
                        tr = xr[b] * wr[j] - xi[b] * wi[j];  // Go complex128 multiply, amd64: no FMA
                        ti = xr[b] * wi[j] + xi[b] * wr[j];
                    }
                    const double ar = xr[a], ai = xi[a];
                    xr[a] = ar + tr;
                    xi[a] = ai + ti;
                    xr[b] = ar - tr;
                    xi[b] = ai - ti;
                }
            }
            if (CHUNK > 0)
                SDR_SCHED_FENCE();
        }
        butterfly_stages<LOGN, P, CHUNK, Q + 1>(xr, xi, u, lo, tw_at);
    }
}

template <int LOGN, int P, int CHUNK = 0, class TW>
SDR_HD inline void butterfly_pass(double *xr, double *xi, int t, TW tw_at)
{
    using PL = Plan<LOGN>;
    constexpr int G = PL::R >> PL::pass_log(P);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int u = 0; u < G; u++) {
        const int lo = (P == 0) ? 0 : tw_pos<LOGN, P>(t, u);  // position inside the row, see make_tw_perm
        butterfly_stages<LOGN, P, CHUNK, 0>(xr, xi, u, lo, tw_at);
    }
}

// Exchange E, write side: scatter the thread's slots (pass E layout) into LDS.
template <int LOGN, int E>
SDR_HD inline void exchange_write(const double *x, int t, double *lds)
{
    using PL = Plan<LOGN>;
    const int base = lds_addr_thread<LOGN, E, E>(t);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int s = 0; s < PL::R; s++)
        lds[base + lds_addr<LOGN, E>(slot_part<LOGN, E>(s))] = x[s];
}

// Exchange E, read side: gather the slots of pass E+1.
template <int LOGN, int E>
SDR_HD inline void exchange_read(double *x, int t, const double *lds)
{
    using PL = Plan<LOGN>;
    const int base = lds_addr_thread<LOGN, E, E + 1>(t);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int s = 0; s < PL::R; s++)
        x[s] = lds[base + lds_addr<LOGN, E>(slot_part<LOGN, E + 1>(s))];
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
// entry OFF_p + ((2^q - 1) + mm) * S_p + pos_p(lo)  =  W[(N / (2 * S_p * 2^q)) * (mm * S_p + lo)]
// (pos_p: the order the threads of pass p read a row in, make_tw_perm).
template <int LOGN, int P>
inline void build_one_pass_twiddles(const double *wre, const double *wim, cplx *out)
{
    using PL = Plan<LOGN>;
    if constexpr (P < PL::NPASS) {
        const int S = 1 << (P * PL::LOGR);
        const int off = PL::tw_offset(P);
        for (int q = 0; q < PL::pass_log(P); q++)
            for (int mm = 0; mm < (1 << q); mm++)
                for (int lo = 0; lo < S; lo++) {
                    const int h = S << q;
                    const int k = (PL::N / (2 * h)) * (mm * S + lo);
                    out[off + ((1 << q) - 1 + mm) * S + tw_pos_of_lo<LOGN, P>(lo)] = cplx{wre[k], wim[k]};
                }
        build_one_pass_twiddles<LOGN, P + 1>(wre, wim, out);
    }
}
template <int LOGN>
inline void build_pass_twiddles(const double *wre, const double *wim, cplx *out)
{
    build_one_pass_twiddles<LOGN, 0>(wre, wim, out);
}

}  // namespace fft64
