// fft_r32.h — the 32-points-per-thread plan of the float64 radix-2 DIT FFT at N = 16384 (dsp/fft.go:26, go-dsp fft.FFT):
// 512 threads x 32 points, three register passes of 5 + 5 + 4 stages, two exchanges through LDS.  Same dataflow graph,
// same IEEE operations and same twiddle VALUES as fft_f64.h (whose header explains why that gives the reference's bits);
// what differs is who holds what and when:
//
//   * A thread's 32 complex128 points are 128 VGPRs; at two waves per SIMD a wave has 256, so the NEXT frame's samples
//     (32 complex64 = 64 VGPRs) are fetched into registers while the current frame computes - the input wait that is a
//     third of a frame in the 16-point kernel (k_fft_psd.hip) overlaps the arithmetic, without any LDS for it.
//   * Pass 0's slots are the five highest sample-number bits, its lanes the six lowest: every wave instruction of the
//     prefetch reads 512 contiguous bytes, each 128-byte line is asked for by exactly one wave.
//   * The cross-wave exchange sits behind pass 0 (E0); passes 1 and 2 share their wave bits, E1 stays inside a wave's own
//     block of LDS.  The psd row leaves through LDS (transposed there into 16-byte runs per lane), which also serves the
//     listeners' tap.
//   * Pass 0's twiddles are wave-uniform (scalar loads), pass 1's 992 entries live in LDS for the workgroup's lifetime,
//     pass 2's stream from L2 in the order the threads read them.
//
// Everything here is SDR_HD and free of HIP intrinsics: tests/emu/emu_fft_r32.cpp runs these very functions thread by
// thread on the CPU against the oracle and audits every LDS map against the MI355X banking rules.
#pragma once
#include <cstdint>

#include "fft_f64.h"

namespace fft32 {

using fft64::cplx;

constexpr int LOGN = 14;
constexpr int N = 1 << LOGN;
constexpr int LOGR = 5;
constexpr int R = 32;          // points per thread
constexpr int T = N / R;       // 512 threads = 8 waves
constexpr int NWAVES = T / 64;
constexpr int NPASS = 3;
SDR_HD constexpr int pass_stages(int P) { return P < 2 ? 5 : 4; }
// first index bit paired by pass P's stages
SDR_HD constexpr int pass_base(int P) { return 5 * P; }

// Index bit i (of the position in the bit-reversed work array) <-> natural sample-number bit 13 - i.
//   pass 0: slots = bits 0-4 (sample bits 13..9); lanes = sample bits 0-5 (index bits 13..8); waves = sample bits 6-8
//   pass 1: slots = bits 5-9; lanes = bits 10,11,12,13,0,1; waves = bits 2,3,4
//   pass 2: slots = bits 10,11,12,13 + bit 9 as a passenger; lanes = bits 0,1,5,6,7,8; waves = bits 2,3,4
struct Layout {
    int sbit[5];  // index bit held by slot bit j
    int tbit[9];  // index bit held by thread-id bit j (six lane bits, then three wave bits)
};
// (functions, not namespace-scope tables: device code would otherwise load the tables from memory at run time)
SDR_HD constexpr Layout layout(int P)
{
    return P == 0   ? Layout{{0, 1, 2, 3, 4}, {13, 12, 11, 10, 9, 8, 7, 6, 5}}
           : P == 1 ? Layout{{5, 6, 7, 8, 9}, {10, 11, 12, 13, 0, 1, 2, 3, 4}}
                    : Layout{{10, 11, 12, 13, 9}, {0, 1, 5, 6, 7, 8, 2, 3, 4}};
}

template <int P>
SDR_HD inline int thread_part(int t)
{
    constexpr Layout L = layout(P);
    int r = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++)
        r |= ((t >> j) & 1) << L.tbit[j];
    return r;
}
SDR_HD constexpr int slot_part(int P, int s)
{
    const Layout L = layout(P);
    int r = 0;
    for (int j = 0; j < 5; j++)
        r |= ((s >> j) & 1) << L.sbit[j];
    return r;
}

// natural sample number of index i
SDR_HD constexpr int sample_of_index(int i)
{
    int r = 0;
    for (int b = 0; b < LOGN; b++)
        r |= ((i >> b) & 1) << (LOGN - 1 - b);
    return r;
}
// pass 0: slot m of thread t holds sample thread_sample(t) + slot_sample(m)
SDR_HD inline int thread_sample(int t) { return sample_of_index(thread_part<0>(t)); }
SDR_HD constexpr int slot_sample(int m) { return sample_of_index(slot_part(0, m)); }

// ---------------------------------------------------------------------------------------------
// LDS maps: an element's address is a weighted sum of its index bits, so address = (thread's part, one VGPR) +
// (slot's part, a compile-time constant in the DS instruction's offset field).  Banking (MI355X_MICROARCH.md, LDS): a
// ds_write_b64 is served in groups of 16 consecutive lanes over 32 four-byte banks - the group's sixteen 8-byte words
// must differ mod 16 - a ds_read_b64 in groups of 32 lanes over 64 banks: thirty-two words that differ mod 32.
//   E0 (cross-wave, pass 0 -> pass 1): the writer's lane bits 0-3 hold index bits 13,12,11,10, the reader's lane bits
//       0-4 hold 10,11,12,13,0: weights 1,2,4,8 for bits 10-13, 16 for bit 0, then plain powers of two: 16384 words,
//       no padding, conflict-free on both sides.
//   E1 (wave-local, pass 1 -> pass 2): the reader's lane bits 0-4 hold 0,1,5,6,7 (weights 1,2,4,8,16), the writer's lane
//       bits 0-3 hold 10-13 (33,66,132,264: residues 1,2,4,8 mod 16), bits 8 and 9 above them (527, 1054); bits 2-4 are
//       the wave id on both sides: wave w owns words [2108 w, 2108 (w+1)).
// ---------------------------------------------------------------------------------------------
struct AddrMap {
    int w[LOGN];
};
inline constexpr int kE1Block = 2108;
SDR_HD constexpr AddrMap addr_map(int E)
{
    return E == 0 ? AddrMap{{16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 1, 2, 4, 8}}
                  : AddrMap{{1, 2, kE1Block, 2 * kE1Block, 4 * kE1Block, 4, 8, 16, 527, 1054, 33, 66, 132, 264}};
}
inline constexpr int kE0Words = N;
inline constexpr int kE1Words = NWAVES * kE1Block;
inline constexpr int kExchangeBytes = (kE1Words > kE0Words ? kE1Words : kE0Words) * 8;

SDR_HD constexpr int map_addr(int E, int i)
{
    const AddrMap A = addr_map(E);
    int r = 0;
    for (int b = 0; b < LOGN; b++)
        r += ((i >> b) & 1) * A.w[b];
    return r;
}
template <int E, int P>
SDR_HD inline int map_addr_thread(int t)
{
    constexpr AddrMap A = addr_map(E);
    constexpr Layout L = layout(P);
    int r = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++)
        r += ((t >> j) & 1) * A.w[L.tbit[j]];
    return r;
}
template <int E>
SDR_HD constexpr int map_addr_slot(int P, int s)
{
    return map_addr(E, slot_part(P, s));
}

// The psd row in LDS: float32 word of spectrum index k (fft-shifted: k = bin ^ N/2) sits at word k with bits 2-4 XORed
// by bits 5-7 - the 32 lanes of a ds_write_b32 group (pass-2 lane bits 0-4 = index bits 0,1,5,6,7) then hit 32 different
// banks, and the 16-byte runs the store side reads stay whole (bits 0-1 untouched).
SDR_HD constexpr int row_word(int k) { return k ^ (((k >> 5) & 7) << 2); }

// ---------------------------------------------------------------------------------------------
// Twiddle tables (host side builds them from go-dsp's factor table W[k] = e^{-2 pi i k / N}, twiddles.h).
//   stage q of pass P pairs index bit 5P + q: h = 2^(5P+q); butterfly (i, i + h) uses W[(N / 2h) * (i mod h)],
//   i mod h = mm * 2^(5P) + lo with mm = the slot bits below the stage's and lo = the index bits below the pass's.
//   row(q, mm) = 2^q - 1 + mm.
//   block 0 (pass 0): 31 entries, one per row (lo = 0): wave-uniform, scalar loads
//   block 1 (pass 1): 31 rows x 32 entries, entry lo = index bits 0-4 (copied to LDS)
//   block 2 (pass 2): 15 rows x 1024 entries at position pos2(lo): the pass-2 thread id's bits that are index bits 0-8,
//                     lanes first, then bit 9 (the passenger slot bit): a wave instruction reads 64 consecutive entries.
// ---------------------------------------------------------------------------------------------
inline constexpr int kTw0 = 0;
inline constexpr int kTw1 = 32;
inline constexpr int kTw1Entries = 31 * 32;
inline constexpr int kTw2 = kTw1 + kTw1Entries;  // 1024
inline constexpr int kTwTotal = kTw2 + 15 * 1024;

// position inside a pass-2 row of the entry for lo (index bits 0-9)
SDR_HD constexpr int pos2_of_lo(int lo)
{
    const Layout L = layout(2);
    int r = 0;
    for (int j = 0; j < 9; j++) {
        const int b = L.tbit[j];
        if (b < 10)
            r |= ((lo >> b) & 1) << j;
    }
    r |= ((lo >> 9) & 1) << 9;
    return r;
}

inline void build_twiddles(const double *wre, const double *wim, cplx *out)
{
    for (int i = 0; i < kTwTotal; i++)
        out[i] = cplx{0.0, 0.0};
    for (int P = 0; P < NPASS; P++) {
        const int S = 1 << pass_base(P);
        for (int q = 0; q < pass_stages(P); q++)
            for (int mm = 0; mm < (1 << q); mm++)
                for (int lo = 0; lo < S; lo++) {
                    const int h = S << q;
                    const int k = (N / (2 * h)) * (mm * S + lo);
                    const int row = (1 << q) - 1 + mm;
                    const int at = P == 0 ? kTw0 + row : P == 1 ? kTw1 + row * 32 + lo : kTw2 + row * 1024 + pos2_of_lo(lo);
                    out[at] = cplx{wre[k], wim[k]};
                }
    }
}

// ---------------------------------------------------------------------------------------------
// One register pass: NST radix-2 stages on slot bits 0..NST-1, for every value u of the slot bits above them.
// tw(row, u) returns the twiddle of row = 2^q - 1 + mm for passenger group u.  P0: pass 0, whose twiddles W[0] = 1 and
// W[N/4] = -i are go-dsp's literal table entries: multiplying by them is exact up to the sign of a zero (which cannot
// reach |X|^2), so those multiplies are skipped, as in fft_f64.h.
// CHUNK: twiddles are fetched CHUNK at a time, one chunk ahead of the butterflies that use them (a scheduling fence
// behind each chunk): left alone the compiler has a stage's sixteen rows in flight at once, 64 registers the kernel does
// not have.
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define SDR_R32_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SDR_R32_FENCE() \
    do {                \
    } while (0)
#endif

struct NoStageHook {
    SDR_HD void operator()(int) const {}
};

// the twiddles of chunk C of stage Q (CHUNK > 0: C-th group of CHUNK (row, passenger group) pairs; 0: the whole stage)
template <int NST, bool P0, int CHUNK, int Q, int C, int CHMAX, class TW>
SDR_HD inline void load_chunk(TW &tw, double (&wr)[CHMAX], double (&wi)[CHMAX])
{
    constexpr int ROWS = 1 << Q;          // twiddles of this stage (per passenger group)
    constexpr int GROUPS = R >> NST;      // passenger groups
    constexpr int TOTAL = ROWS * GROUPS;  // (row, group) pairs
    constexpr int CH = (CHUNK > 0 && CHUNK < TOTAL) ? CHUNK : TOTAL;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < CH; j++) {
        const int idx = C * CH + j;
        const int mm = idx % ROWS, u = idx / ROWS;
        const bool one = P0 && mm == 0;
        const bool minus_i = P0 && Q >= 1 && mm == (ROWS >> 1);
        wr[j] = 1.0;
        wi[j] = 0.0;
        if (!one && !minus_i) {
            const cplx w = tw(ROWS - 1 + mm, u);
            wr[j] = w.x;
            wi[j] = w.y;
        }
    }
}

struct NoButterflyHook {
    SDR_HD void operator()(int, int) const {}
};

// bf(a, b) is called behind each butterfly (a, b) of the pass's LAST stage: slots a and b are final for the pass from
// then on (the kernel writes them into the exchange that follows while the stage's other butterflies compute)
template <int NST, bool P0, int CHUNK, int Q, int C, int CHMAX, class BF>
SDR_HD inline void compute_chunk(double *xr, double *xi, const double (&wr)[CHMAX], const double (&wi)[CHMAX], BF &bf)
{
    constexpr int ROWS = 1 << Q;
    constexpr int GROUPS = R >> NST;
    constexpr int TOTAL = ROWS * GROUPS;
    constexpr int CH = (CHUNK > 0 && CHUNK < TOTAL) ? CHUNK : TOTAL;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < CH; j++) {
        const int idx = C * CH + j;
        const int mm = idx % ROWS, u = idx / ROWS;
        const bool one = P0 && mm == 0;
        const bool minus_i = P0 && Q >= 1 && mm == (ROWS >> 1);
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int k = 0; k < ((1 << NST) >> (Q + 1)); k++) {
            const int a = (u << NST) + mm + (k << (Q + 1));
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
            if constexpr (Q == NST - 1)
                bf(a, b);
        }
    }
}

// chunk (Q, C) with its twiddles in (wr, wi): request the NEXT chunk's twiddles, compute this one, fence, go on.  The
// requests of a chunk are thus one chunk's arithmetic ahead of their use, in program order, whatever the scheduler does.
template <int NST, bool P0, int CHUNK, int Q, int C, int CHMAX, class TW, class HOOK, class BF>
SDR_HD inline void pass_chunks_from(double *xr, double *xi, TW &tw, HOOK &hook, BF &bf, const double (&wr)[CHMAX],
                                    const double (&wi)[CHMAX])
{
    constexpr int ROWS = 1 << Q;
    constexpr int GROUPS = R >> NST;
    constexpr int TOTAL = ROWS * GROUPS;
    constexpr int CH = (CHUNK > 0 && CHUNK < TOTAL) ? CHUNK : TOTAL;
    constexpr bool LAST = (C + 1 == TOTAL / CH);
    constexpr int NQ = LAST ? Q + 1 : Q, NC = LAST ? 0 : C + 1;
    double nwr[CHMAX], nwi[CHMAX];
    if constexpr (NQ < NST)
        load_chunk<NST, P0, CHUNK, NQ, NC, CHMAX>(tw, nwr, nwi);
    compute_chunk<NST, P0, CHUNK, Q, C, CHMAX>(xr, xi, wr, wi, bf);
    if constexpr (LAST)
        hook(Q);
    if (CHUNK > 0)
        SDR_R32_FENCE();
    if constexpr (NQ < NST)
        pass_chunks_from<NST, P0, CHUNK, NQ, NC, CHMAX>(xr, xi, tw, hook, bf, nwr, nwi);
}

// The same two chunks deep (pass 2: a chunk's arithmetic - four butterflies, 160 clocks - is a quarter of the L2 round
// trip its successor's twiddles need): chunk (Q, C) in (wr, wi), its successor in (nwr, nwi), the one after that is
// requested here.
struct ChunkId {
    int q, c;
};
template <int NST, int CHUNK>
SDR_HD constexpr ChunkId next_chunk(ChunkId k)
{
    const int total = (1 << k.q) * (R >> NST);
    const int ch = (CHUNK > 0 && CHUNK < total) ? CHUNK : total;
    return (k.c + 1 == total / ch) ? ChunkId{k.q + 1, 0} : ChunkId{k.q, k.c + 1};
}
template <int NST, bool P0, int CHUNK, int Q, int C, int CHMAX, class TW, class HOOK, class BF>
SDR_HD inline void pass_chunks2_from(double *xr, double *xi, TW &tw, HOOK &hook, BF &bf, const double (&wr)[CHMAX],
                                     const double (&wi)[CHMAX], const double (&nwr)[CHMAX], const double (&nwi)[CHMAX])
{
    constexpr ChunkId N1 = next_chunk<NST, CHUNK>(ChunkId{Q, C});
    constexpr ChunkId N2 = next_chunk<NST, CHUNK>(N1);
    double nnwr[CHMAX], nnwi[CHMAX];
    if constexpr (N1.q < NST && N2.q < NST)
        load_chunk<NST, P0, CHUNK, N2.q, N2.c, CHMAX>(tw, nnwr, nnwi);
    compute_chunk<NST, P0, CHUNK, Q, C, CHMAX>(xr, xi, wr, wi, bf);
    if constexpr (N1.q != Q)
        hook(Q);
    if (CHUNK > 0)
        SDR_R32_FENCE();
    if constexpr (N1.q < NST)
        pass_chunks2_from<NST, P0, CHUNK, N1.q, N1.c, CHMAX>(xr, xi, tw, hook, bf, nwr, nwi, nnwr, nnwi);
}

// hook(q) is called behind the last butterfly of stage q (the kernel hangs its input prefetch there)
template <int NST, int CHUNK>
inline constexpr int kChunkMax = CHUNK > 0 ? CHUNK : (R >> NST) << (NST - 1);

// The pass's first chunk of twiddles can be requested ahead of the pass (pass 2's come from L2: behind the exchange in
// front of it the wave would sit out a full round trip): first_chunk() in front of the exchange, run_pass_with() behind.
template <int NST, int CHUNK>
struct FirstChunk {
    double wr[kChunkMax<NST, CHUNK>], wi[kChunkMax<NST, CHUNK>];
};
template <int NST, bool P0, int CHUNK, class TW>
SDR_HD inline void first_chunk(TW tw, FirstChunk<NST, CHUNK> &f)
{
    load_chunk<NST, P0, CHUNK, 0, 0, kChunkMax<NST, CHUNK>>(tw, f.wr, f.wi);
}
template <int NST, bool P0, int CHUNK, class TW, class HOOK = NoStageHook, class BF = NoButterflyHook>
SDR_HD inline void run_pass_with(double *xr, double *xi, const FirstChunk<NST, CHUNK> &f, TW tw, HOOK hook = HOOK{}, BF bf = BF{})
{
    pass_chunks_from<NST, P0, CHUNK, 0, 0, kChunkMax<NST, CHUNK>>(xr, xi, tw, hook, bf, f.wr, f.wi);
}
// two chunks deep
template <int NST, int CHUNK>
struct FirstChunks2 {
    double wr[kChunkMax<NST, CHUNK>], wi[kChunkMax<NST, CHUNK>], nwr[kChunkMax<NST, CHUNK>], nwi[kChunkMax<NST, CHUNK>];
};
template <int NST, bool P0, int CHUNK, class TW>
SDR_HD inline void first_chunks2(TW tw, FirstChunks2<NST, CHUNK> &f)
{
    constexpr ChunkId N1 = next_chunk<NST, CHUNK>(ChunkId{0, 0});
    static_assert(N1.q < NST, "a pass of one chunk has no second one");
    load_chunk<NST, P0, CHUNK, 0, 0, kChunkMax<NST, CHUNK>>(tw, f.wr, f.wi);
    load_chunk<NST, P0, CHUNK, N1.q, N1.c, kChunkMax<NST, CHUNK>>(tw, f.nwr, f.nwi);
}
template <int NST, bool P0, int CHUNK, class TW, class HOOK = NoStageHook, class BF = NoButterflyHook>
SDR_HD inline void run_pass2_with(double *xr, double *xi, const FirstChunks2<NST, CHUNK> &f, TW tw, HOOK hook = HOOK{}, BF bf = BF{})
{
    pass_chunks2_from<NST, P0, CHUNK, 0, 0, kChunkMax<NST, CHUNK>>(xr, xi, tw, hook, bf, f.wr, f.wi, f.nwr, f.nwi);
}
template <int NST, bool P0, int CHUNK = 0, class TW, class HOOK = NoStageHook, class BF = NoButterflyHook>
SDR_HD inline void run_pass(double *xr, double *xi, TW tw, HOOK hook = HOOK{}, BF bf = BF{})
{
    FirstChunk<NST, CHUNK> f;
    first_chunk<NST, P0, CHUNK>(tw, f);
    run_pass_with<NST, P0, CHUNK>(xr, xi, f, tw, hook, bf);
}

// element (of the pass-1 twiddle block) a pass-1 thread reads for row r: its index bits 0-4
SDR_HD inline int tw1_lo(int t) { return thread_part<1>(t) & 31; }
// position (inside a pass-2 row) a pass-2 thread reads for passenger value u
SDR_HD inline int tw2_pos(int t, int u) { return pos2_of_lo((thread_part<2>(t) & 511) | (u << 9)); }

// natural-order DFT bin of slot s of thread t after pass 2
SDR_HD inline int output_bin(int t, int s) { return thread_part<2>(t) | slot_part(2, s); }

}  // namespace fft32
