// k_noise.hip — dsp.FindNoiseFloor (dsp/fft.go:215-252) and the rolling thresholds of Receiver.run
// (rx/receiver.go:381-385).  Compiled with -ffp-contract=off.
//
// FindNoiseFloor is two SEQUENTIAL float64 accumulations per frame (window sums, then the variance
// about the winning window's mean): float64 addition is not associative, so to reproduce the
// reference's bits each chain keeps its order — one lane per chain.  What is parallel is everything
// around the chain: a chain group owns 64 chains (64 consecutive frames); its producer waves take turns fetching
// each chain's next 64 values (one fully coalesced 256-byte load per chain) and lay them down transposed in an LDS
// ring guarded by flags - widened, mean subtracted and squared in float64 for the variance pass, as they are
// (float32) for the window sums; the consumer wave's lane i then only reads row i and adds - the strictly serial
// part is one v_add_f64 per term.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/sdrainer_hip.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

constexpr int TILE = 64;
constexpr int HALF = TILE / 2;
constexpr int MAX_WAVES = 16;  // a workgroup is 1024 threads: one to four chain groups, each one consumer wave + producers
// Window sums: FOUR chain groups of four waves (consumer + 3 producers) share a workgroup; their ring tiles hold the
// psd values as they are (float32), the consumer widens them on its way to the addition (a conversion issues in the
// shadow of the dependent add before it), so the producers only move data and three of them keep a consumer fed.
// History: one group of sixteen waves compiled to 102 VGPRs, i.e. one workgroup per CU whatever its LDS says, and the
// 320 groups of config 3 took two rounds over 256 CUs (0.045 ms); two groups of eight with float64 tiles: 160
// workgroups, 0.028 ms; half-size workgroups (two to a CU) are as fast standalone but spread over every CU and each
// keeps an FFT workgroup - which needs a whole CU - off it (0.2495 ms per pipelined step against 0.2465).
#ifndef SDR_WM_GROUPS
#define SDR_WM_GROUPS 4
#endif
constexpr int WM_GROUPS = SDR_WM_GROUPS;
// ... and HALF-SIZE workgroups (two groups, eight waves, 70 KB of LDS) where the FFT kernel's workgroups are not whole
// CUs (block sizes up to 8192: 512 threads, two to a CU): a 1024-thread workgroup then waits for a CU that BOTH of its
// FFT workgroups have left, a half-size one slots in beside one of them.  Config 5's share (8 bands x 8192), one box,
// interleaved twice: 191.8 / 192.7 GS/s against 186.2 / 187.0 (graph-captured 190.3 / 188.8 against 184.0 / 184.1);
// config 3, whose FFT workgroups ARE whole CUs, 167.2 / 167.9 against 167.2 / 168.2 - there the note above still holds.
// (The variance chains the same way - one vector-ALU group of eight waves per 512-thread workgroup - were measured too and
// are worse everywhere: config 5's share 181.5 against 186.3 with the matrix-pipe kernel, config 3 160.9 against 162.8.)
constexpr int WM_GROUPS_HALF = 2;
// Variance chains (SDR_VAR_MFMA=0, the default from round 4 on): TWO chain groups of eight waves per workgroup, vector-ALU
// consumers that square for themselves (chain_consumer<VAR>).  Round 3 ran them on the float64 matrix pipe, one group of
// 64 chains per workgroup over four consumer waves: 5.4 clocks per term for ONE wave alone on a CU (tools/ubench_mfma_f64)
// - but traced inside the kernel (tools/noise_trace.py) each consumer's dependent matrix instruction took 64 clocks, not
// 21.5: four consumers on four SIMDs do not get four pipes' worth, and 16 clocks per term is what a single vector-ALU
// consumer delivers as well (cvt, sub, mul, add: issue-bound).  Two such consumers on two SIMDs double the rate per CU,
// at which point the producers (LDS ring capacity over memory latency) are the limit.
// Both are built; the launch picks (launch_noise_stats): the two-group kernel where the CU time it holds is what counts
// (long batches: config 3 at 8192 frames 161.4 against 156.7 GS/s), the matrix-pipe kernel - half the frames per
// workgroup, twice the workgroups, 0.115 against 0.18 ms - where the noise stream's LATENCY bounds the step (config 5's
// share 183 against 176 GS/s, config 3 at 2048 frames 143 against 135).  SDR_VAR_MFMA=0 / 1 forces one (development).
constexpr int NS_GROUPS_VALU = 2;
constexpr int CHAIN_THREADS = 64 * MAX_WAVES;
// SLOTS = LDS tiles between producers and consumer: 4 float64 tiles (133 KB) for the long variance chains, 2 float32
// tiles per group (4 x 35 KB) for the short window sums.
template <int SLOTS, class T, int PAD_ = (sizeof(T) == 8 ? 2 : 4), int TAG_ = 0>
struct ChainShared {
    using term_t = T;
    static constexpr int RING_SLOTS = SLOTS;
    static constexpr int TAG = TAG_;  // (tells rings of the same shape apart: each kernel's LDS holds only its own)
    // [slot][chain][column]; rows are 528 bytes apart (float64) / 272 bytes (float32): 16-byte aligned for the
    // consumer's ds_read_b128, and the sixteen lanes one LDS cycle serves start on sixteen different 16-byte bank
    // groups, so neither side has bank conflicts.  (The variance ring's rows are 264 bytes apart: its consumers read
    // single words, sixteen rows x two columns per 32-lane group - word 66 c + k: all 32 banks.)
    static constexpr int PAD = PAD_;
    alignas(16) T term[SLOTS][TILE][TILE + PAD];
    double mean[TILE];                        // per chain: value subtracted before squaring (variance pass)
    int n_terms[TILE];                        // per chain: number of leading terms that count
    int ready[SLOTS][2];                      // ready[t % SLOTS][h] == t + 1  <=>  rows 32h..32h+31 of tile t are published
    int consumed;                             // tiles the consumer has finished with
    int simd_of_wave[MAX_WAVES];              // which SIMD each of the group's waves landed on (see chain_run)
    int consumed_by[4];                       // variance chains on the matrix pipe: tiles each of the four consumer waves is done with
    double chain_sum[TILE];                   // ... and where they leave their chains' sums
};
using WindowRing = ChainShared<2, float>;
using WindowRingHalf = ChainShared<2, float, 4, 1>;  // the half-size window-sum workgroups' (two of them instead of four)
using VarianceRing = ChainShared<6, float, 2>;  // (matrix-pipe variant)
using NoiseRing = ChainShared<4, float>;        // variance chains on the vector ALU: two groups x four float32 tiles

#if defined(SDR_NOISE_TRACE)
// diagnostic builds (tools/abl): where does the consumer of workgroup 0 spend its time?
__device__ unsigned long long g_noise_trace[32];  // variance consumer 0 of workgroup 7: [0] total, [1] waiting for tiles, [2] tiles, [3] spins, [4] total in shader clocks; [8..12] the same for the window sums' consumer of group 0
extern "C" __attribute__((visibility("default"))) int sdr_debug_noise_trace(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_noise_trace), sizeof(g_noise_trace));
}
#endif

// The rings live at file scope so that the consumer can be a function of its own (see chain_consumer).
__shared__ WindowRing g_ring_wm[WM_GROUPS];
__shared__ WindowRingHalf g_ring_wm_half[WM_GROUPS_HALF];
__shared__ VarianceRing g_ring_ns[1];
__shared__ NoiseRing g_ring_var[NS_GROUPS_VALU];
template <class Shared>
__device__ __forceinline__ Shared &ring(int group)
{
    if constexpr (Shared::RING_SLOTS == WindowRing::RING_SLOTS && Shared::TAG == 1)
        return g_ring_wm_half[group];
    else if constexpr (Shared::RING_SLOTS == WindowRing::RING_SLOTS)
        return g_ring_wm[group];
    else if constexpr (Shared::RING_SLOTS == VarianceRing::RING_SLOTS)
        return g_ring_ns[group];
    else
        return g_ring_var[group];
}

// Ordering between a wave's LDS accesses and its flag accesses.  A wave's DS instructions execute in issue
// order and LDS is one memory for the whole workgroup, so "data before flag" (producer) and "flag before
// data" (consumer) only need the COMPILER to keep the program order.  A real workgroup-scope fence compiles
// to s_waitcnt vmcnt(0) lgkmcnt(0): it would drain the producer's global loads of the unit after next and
// the consumer's reads of the next chunks, which is exactly the latency both sides run ahead to hide.
//
// HARDWARE DEPENDENCY (gfx950 / GCN-CDNA DS pipeline): this protocol is only correct because (a) one wave's
// LDS (DS) instructions are issued to and executed by the CU's single LDS unit in program order, and (b) all
// waves of a workgroup share that one LDS, so a flag write that executes after the data writes is also
// observed after them by every other wave.  The C++ memory model does not promise either (relaxed atomics +
// a compiler barrier order nothing formally); a target whose DS operations can complete out of order, or
// whose workgroup can span more than one LDS, needs release/acquire fences here instead.  The chains are
// checked bit for bit against the oracle on every GPU test run (tests/test_gpu_parity.py), which is what
// would catch a violation.
//
// -DSDR_SAFE_FENCES builds the protocol the memory model does promise - a workgroup-scope fence wherever program order
// is relied on, acquire loads and release stores on the flags - at the price described above.  That variant is built by
// __graft_entry__.build() beside the product (libsdrainer_hip_safe_fences.so) and the GPU parity tests are run against
// it too (tests/test_safe_fences.py): the fast protocol's dependence on the hardware is bounded by a build that does not
// have it and gives the same bits.
#if defined(SDR_SAFE_FENCES)
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ __forceinline__ int lds_flag_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_flag_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ int lds_flag_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_flag_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif

// lane `src` (a compile-time constant after unrolling) -> wave-uniform SGPR value: v_readlane_b32, not the
// LDS-crossbar ds_bpermute a generic __shfl lowers to
__device__ __forceinline__ int readlane_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double readlane_f64(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// A work unit is half a tile: 32 chains x 64 columns.  Producer wave p owns units u = p, p + 15, ...
// (unit u = rows 32*(u&1).. of tile u>>1): it fetches the 32 chains' next 64 values (row r = chain r, one
// coalesced 256-byte read per row, 32 loads in flight), widens / subtracts / squares them in float64
// and lays them down transposed in ring slot t % 4, then raises the half's flag.  Memory latency is
// hidden by the other fourteen producers, not by software pipelining inside one wave.
// Terms past a chain's own end are stored as +0.0: adding +0.0 to a non-negative float64 sum leaves it
// bit-identical, so the consumer needs no per-lane predicate.
template <bool VARIANCE, class Shared>
__device__ __forceinline__ void chain_producer(Shared &sh, const float *__restrict__ base, unsigned row_stride,
                                               int rows, int n_cols, int n_tiles, int min_terms, int p, int np,
                                               int lane)
{
    // lane r keeps chain r's mean / length; rows read them with v_readlane (wave-uniform, no LDS traffic)
    const double mean_of_lane = sh.mean[lane];
    const int terms_of_lane = sh.n_terms[lane];
    const int n_units = 2 * n_tiles;
    // unconditional loads (a predicated load compiles to branch + load + vmcnt(0): serial round
    // trips): out-of-range rows / columns / units are clamped to a valid address and their values are
    // discarded below by the chain-length test (such chains have n_terms <= col) or never used
    auto fetch = [&](float (&v)[HALF], int u) {
        u = min(u, n_units - 1);
        const int t = u >> 1, h = u & 1;
        const unsigned ccol = min((unsigned)(t * TILE + lane), (unsigned)(n_cols - 1));
#pragma unroll
        for (int i = 0; i < HALF; i++)
            v[i] = base[(unsigned)min(h * HALF + i, rows - 1) * row_stride + ccol];
    };
    auto publish = [&](const float (&v)[HALF], int u) {
        const int t = u >> 1, h = u & 1;
        const unsigned col = (unsigned)(t * TILE + lane);
        constexpr int RING_SLOTS = Shared::RING_SLOTS;
        const int slot = t % RING_SLOTS;
        if (t >= RING_SLOTS)
            while (lds_flag_load(&sh.consumed) < t - RING_SLOTS + 1)
                __builtin_amdgcn_s_sleep(1);
        lds_order();
        const bool full = (t + 1) * TILE <= min_terms;  // every chain still covers the whole tile
        if constexpr (sizeof(typename Shared::term_t) == 4) {
            // float32 ring: the values as they are - and in a full tile nothing but the stores (the producers' vector-ALU
            // work lands on SIMDs that other groups' consumers chain on: with a conversion, a lane read, a compare and a
            // select per element the two-group variance consumers ran at 32 clocks per term instead of their own 16)
            if (full) {
#pragma unroll
                for (int i = 0; i < HALF; i++)
                    sh.term[slot][h * HALF + i][lane] = v[i];
                lds_order();
                if (lane == 0)
                    lds_flag_store(&sh.ready[slot][h], t + 1);
                return;
            }
        }
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < HALF; i++) {
                double x = (double)v[i];
                if (VARIANCE && sizeof(typename Shared::term_t) == 8) {  // (float64 tiles only: a float32 ring takes the values as they are)
                    const double d = x - readlane_f64(mean_of_lane, i);
                    x = d * d;  // math.Pow(d, 2)
                }
                if (!full && (int)col >= readlane_i32(terms_of_lane, i))
                    x = 0.0;
                sh.term[slot][i][lane] = (typename Shared::term_t)x;
            }
        } else {
#pragma unroll
            for (int i = 0; i < HALF; i++) {
                double x = (double)v[i];
                if (VARIANCE && sizeof(typename Shared::term_t) == 8) {
                    const double d = x - readlane_f64(mean_of_lane, HALF + i);
                    x = d * d;
                }
                if (!full && (int)col >= readlane_i32(terms_of_lane, HALF + i))
                    x = 0.0;
                sh.term[slot][HALF + i][lane] = (typename Shared::term_t)x;
            }
        }
        lds_order();
        if (lane == 0)
            lds_flag_store(&sh.ready[slot][h], t + 1);
    };
    // (Two units in flight per producer - the next unit's loads issued before this one is published - were
    // measured after the fences were gone too: window means 0.058 instead of 0.047 ms, variance chain equal.)
    if constexpr (Shared::RING_SLOTS >= 4) {
        // a ring of four tiles lets the producers run ahead: the next unit's reads are in flight while this one is written
        // (what is on its way per CU is producers x units x 8 KB, and that over the memory latency is the kernel's rate:
        // one unit each gave the two-group variance chains 33 GB/s per CU, tools/noise_trace.py)
        float v[HALF], nv[HALF];
        if (p < n_units)
            fetch(nv, p);
        for (int u = p; u < n_units; u += np) {
#pragma unroll
            for (int i = 0; i < HALF; i++)
                v[i] = nv[i];
            if (u + np < n_units)
                fetch(nv, u + np);
            publish(v, u);
        }
    } else {
        float v[HALF];
        for (int u = p; u < n_units; u += np) {
            fetch(v, u);
            publish(v, u);
        }
    }
}

// Consumer: lane = chain; per tile 64 strictly ordered float64 additions.  Only the v_add_f64 chain is
// serial (6.5 clk per term); the ds_read_b128 that feed it run 32 terms ahead, across tile boundaries too
// (the next tile's flags are requested two chunks before they are needed), so no LDS latency sits
// between two additions unless the producers are late.
// Not inlined on purpose: inside the kernel the register allocator has the producers' half-tile of loads
// and conversions in the same function and ends up spilling the chain's operands; as a function of its
// own the consumer gets a clean allocation (about 70 VGPRs, nothing spilled).
// VAR: the chain is a variance chain - term = (x - mean)^2 (math.Pow(d, 2), dsp/fft.go:247), widened, subtracted and squared
// here from the ring's float32 values (three float64 operations per term beside the serial addition: the consumer becomes
// issue-bound at 16 clocks per term, which two consumers on two SIMDs of the CU still deliver at twice the rate the matrix
// pipe did for four); terms past the lane's chain end count as +0 (only in the tiles where some chain of the group ends:
// a scalar branch).
template <class Shared, bool VAR = false>
__device__ __attribute__((noinline)) double chain_consumer(int n_tiles_any_lane, int lane, int group_any_lane, double mean = 0.0, int n_terms = 0,
                                                           int min_terms_any_lane = 0)
{
    using T = typename Shared::term_t;
    const int min_terms = __builtin_amdgcn_readfirstlane(min_terms_any_lane);
    const int n_tiles = __builtin_amdgcn_readfirstlane(n_tiles_any_lane);  // wave-uniform: scalar loop control
    Shared &sh = ring<Shared>(__builtin_amdgcn_readfirstlane(group_any_lane));
    constexpr int RING_SLOTS = Shared::RING_SLOTS;
    constexpr int CH = 8;
    double sum = 0;
    if (n_tiles <= 0)
        return sum;
    __builtin_amdgcn_s_setprio(3);
#if defined(SDR_NOISE_TRACE)
    unsigned long long tr_wait = 0, tr_spins = 0;
    const unsigned long long tr_start = wall_clock64();
    const unsigned long long tr_clk0 = clock64();
#endif
    auto wait_tile = [&](int t, int f0, int f1) {
        const int slot = t % RING_SLOTS;
#if defined(SDR_NOISE_TRACE)
        const unsigned long long w0 = wall_clock64();
#endif
        while (f0 != t + 1 || f1 != t + 1) {
            __builtin_amdgcn_s_sleep(1);
            f0 = lds_flag_load(&sh.ready[slot][0]);
            f1 = lds_flag_load(&sh.ready[slot][1]);
#if defined(SDR_NOISE_TRACE)
            tr_spins++;
#endif
        }
#if defined(SDR_NOISE_TRACE)
        tr_wait += wall_clock64() - w0;
#endif
        lds_order();
    };
    // (the buffers hold the ring's own type: float32 terms are widened when they are added, not when they are read -
    // a conversion right behind the read would wait for it and undo the read-ahead)
    typedef T vec16 __attribute__((ext_vector_type(16 / sizeof(T))));
    constexpr int PER_READ = 16 / (int)sizeof(T);
    auto load = [&](T (&b)[CH], int t, int c) {
        const vec16 *row = reinterpret_cast<const vec16 *>(&sh.term[t % RING_SLOTS][lane][c * CH]);
#pragma unroll
        for (int j = 0; j < CH / PER_READ; j++) {
            const vec16 v = row[j];  // ds_read_b128
#pragma unroll
            for (int k = 0; k < PER_READ; k++)
                b[PER_READ * j + k] = v[k];
        }
    };
    // VAR: the squares of the chunk that is added next (carried from step to step)
    double pc[CH] = {};
    auto squares = [&](double (&p)[CH], const T (&b)[CH], int t, int c) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const double d = (double)b[j] - mean;
            p[j] = d * d;
        }
        if ((t + 1) * TILE > min_terms) {  // (wave-uniform: some chain of the group ends in this tile or before it)
#pragma unroll
            for (int j = 0; j < CH; j++)
                p[j] = (t * TILE + c * CH + j < n_terms) ? p[j] : 0.0;  // (+0.0 leaves a non-negative sum bit-identical)
        }
    };
    // one step = the eight ordered additions of chunk (t, c).  VAR: its squares were formed a step ago; the NEXT chunk's
    // (`bn`, chunk (tn, cn), read three steps ago) are formed here, in the same straight-line block, so that their 24
    // independent operations fill the issue slots each dependent addition leaves open (squares first, then additions:
    // 32 clocks per term measured; term by term: 35; the pipe's own rate for four operations per term is 16).
    auto add = [&](const T (&b)[CH], int t, int c, const T (&bn)[CH], int tn, int cn, bool have_next) {
        if constexpr (!VAR) {
#pragma unroll
            for (int j = 0; j < CH; j++)
                sum += (double)b[j];
        } else {
            double pn[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) {
                sum += pc[j];
                const double d = (double)bn[j] - mean;
                pn[j] = d * d;
            }
            if (have_next && (tn + 1) * TILE > min_terms) {
#pragma unroll
                for (int j = 0; j < CH; j++)
                    pn[j] = (tn * TILE + cn * CH + j < n_terms) ? pn[j] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < CH; j++)
                pc[j] = pn[j];
        }
    };
    // Four 8-term buffers, each chunk read four chunks (32 terms, > 200 clocks of additions) before it is
    // added: chunk c of a tile lives in buffer c % 4.  Entering tile t, its chunks 0-3 are already on
    // their way (requested during the second half of tile t-1).
    T q0[CH], q1[CH], q2[CH], q3[CH];
    wait_tile(0, lds_flag_load(&sh.ready[0][0]), lds_flag_load(&sh.ready[0][1]));
    load(q0, 0, 0);
    load(q1, 0, 1);
    load(q2, 0, 2);
    load(q3, 0, 3);
    if constexpr (VAR)
        squares(pc, q0, 0, 0);
    for (int t = 0; t < n_tiles; t++) {
        const bool more = t + 1 < n_tiles;
        const int nslot = (t + 1) % RING_SLOTS;
        // (each empty asm pins the running sum and memory: it keeps the reads where they are written -
        // ahead of the additions they overlap - and stops the scheduler from sinking additions below later
        // reads, which keeps a whole tile of operands alive and spills them)
#define SDR_PIN asm volatile("" : "+v"(sum)::"memory")
        add(q0, t, 0, q1, t, 1, true); SDR_PIN; load(q0, t, 4); SDR_PIN;
        add(q1, t, 1, q2, t, 2, true); SDR_PIN; load(q1, t, 5); SDR_PIN;
        int f0 = 0, f1 = 0;
        if (more) {
            f0 = lds_flag_load(&sh.ready[nslot][0]);
            f1 = lds_flag_load(&sh.ready[nslot][1]);
        }
        add(q2, t, 2, q3, t, 3, true); SDR_PIN; load(q2, t, 6); SDR_PIN;
        add(q3, t, 3, q0, t, 4, true); SDR_PIN; load(q3, t, 7); SDR_PIN;
        if (more)
            wait_tile(t + 1, f0, f1);
        add(q0, t, 4, q1, t, 5, true); SDR_PIN; if (more) load(q0, t + 1, 0); SDR_PIN;
        add(q1, t, 5, q2, t, 6, true); SDR_PIN; if (more) load(q1, t + 1, 1); SDR_PIN;
        add(q2, t, 6, q3, t, 7, true); SDR_PIN; if (more) load(q2, t + 1, 2); SDR_PIN;
        add(q3, t, 7, q0, t + 1, 0, more); SDR_PIN; if (more) load(q3, t + 1, 3); SDR_PIN;
#undef SDR_PIN
        lds_order();
        if (lane == 0)
            lds_flag_store(&sh.consumed, t + 1);
    }
    __builtin_amdgcn_s_setprio(0);
#if defined(SDR_NOISE_TRACE)
    if (blockIdx.x == 7 && blockIdx.y == 0 && lane == 0) {
        const int grp = __builtin_amdgcn_readfirstlane(group_any_lane);
        if (grp == 0) {
            g_noise_trace[8] = wall_clock64() - tr_start;
            g_noise_trace[9] = tr_wait;
            g_noise_trace[10] = (unsigned long long)n_tiles;
            g_noise_trace[11] = tr_spins;
            g_noise_trace[12] = clock64() - tr_clk0;
        }
        g_noise_trace[16 + 2 * grp] = wall_clock64() - tr_start;  // every group of the workgroup: chain time, waiting
        g_noise_trace[17 + 2 * grp] = tr_wait;
    }
#endif
    return sum;
}

// Runs 64 chains (lane i of wave 0 owns chain i).  `my_terms` / `my_mean` are the consumer lane's chain
// length and mean; returns the chain's sum in the consumer lanes.  All waits are on waves of the same
// workgroup (co-resident by construction), so every spin terminates.
template <bool VARIANCE, int GROUPS, class Shared, int WAVES = MAX_WAVES>
__device__ __forceinline__ double chain_run(Shared &sh, int group, int wig, const float *__restrict__ base,
                                            size_t row_stride, int rows, int n_cols, int my_terms, double my_mean)
{
    // `group` = which of the workgroup's chain groups this wave belongs to (sh is that group's ring), `wig` = the
    // wave's index in the group, wave-uniform both; wave 0 of a group is its consumer.
    constexpr int N_PRODUCERS = WAVES / GROUPS - 1;
    const int lane = threadIdx.x & 63;
    // HW_REG_HW_ID bits 5:4 = SIMD this wave runs on.  The consumer's additions are a pure latency chain;
    // a producer on the same SIMD puts its float64 instructions between them (measured: 13.5 instead of
    // 6.5 clocks per term).  Producers that share their consumer's SIMD therefore sit the chain out.
    const int my_simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
    if (lane == 0)
        sh.simd_of_wave[wig] = my_simd;
    if (wig == 0) {
        sh.n_terms[lane] = my_terms;
        sh.mean[lane] = my_mean;
        if (lane < Shared::RING_SLOTS) {
            sh.ready[lane][0] = 0;
            sh.ready[lane][1] = 0;
        }
        if (lane == 0)
            sh.consumed = 0;
    }
    __syncthreads();
    int max_terms = sh.n_terms[lane], min_terms = max_terms;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        max_terms = max(max_terms, __shfl_xor(max_terms, o));
        min_terms = min(min_terms, __shfl_xor(min_terms, o));
    }
    const int n_tiles = __builtin_amdgcn_readfirstlane((max_terms + TILE - 1) / TILE);
    min_terms = __builtin_amdgcn_readfirstlane(min_terms);
    // rank of this wave among the producers that are not on the consumer's SIMD, and their number
    int p = 0, np = 0;
    const int consumer_simd = sh.simd_of_wave[0];
    for (int w = 1; w <= N_PRODUCERS; w++) {
        const bool active = sh.simd_of_wave[w] != consumer_simd;
        p += (active && w < wig) ? 1 : 0;
        np += active ? 1 : 0;
    }
    p = __builtin_amdgcn_readfirstlane(p);
    np = __builtin_amdgcn_readfirstlane(np);
    const bool sits_out = np > 0 && my_simd == consumer_simd;
    if (np == 0) {  // (every producer on the consumer's SIMD: not with 8 or 16 waves dealt over 4 SIMDs)
        p = wig - 1;
        np = N_PRODUCERS;
    }
    double sum = 0;
    if (wig == 0)
        sum = chain_consumer<Shared, VARIANCE && sizeof(typename Shared::term_t) == 4>(n_tiles, lane, group, my_mean, my_terms, min_terms);
    else if (!sits_out)
        chain_producer<VARIANCE>(sh, base, (unsigned)row_stride, rows, n_cols, n_tiles, min_terms, p, np, lane);
    __syncthreads();  // everyone is done with the rings before a caller re-initialises them
    return sum;
}

// ---------------------------------------------------------------------------------------------
// The variance chains on the MATRIX pipe.  A dependent v_add_f64 costs 10 clocks, so a chain of up to 16384 ordered
// additions cannot take less than 68 us on the vector ALU whatever feeds it (round 2: 108 us).  v_mfma_f64_4x4x4_4b
// computes D = C + A x B as a strictly sequential chain of float64 fused multiply-adds in k order - measured, every bit
// (tools/ubench_mfma_f64.hip: 64 000 outputs on wide-range random data equal fma(a3,1, fma(a2,1, fma(a1,1, fma(a0,1,c))))
// and no other association) - and with B = 1 an fma IS the rounded addition the reference performs (dsp/fft.go:246-248:
// sum += term, term already rounded).  One instruction therefore adds FOUR consecutive terms to each of sixteen
// chains, in order, in 21.5 clocks: 5.4 per term.  Operand layout (probed): lane L supplies term k = L >> 4 of chain
// L & 15; the sum of chain c sits in lanes 16 (c & 3) + 4 (c >> 2) + j, j = 0..3.
// Four consumer waves (one per SIMD: the matrix pipe is per SIMD) take sixteen of the workgroup's 64 chains each and
// read what the twelve producers laid down (the psd values as they are, rows 264 bytes apart: conflict-free for this
// read pattern) and widen, subtract and square them in the shadow of the matrix instruction before; each reports the
// tiles it is done with on its own, the producers wait for the slowest.
// ---------------------------------------------------------------------------------------------
constexpr int NV_CONSUMERS = 4;
constexpr int NV_PRODUCERS = MAX_WAVES - NV_CONSUMERS;

__device__ __attribute__((noinline)) void variance_consumer_mfma(int n_tiles_any_lane, int lane, int consumer_any_lane, int min_terms_any_lane)
{
    const int n_tiles = __builtin_amdgcn_readfirstlane(n_tiles_any_lane);
    const int g = __builtin_amdgcn_readfirstlane(consumer_any_lane);
    const int min_terms = __builtin_amdgcn_readfirstlane(min_terms_any_lane);
    VarianceRing &sh = g_ring_ns[0];
    constexpr int RING_SLOTS = VarianceRing::RING_SLOTS;
    constexpr int Q = TILE / 4;  // matrix instructions per tile
    const int row = 16 * g + (lane & 15), k = lane >> 4;
    const double mean = sh.mean[row];     // this lane's chain: the value subtracted before squaring
    const int n_terms = sh.n_terms[row];  // ... and how many leading terms count
    double acc = 0.0;
    if (n_tiles <= 0)
        return;
    __builtin_amdgcn_s_setprio(3);
#if defined(SDR_NOISE_TRACE)
    unsigned long long tr_wait = 0, tr_spins = 0;
    const unsigned long long tr_start = wall_clock64();
    const unsigned long long tr_clk0 = clock64();
#endif
    auto ready = [&](int t) {
        const int slot = t % RING_SLOTS;
        return lds_flag_load(&sh.ready[slot][0]) == t + 1 && lds_flag_load(&sh.ready[slot][1]) == t + 1;
    };
    auto wait_tile = [&](int t) {
#if defined(SDR_NOISE_TRACE)
        const unsigned long long w0 = wall_clock64();
#endif
        while (!ready(t)) {
            __builtin_amdgcn_s_sleep(1);
#if defined(SDR_NOISE_TRACE)
            tr_spins++;
#endif
        }
#if defined(SDR_NOISE_TRACE)
        tr_wait += wall_clock64() - w0;
#endif
        lds_order();
    };
    // terms 4q .. 4q+3 of sixteen chains per ds_read_b32
    auto load = [&](float (&x)[Q], int t) {
        const int slot = t % RING_SLOTS;
#pragma unroll
        for (int q = 0; q < Q; q++)
            x[q] = sh.term[slot][row][4 * q + k];
    };
    // A tile's sixteen reads are issued a whole tile AHEAD of the matrix instructions that consume them: round 3 read a
    // tile and then waited out the LDS - shared with twelve producers' writes - before its first instruction, every tile
    // (16 clocks per term measured, tools/noise_trace.py, against the pipe's 5.4).  Now tile t+1 is requested - if its
    // flags are already up, the usual case: the producers run ahead - before tile t's chain starts, and its slot is handed
    // back to the producers as soon as those reads have been issued and waited for once, at the top of the next round.
    // One straight path per tile (no branch between the reads and the chain: at a join the compiler must assume the
    // youngest LDS reads are the ones it needs and waits for all of them - which is what it did to the first version of
    // this loop): tile t+1's flags are waited for, its reads issued, THEN tile t's chain runs; the two register sets
    // alternate (a copy would wait for the reads it copies).
    auto chain = [&](const float (&x)[Q], int t) {
        // term = (psd - mean)^2, math.Pow(d, 2) (dsp/fft.go:247), widened, subtracted and squared in float64 here, in
        // the shadow of the previous instruction on the matrix pipe; terms past the chain's end count as +0 (only in
        // the tiles where some chain of the workgroup ends: a scalar branch)
        if ((t + 1) * TILE <= min_terms) {
#if defined(SDR_VAR_PRE)
            double p[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const double d = (double)x[q] - mean;
                p[q] = d * d;
            }
#pragma unroll
            for (int q = 0; q < Q; q++)
                asm volatile("" : "+v"(p[q]));  // (all squares first, then nothing but the chain)
#pragma unroll
            for (int q = 0; q < Q; q++)
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(p[q], 1.0, acc, 0, 0, 0);
#else
            double p_next = ((double)x[0] - mean) * ((double)x[0] - mean);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const double p = p_next;
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(p, 1.0, acc, 0, 0, 0);
                if (q + 1 < Q) {
                    const double d = (double)x[q + 1] - mean;
                    p_next = d * d;
                }
            }
#endif
        } else {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const double d = (double)x[q] - mean;
                double p = d * d;
                if (t * TILE + 4 * q + k >= n_terms)
                    p = 0.0;
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(p, 1.0, acc, 0, 0, 0);
            }
        }
        asm volatile("" : "+v"(acc)::"memory");
        // tile t's values are in registers (and used): its slot may be refilled
        lds_order();
        if (lane == 0)
            lds_flag_store(&sh.consumed_by[g], t + 1);
    };
    // (the values a tile's reads deliver are first TOUCHED behind the chain of the tile before - an empty asm per
    // register, ordered after the chain's own: otherwise the scheduler hoists their conversions above that chain, and
    // with them the wait for the reads)
    auto landed = [](float (&x)[Q]) {
#pragma unroll
        for (int q = 0; q < Q; q++)
            asm volatile("" : "+v"(x[q]));
    };
    float xa[Q], xb[Q];
    wait_tile(0);
    load(xa, 0);
    landed(xa);
    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
        wait_tile(t + 1);
        load(xb, t + 1);
        chain(xa, t);
        landed(xb);
        if (t + 2 < n_tiles) {
            wait_tile(t + 2);
            load(xa, t + 2);
        }
        chain(xb, t + 1);
        if (t + 2 < n_tiles)
            landed(xa);
    }
    if (t < n_tiles)
        chain(xa, t);
    __builtin_amdgcn_s_setprio(0);
#if defined(SDR_NOISE_TRACE)
    if (blockIdx.x == 7 && g == 0 && lane == 0) {
        g_noise_trace[0] = wall_clock64() - tr_start;
        g_noise_trace[1] = tr_wait;
        g_noise_trace[2] = (unsigned long long)n_tiles;
        g_noise_trace[3] = tr_spins;
        g_noise_trace[4] = clock64() - tr_clk0;
    }
#endif
    // chain c of this consumer: lanes 16 (c & 3) + 4 (c >> 2) + j hold its sum; j = 0 writes it
    if ((lane & 3) == 0) {
        const int c = 4 * ((lane >> 2) & 3) + (lane >> 4);
        sh.chain_sum[16 * g + c] = acc;
    }
}

// producers of the matrix-pipe variance chains: they only move data - each unit is 32 chains' next 64 psd values (one
// coalesced 256-byte read per chain), laid down as they are (float32); the next unit's reads are in flight while this
// one is written.  They wait for the slowest of the four consumers before reusing a ring slot.
__device__ __forceinline__ void variance_producer(VarianceRing &sh, const float *__restrict__ base, unsigned row_stride, int rows,
                                                  int n_cols, int n_tiles, int p, int np, int lane)
{
    const int n_units = 2 * n_tiles;
    constexpr int RING_SLOTS = VarianceRing::RING_SLOTS;
    auto fetch = [&](float (&v)[HALF], int u) {
        u = min(u, n_units - 1);
        const int t = u >> 1, h = u & 1;
        const unsigned ccol = min((unsigned)(t * TILE + lane), (unsigned)(n_cols - 1));
#pragma unroll
        for (int i = 0; i < HALF; i++)
            v[i] = base[(unsigned)min(h * HALF + i, rows - 1) * row_stride + ccol];
    };
    float v[HALF], nv[HALF];
    if (p < n_units)
        fetch(nv, p);
    for (int u = p; u < n_units; u += np) {
#pragma unroll
        for (int i = 0; i < HALF; i++)
            v[i] = nv[i];
        if (u + np < n_units)
            fetch(nv, u + np);
        const int t = u >> 1, h = u & 1;
        const int slot = t % RING_SLOTS;
        if (t >= RING_SLOTS) {
            const int need = t - RING_SLOTS + 1;
            while (lds_flag_load(&sh.consumed_by[0]) < need || lds_flag_load(&sh.consumed_by[1]) < need ||
                   lds_flag_load(&sh.consumed_by[2]) < need || lds_flag_load(&sh.consumed_by[3]) < need)
                __builtin_amdgcn_s_sleep(1);
        }
        lds_order();
#pragma unroll
        for (int i = 0; i < HALF; i++)
            sh.term[slot][h * HALF + i][lane] = v[i];
        lds_order();
        if (lane == 0)
            lds_flag_store(&sh.ready[slot][h], t + 1);
    }
}

// Runs the workgroup's 64 variance chains; `my_terms` / `my_mean`: chain (frame) `lane`'s length and mean, valid in
// wave 0.  Returns chain `lane`'s sum in every wave (read back from LDS behind the closing barrier).
__device__ __forceinline__ double variance_run(int wave, const float *__restrict__ base, size_t row_stride, int rows, int n_cols,
                                               int my_terms, double my_mean)
{
    VarianceRing &sh = g_ring_ns[0];
    const int lane = threadIdx.x & 63;
    if (wave == 0) {
        sh.n_terms[lane] = my_terms;
        sh.mean[lane] = my_mean;
        sh.chain_sum[lane] = 0.0;
        if (lane < VarianceRing::RING_SLOTS) {
            sh.ready[lane][0] = 0;
            sh.ready[lane][1] = 0;
        }
        if (lane < NV_CONSUMERS)
            sh.consumed_by[lane] = 0;
    }
    __syncthreads();
    int max_terms = sh.n_terms[lane], min_terms = max_terms;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        max_terms = max(max_terms, __shfl_xor(max_terms, o));
        min_terms = min(min_terms, __shfl_xor(min_terms, o));
    }
    const int n_tiles = __builtin_amdgcn_readfirstlane((max_terms + TILE - 1) / TILE);
    min_terms = __builtin_amdgcn_readfirstlane(min_terms);
    if (wave < NV_CONSUMERS)
        variance_consumer_mfma(n_tiles, lane, wave, min_terms);
    else
        variance_producer(sh, base, (unsigned)row_stride, rows, n_cols, n_tiles, wave - NV_CONSUMERS, NV_PRODUCERS, lane);
    __syncthreads();
    return sh.chain_sum[lane];
}

// pass 1: mean of window w of frame f = sequential float64 sum of psd[edge + w*W .. +W) / W (:239-241,:230).
// A workgroup walks `windows_per_block` windows of its 64 frames back to back: with many bands there are
// thousands of (frame group, window) chains, and one long-lived workgroup per CU beats ten short ones.
template <int GROUPS, class Ring>
__global__ __launch_bounds__(64 * (MAX_WAVES / WM_GROUPS) * GROUPS) void k_window_means(const float *__restrict__ psd,
                                                                double *__restrict__ win_mean, NoiseGeom g,
                                                                int n_frames, int stride, int windows_per_block)
{
    constexpr int WPG = MAX_WAVES / WM_GROUPS, WAVES = WPG * GROUPS;  // four waves per group either way
    // (readfirstlane: the role split in chain_run becomes scalar branches instead of exec-masked regions)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int group = wave / WPG;
    // group k's consumer is its wave k, so that the consumers of a workgroup sit on different SIMDs (waves are dealt
    // to SIMDs round-robin; chain_run asks the hardware where each one landed anyway)
    const int wig = (wave - group * WPG - group + WPG) % WPG;
    Ring &sh = ring<Ring>(group);
    const int f0 = (blockIdx.x * GROUPS + group) * TILE, band = blockIdx.z;
    const int rows = max(0, min(TILE, n_frames - f0));  // (0: the odd group out at the end of a band sums nothing)
    const size_t frame0 = (size_t)band * stride + f0;
    const int w_end = min(g.n_windows, (int)(blockIdx.y + 1) * windows_per_block);
#if defined(SDR_NOISE_TRACE)
    const unsigned long long wg_t0 = wall_clock64();
#endif
    for (int w = blockIdx.y * windows_per_block; w < w_end; w++) {
        const float *base = psd + frame0 * g.n + g.edge + (size_t)w * g.window;
        const double sum = chain_run<false, GROUPS, Ring, WAVES>(sh, group, wig, base, g.n, rows, g.window, lane < rows ? g.window : 0, 0.0);
        if (wig == 0 && lane < rows)
            win_mean[(frame0 + lane) * 10 + w] = sum / (double)g.window;
    }
#if defined(SDR_NOISE_TRACE)
    if (blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0)
        g_noise_trace[24] = wall_clock64() - wg_t0;  // the workgroup's wave 0, first instruction to last
#endif
}

// pass 2: pick the minimum window in the reference's order, then the variance chain over
// psd[edge .. resultTo] (inclusive) about that mean, divided by windowSize (:244-249, App. C1);
// finally the two dB inputs of the rolling means (rx/receiver.go:383-384).
template <bool MFMA>
__global__ __launch_bounds__(CHAIN_THREADS) void k_noise_stats(const float *__restrict__ psd,
                                                               const double *__restrict__ win_mean,
                                                               sdr_frame_rec *__restrict__ recs, NoiseGeom g,
                                                               int n_frames, int stride)
{
    constexpr int NS_GROUPS = MFMA ? 1 : NS_GROUPS_VALU;
    constexpr int WPG = MAX_WAVES / NS_GROUPS;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int group = wave / WPG;
    const int wig = (wave - group * WPG - group + WPG) % WPG;  // (as in k_window_means)
    const bool consumer = wig == 0;
    const int f0 = (blockIdx.x * NS_GROUPS + group) * TILE, band = blockIdx.y;
    const int rows = max(0, min(TILE, n_frames - f0));
    const size_t frame0 = (size_t)band * stride + f0;
    const bool valid = consumer && lane < rows;
    const size_t frame = frame0 + (lane < rows ? lane : 0);

    double minValue = 0, resultMean = 0;
    int n_terms = 0;
    if (valid) {
        minValue = (double)psd[frame * g.n];  // :217, overridden by `first` as soon as a window is evaluated
        bool first = true;
        n_terms = 1;
        for (int w = 0; w < g.n_windows; w++) {
            const double mean = win_mean[frame * 10 + w];
            if (mean < minValue || first) {  // :232
                minValue = mean;
                first = false;
                resultMean = mean;
                // resultFrom = edge (`from` is only assigned on the first iteration, App. C1),
                // resultTo = edge + (w+1)*W  ->  (w+1)*W + 1 terms
                n_terms = (w + 1) * g.window + 1;
            }
        }
    }
    // sdr_create guarantees n_windows >= 9, so every chain starts at `edge`
    const float *base = psd + frame0 * g.n + g.edge;
    double sum;
    if constexpr (MFMA)
        sum = variance_run(wave, base, g.n, rows, g.n - g.edge, n_terms, resultMean);
    else
        sum = chain_run<true, NS_GROUPS>(ring<NoiseRing>(group), group, wig, base, g.n, rows, g.n - g.edge, n_terms, resultMean);
    if (!valid)
        return;
    const double variance = sum / (double)g.window;
    const float psdNoiseFloor = (float)minValue;
    sdr_frame_rec r;
    r.min_mean = psdNoiseFloor;
    r.variance = variance;
    // rx/receiver.go:383  T(float64(PSDValueIndB(T(Sqrt(var)), N) + dBmShift) * 0.25)
    r.dev_in = (float)((double)(gomath::psd_value_in_db((float)::sqrt(variance), g.inv_n2) + 120.0f) * 0.25);
    // rx/receiver.go:384  PSDValueIndB(psdNoiseFloor, N) + dBmShift
    r.nf_in = gomath::psd_value_in_db(psdNoiseFloor, g.inv_n2) + 120.0f;
    r.noise_dev = r.noise_floor = r.peak_thr = r.listen_thr = 0;
    r.pad = 0;
    recs[frame] = r;
}

// ---------------------------------------------------------------------------------------------
// k_thresholds — RollingMean.Put x2 per frame (dsp/dsp.go:257-268, rx/receiver.go:383-385) in frame
// order.  The value leaving the 60-frame window at frame f is the input of frame f-60 (or the ring
// carried over from the previous batch), so only the two float32 running sums form a serial chain - two
// dependent additions per frame: lanes 0 and 1 of wave 0 run them side by side, chunk after chunk without a
// pause, while the other waves fetch the next chunk's inputs and turn the previous chunk's sums into
// thresholds (one barrier per chunk; round 4: 0.127 -> see DESIGN.md, the chain used to wait for both).
// ---------------------------------------------------------------------------------------------
#ifndef SDR_THR_CHUNK
#define SDR_THR_CHUNK 1024
#endif
constexpr int THR_CHUNK = SDR_THR_CHUNK;
static_assert(THR_CHUNK % 8 == 0, "the chain walks eight frames at a time");

__global__ __launch_bounds__(256) void k_thresholds(sdr_frame_rec *__restrict__ recs, BandState *__restrict__ st,
                                                    int n_frames, int stride)
{
    __shared__ __attribute__((aligned(16))) float s_in[2][2][THR_CHUNK];   // [buffer][nf / dev][frame of the chunk]
    __shared__ __attribute__((aligned(16))) float s_old[2][2][THR_CHUNK];
    __shared__ __attribute__((aligned(16))) float s_sum[2][2][THR_CHUNK];
    __shared__ float s_ring[2][SDR_NOISE_WINDOW];
    const int band = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6;
    BandState *s = &st[band];
    sdr_frame_rec *r = recs + (size_t)band * stride;
    const int next0 = s->next;
    const float peak_threshold = s->peak_threshold;
    if (tid < SDR_NOISE_WINDOW) {
        s_ring[0][tid] = s->nf_ring[tid];
        s_ring[1][tid] = s->dev_ring[tid];
    }
    float sum = 0.f;  // (lanes 0 and 1 of wave 0: the chains)
    if (tid < 2)
        sum = tid ? s->dev_sum : s->nf_sum;
    __syncthreads();
    const int n_chunks = (n_frames + THR_CHUNK - 1) / THR_CHUNK;
    const int helpers = (int)blockDim.x - 64, hid = tid - 64;
    // chunk c's inputs and the values that leave the window at its frames (terms past the batch's end: zeros, never used)
    auto load_chunk = [&](int c, int first, int step) {
        const int base = c * THR_CHUNK;
        for (int j = first; j < THR_CHUNK; j += step) {
            const int f = base + j;
            float in0 = 0.f, in1 = 0.f, old0 = 0.f, old1 = 0.f;
            if (f < n_frames) {
                in0 = r[f].nf_in;
                in1 = r[f].dev_in;
                if (f >= SDR_NOISE_WINDOW) {
                    old0 = r[f - SDR_NOISE_WINDOW].nf_in;
                    old1 = r[f - SDR_NOISE_WINDOW].dev_in;
                } else {
                    const int slot = (next0 + f) % SDR_NOISE_WINDOW;
                    old0 = s_ring[0][slot];
                    old1 = s_ring[1][slot];
                }
            }
            s_in[c & 1][0][j] = in0;
            s_in[c & 1][1][j] = in1;
            s_old[c & 1][0][j] = old0;
            s_old[c & 1][1][j] = old1;
        }
    };
    auto finish_chunk = [&](int c, int first, int step) {
        const int base = c * THR_CHUNK, cnt = min(THR_CHUNK, n_frames - base);
        for (int j = first; j < cnt; j += step) {
            const int f = base + j;
            const float noiseFloor = __fdiv_rn(s_sum[c & 1][0][j], (float)SDR_NOISE_WINDOW);
            const float noiseDeviation = __fdiv_rn(s_sum[c & 1][1][j], (float)SDR_NOISE_WINDOW);
            r[f].noise_floor = noiseFloor;
            r[f].noise_dev = noiseDeviation;
            r[f].peak_thr = peak_threshold + noiseFloor;    // rx/receiver.go:385
            r[f].listen_thr = noiseFloor + noiseDeviation;  // rx/receiver.go:394
        }
    };
    if (n_chunks > 0)
        load_chunk(0, tid, (int)blockDim.x);
    __syncthreads();
    for (int c = 0; c <= n_chunks; c++) {
        if (wave == 0) {
            if (tid < 2 && c < n_chunks) {
                // eight frames at a time, the next eight's operands on their way while these are added
                const int cnt = min(THR_CHUNK, n_frames - c * THR_CHUNK);
                const float4 *in = reinterpret_cast<const float4 *>(s_in[c & 1][tid]);
                const float4 *old = reinterpret_cast<const float4 *>(s_old[c & 1][tid]);
                float4 *out = reinterpret_cast<float4 *>(s_sum[c & 1][tid]);
                float4 i0 = in[0], i1 = in[1], o0 = old[0], o1 = old[1];
                const int groups = cnt / 8;
                for (int q = 0; q < groups; q++) {
                    const float4 a0 = i0, a1 = i1, b0 = o0, b1 = o1;
                    const int qn = min(q + 1, THR_CHUNK / 8 - 1);
                    i0 = in[2 * qn];
                    i1 = in[2 * qn + 1];
                    o0 = old[2 * qn];
                    o1 = old[2 * qn + 1];
                    float4 r0, r1;
#define SDR_THR_PUT(dst, o, i)                              \
    sum = sum - (o); /* v.sumForMean -= v.values[v.next] */ \
    sum = sum + (i); /* v.sumForMean += v.values[v.next] */ \
    dst = sum
                    SDR_THR_PUT(r0.x, b0.x, a0.x);
                    SDR_THR_PUT(r0.y, b0.y, a0.y);
                    SDR_THR_PUT(r0.z, b0.z, a0.z);
                    SDR_THR_PUT(r0.w, b0.w, a0.w);
                    SDR_THR_PUT(r1.x, b1.x, a1.x);
                    SDR_THR_PUT(r1.y, b1.y, a1.y);
                    SDR_THR_PUT(r1.z, b1.z, a1.z);
                    SDR_THR_PUT(r1.w, b1.w, a1.w);
                    out[2 * q] = r0;
                    out[2 * q + 1] = r1;
                }
                for (int j = groups * 8; j < cnt; j++) {  // (the batch's last frames, fewer than eight)
                    SDR_THR_PUT(s_sum[c & 1][tid][j], s_old[c & 1][tid][j], s_in[c & 1][tid][j]);
                }
#undef SDR_THR_PUT
            }
        } else {
            if (c >= 1)
                finish_chunk(c - 1, hid, helpers);
            if (c + 1 < n_chunks)
                load_chunk(c + 1, hid, helpers);
        }
        __syncthreads();
    }
    // carry the window over to the next batch: slot (next0+f)%60 holds the last input written there
    if (tid < SDR_NOISE_WINDOW) {
        // last frame f in [0,n_frames) with (next0 + f) % 60 == tid
        const int off = (tid - next0 % SDR_NOISE_WINDOW + SDR_NOISE_WINDOW) % SDR_NOISE_WINDOW;  // smallest f
        if (off < n_frames) {
            const int f = off + ((n_frames - 1 - off) / SDR_NOISE_WINDOW) * SDR_NOISE_WINDOW;
            s->nf_ring[tid] = r[f].nf_in;
            s->dev_ring[tid] = r[f].dev_in;
        }
    }
    if (tid == 0)
        s->nf_sum = sum;
    if (tid == 1)
        s->dev_sum = sum;
    if (tid == 0)
        s->next = (next0 + n_frames) % SDR_NOISE_WINDOW;
}


// ---------------------------------------------------------------------------------------------
// Self-check of what the variance chains assume about the float64 matrix pipe (sdr_self_check, run by sdr_create once per
// device and process).  variance_consumer_mfma is bit-exact only if v_mfma_f64_4x4x4 with B = 1 adds its four terms to the
// accumulator one after the other, each step rounded to float64, in k order - established on MI355X silicon by
// tools/ubench_mfma_f64.hip and written down in no ISA document.  A part, stepping or firmware that evaluates the chain
// any other way (a tree, one rounding at the end, accumulator last) would shift variances by an ulp now and then and with
// them thresholds and keying edges, silently.  So: one wave feeds the pipe 64 dependent instructions of sixteen chains
// each (1024 quadruples) of wide-range terms - exponents spread over 2^-40 .. 2^40 and both signs, where every other
// association rounds differently - and compares the accumulator after EVERY instruction with the same chain on the
// vector ALU (v_add_f64, program order).  Any difference fails the creation of the bank: there is no second code path.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double probe_term(unsigned chain, unsigned round, unsigned k)
{
    unsigned long long h = ((unsigned long long)(round * 16u + chain) << 2 | k) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    const int e = (int)((h >> 52) % 81u) - 40;                      // binary exponent
    const unsigned long long mant = h & 0x000FFFFFFFFFFFFFull;       // 52 random mantissa bits
    const unsigned long long sign = (h >> 51) & 1ull ? 0x8000000000000000ull : 0ull;
    return __longlong_as_double((long long)(sign | ((unsigned long long)(1023 + e) << 52) | mant));
}

// order: 0 = the chain the library assumes; 1 / 2 = a pairwise tree / the accumulator added last - what the pipe must NOT
// compute (SDR_SELF_CHECK_ORDER, tests only: the check has to fail against them, or it checks nothing)
__global__ __launch_bounds__(64) void k_mfma_order_probe(unsigned *__restrict__ mismatches, int order)
{
    const unsigned lane = threadIdx.x & 63u;
    // operand layout (probed, see variance_consumer_mfma): lane L supplies term k = L >> 4 of chain L & 15; the sum of
    // chain c sits in lanes 16 (c & 3) + 4 (c >> 2) + j, j = 0..3
    const unsigned a_chain = lane & 15u, a_k = lane >> 4;
    const unsigned my_chain = 4u * ((lane >> 2) & 3u) + (lane >> 4);
    double acc = 0.0;
    unsigned bad = 0;
    for (unsigned r = 0; r < 64u; r++) {
        const double before = acc;
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(probe_term(a_chain, r, a_k), 1.0, acc, 0, 0, 0);
        double t[4];
        for (unsigned k = 0; k < 4u; k++) {
            t[k] = probe_term(my_chain, r, k);
            asm volatile("" : "+v"(t[k]));  // (the additions below stay as written)
        }
        double want;
        if (order == 0)
            want = (((before + t[0]) + t[1]) + t[2]) + t[3];  // dsp/fft.go:246-248: sum += term, one rounding per step
        else if (order == 1)
            want = before + ((t[0] + t[1]) + (t[2] + t[3]));
        else
            want = (((t[0] + t[1]) + t[2]) + t[3]) + before;
        bad += __double_as_longlong(acc) != __double_as_longlong(want) ? 1u : 0u;
    }
    if (bad)
        atomicAdd(mismatches, bad);
}

hipError_t launch_mfma_order_probe(unsigned *mismatches, int order, hipStream_t stream)
{
    hipLaunchKernelGGL(k_mfma_order_probe, dim3(1), dim3(64), 0, stream, mismatches, order);
    return hipGetLastError();
}

hipError_t launch_window_means(const float *psd, double *win_mean, NoiseGeom g, int n_frames, int n_bands, int stride,
                               hipStream_t stream)
{
    // (rounds 2 - 3, 2048-frame batches: config 3's 160 workgroups of one window each; two windows back to back per
    // workgroup, 80 workgroups: 0.050 instead of 0.030 ms standalone and 0.226 - 0.238 instead of 0.222 - 0.225 ms per
    // pipelined step; four windows: 0.093 ms, 0.253)
    const bool half = g.n <= 8192;  // the FFT kernel's workgroups are 512 threads or fewer: see WM_GROUPS_HALF
    const int groups = half ? WM_GROUPS_HALF : WM_GROUPS;
    const int per_band = ((n_frames + TILE - 1) / TILE + groups - 1) / groups;
    // windows per workgroup.  Whole-CU workgroups (N = 16384): fewer, longer-lived ones hold less CU time as long as there
    // are enough of them for the kernel's latency - about 64: config 3 at 8192 frames per batch (320 window-groups), one box,
    // interleaved twice, 1 / 2 / 5 / 10 windows per workgroup 166.0 / 167.5 / 168.8 / 167.7 and 166.2 / 167.7 / 168.8 / 168.0
    // GS/s; at 2048 frames (80 window-groups) five windows per workgroup leave 16 workgroups: 105 against 162.5.  Half-size
    // workgroups (config 5's share: 1280 of them) stay at one window each: two cost 5 % there (round 4, first half).
    // (4096 frames: two windows per workgroup 171.1 against 172.3 with one - several only pay from about four on)
    int wpb = half ? (per_band * n_bands * g.n_windows) / (512 * WM_GROUPS / groups) : (per_band * n_bands * g.n_windows) / 64;
    if (!half && wpb < 4)
        wpb = 1;
    wpb = wpb < 1 ? 1 : (wpb > g.n_windows ? g.n_windows : wpb);
    static const int wpb_env = getenv("SDR_WM_WPB") ? atoi(getenv("SDR_WM_WPB")) : 0;  // (development)
    if (wpb_env > 0)
        wpb = wpb_env > g.n_windows ? g.n_windows : wpb_env;

    const dim3 grid(per_band, (g.n_windows + wpb - 1) / wpb, n_bands);
    if (half)
        launch_kernel((k_window_means<WM_GROUPS_HALF, WindowRingHalf>), grid, dim3(64 * (MAX_WAVES / WM_GROUPS) * WM_GROUPS_HALF), 0, stream, psd, win_mean, g,
                      n_frames, stride, wpb);
    else
        launch_kernel((k_window_means<WM_GROUPS, WindowRing>), grid, dim3(CHAIN_THREADS), 0, stream, psd, win_mean, g, n_frames, stride, wpb);
    return hipGetLastError();
}

hipError_t launch_noise_stats(const float *psd, const double *win_mean, sdr_frame_rec *recs, NoiseGeom g, int n_frames,
                              int n_bands, int stride, hipStream_t stream)
{
    // long batches: two vector-ALU chain groups per workgroup (less CU time); short ones: the matrix-pipe kernel (less
    // latency) - see NS_GROUPS_VALU
    static const int force = getenv("SDR_VAR_MFMA") ? atoi(getenv("SDR_VAR_MFMA")) : -1;  // (development)
    const bool mfma = force >= 0 ? force != 0 : n_frames < 4096;
    const int groups_of_64 = (n_frames + TILE - 1) / TILE;
    if (mfma)
        launch_kernel(k_noise_stats<true>, dim3(groups_of_64, n_bands), dim3(CHAIN_THREADS), 0, stream, psd, win_mean, recs, g, n_frames, stride);
    else
        launch_kernel(k_noise_stats<false>, dim3((groups_of_64 + NS_GROUPS_VALU - 1) / NS_GROUPS_VALU, n_bands), dim3(CHAIN_THREADS), 0, stream, psd,
                      win_mean, recs, g, n_frames, stride);
    return hipGetLastError();
}

hipError_t launch_thresholds(sdr_frame_rec *recs, BandState *st, int n_frames, int n_bands, int stride,
                             hipStream_t stream)
{
    launch_kernel(k_thresholds, dim3(n_bands), dim3(256), 0, stream, recs, st, n_frames, stride);
    return hipGetLastError();
}


}  // namespace sdr
