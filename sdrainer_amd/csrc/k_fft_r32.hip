// k_fft_r32.hip — the FFT -> PSD kernel for N = 16384 in its 32-points-per-thread form (plan: fft_r32.h).
// dsp/fft.go:23-37 IQToSpectrumAndPSD (the psd half), :59-69 setSamplesFromIQ, :54-57 fftshift, go-dsp fft.FFT.
//
// Why a second kernel.  k_fft_psd<14> (1024 threads x 16 points, one workgroup per CU, 126 of 128 VGPRs) runs a frame's
// phases one after the other - 13 000 of a frame's 42 000 clocks are the wait for its own 128 KB of input, with every
// wave of the CU parked (profiles/r04_fft_phase_order.txt) - and has neither the registers nor the LDS to fetch the next
// frame meanwhile.  Here a workgroup is 512 threads x 32 points at two waves per SIMD: 256 VGPRs per thread, 128 for the
// frame's complex128 state, 64 for the NEXT frame's complex64 samples, which are requested right after the current
// frame's have been widened and have the whole frame to arrive.  A workgroup takes `fpw` consecutive frames; only its
// first frame's input is waited for.
//
// Frame schedule (per wave; B = workgroup barrier):
//   widen the prefetched samples (f32 -> f64), request the next frame's
//   pass 0   stages 1-5, twiddles wave-uniform (scalar loads)
//   E0       cross-wave exchange through LDS, real then imaginary parts (B write B read B write B read)
//   pass 1   stages 6-10, twiddles from the 15.5 KB table in LDS
//   E1       exchange inside the wave's own LDS block (no barrier)
//   pass 2   stages 11-14, twiddles streamed from L2 (64 consecutive entries per wave instruction)
//   psd = f32(re^2 + im^2) -> LDS row (fft-shifted, swizzled) B -> 16-byte runs per lane -> global (1 KB per wave
//   instruction) + the listeners' tap straight from the row
// Compiled with -ffp-contract=off: the butterflies are the reference's ten float64 operations, no FMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "fft_r32.h"
#include "sdr_device.h"

#if !defined(SDR_R32_IN_AUX)
#define SDR_R32_IN_AUX 0  // cache policy bits of the input loads (2 = nt)
#endif
#if !defined(SDR_R32_CHUNK0)
#define SDR_R32_CHUNK0 4  // pass 0: scalar twiddles fetched per chunk (a stage's sixteen rows at once are 64 SGPRs)
#endif
// The next frame's 32 loads per thread are dealt over the frame (loads per program point; seventeen points: behind the
// widening, behind each of pass 0's five stages, at the four steps of E0, in front of pass 1, behind pass 1's stages 0-3,
// behind pass 2, at the row).  Issued in one go - 256 wave instructions, 128 KB per CU - they do not overlap anything: a
// wave cannot issue a vector memory instruction while the CU's memory pipeline is backed up, and it takes the pipeline
// 5 us to work a frame off (first build: 13 000 of a frame's 44 000 clocks went into issuing them).  Two at every point
// but the first, where the previous frame's row starts to leave: measured 1 % faster than {4, 2 x 14, 0, 0} (nothing
// behind pass 1's stage 3, so that the last group lands before pass 2's twiddle loads, which return behind it), and the
// kernel compiles without a spilled vector register this way.
#if !defined(SDR_R32_PF_PLAN)
#define SDR_R32_PF_PLAN {0, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2}
#endif
// Issue priority, alternating between the two waves of a SIMD stage by stage (1; 0 = leave it to the hardware).  The
// arbiter serves the OLDER of two waves first: waves 0-3 ran a pass at the pace of a wave alone on its SIMD (6.3 clocks per
// float64 instruction) while their partners got 40 % of that, finished the frame 6 000 clocks early and sat at the row's
// barrier while the partners finished alone, again at the single-wave pace.  Two waves issuing side by side manage 4.5 -
// 4.8 clocks per instruction between them (tools/ubench_f64: 29 against 22 T lane-operations/s), but only while both have
// arithmetic to issue: taking turns at the higher priority keeps them within a stage of each other.
#if !defined(SDR_R32_PRIO)
#define SDR_R32_PRIO 1
#endif
#if !defined(SDR_R32_DEPTH2)
#define SDR_R32_DEPTH2 1  // pass 2: twiddles requested this many chunks ahead of their butterflies (1 or 2)
#endif
#if !defined(SDR_R32_CHUNK1)
#define SDR_R32_CHUNK1 2  // twiddles fetched per chunk in pass 1 (0: the compiler decides - and spills; 4 / 4 / 4 spills prefetched samples)
#endif
#if !defined(SDR_R32_PRIO_FLIP2)
#define SDR_R32_PRIO_FLIP2 0
#endif
#if !defined(SDR_R32_PRIO_FLIP1)
#define SDR_R32_PRIO_FLIP1 1  // 1: pass 1's stages favour the other half of the workgroup than pass 0's do (0: waves 0-3 lead
                              // in three of five stages of both passes and then idle at the row's barrier: 1.5 % slower)
#endif
#if !defined(SDR_R32_CHUNK2)
#define SDR_R32_CHUNK2 4
#endif

namespace sdr {
namespace r32 {

using fft32::cplx;
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    // inputs are made provably wave-uniform first, otherwise the descriptor is rebuilt per lane (waterfall)
    const unsigned long long b = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// orders one wave's LDS stores before its later LDS loads (and the reverse): lanes of ONE wave exchanging data, so
// wavefront scope is the scope the memory model asks for
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Development aid (tools/fft_r32_bench.hip only): -DSDR_R32_PHASES=<workgroup> makes every wave of that workgroup read the
// shader clock at each phase boundary of its SECOND frame (steady state: the frame's input was prefetched) into SGPRs and
// store the stamps at its last instruction.
enum R32Stamp { RS_TOP = 0, RS_LANDED, RS_CVT, RS_FLUSHED, RS_WIDENED, RS_PASS0, RS_E0, RS_PASS1, RS_E1, RS_PASS2, RS_ROW, RS_STORED, RS_COUNT };
#if defined(SDR_R32_PHASES)
// (the stamps live in the lanes of ONE vector register: the kernel has neither SGPRs nor VGPRs to spare.  Every wave of
// every workgroup takes them, unconditionally: a branch around each stamp cut the unrolled frame into a dozen basic blocks
// and the register allocator spilled 169 registers.  The stamped build is NOT a timing build.)
struct R32Stamps {
    unsigned v;
    bool on;
};
#define SDR_R32_STAMP(st, k)                                                                                \
    do {                                                                                                    \
        {                                                                                                   \
            const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();                                   \
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"((st).v) : "s"(now_), "n"((int)(k)));         \
        }                                                                                                   \
    } while (0)
#else
struct R32Stamps {
};
#define SDR_R32_STAMP(st, k) \
    do {                     \
    } while (0)
#endif

constexpr int kPfPoints = 17;
constexpr int kPfPlan[kPfPoints] = SDR_R32_PF_PLAN;
constexpr int pf_begin(int point)
{
    int b = 0;
    for (int i = 0; i < point; i++)
        b += kPfPlan[i];
    return b;
}
static_assert(pf_begin(kPfPoints) == fft32::R, "the prefetch plan must cover the frame's 32 slots");
// the previous frame's eight row stores per thread, dealt over the first program points
constexpr int kStorePoints = 7;
#if !defined(SDR_R32_ST_PLAN)
#define SDR_R32_ST_PLAN {2, 1, 1, 1, 1, 1, 1}
#endif
constexpr int kStPlan[kStorePoints] = SDR_R32_ST_PLAN;
constexpr int st_begin(int point)
{
    int b = 0;
    for (int i = 0; i < point; i++)
        b += kStPlan[i];
    return b;
}
static_assert(st_begin(kStorePoints) == fft32::N / 4 / fft32::T, "the store plan must cover the row's eight runs per thread");
enum PfPoint { PF_WIDENED = 0, PF_PASS0 = 1 /* +q */, PF_E0 = 6 /* +step */, PF_PRE1 = 10, PF_PASS1 = 11 /* +q */, PF_POST2 = 15, PF_ROW = 16 };

constexpr int kTw1LdsBytes = fft32::kTw1Entries * 16;
constexpr int kSoftCounters = 4;
constexpr int kLdsBytes = fft32::kExchangeBytes + kTw1LdsBytes + kSoftCounters * 4;
// "Soft" barriers for the write-after-read hazards of the shared exchange area: a wave ARRIVES (one LDS add) when it has
// read what it wanted and WAITS (polls the counter) right before it writes - a whole pass later, when everybody has long
// arrived.  An s_barrier there would make the wave that is ahead sit out the other's pass: with two waves per SIMD the
// overlap of one wave's exchange with the other's arithmetic is all the latency hiding there is.  The counters only
// grow: eight arrivals per frame.
enum SoftId { SOFT_ROW = 0, SOFT_E0 = 1, SOFT_E1 = 2 };
// (Both are single asm statements: as C++ - a poll loop, a one-lane branch around the add - they cut the unrolled frame
// into basic blocks, and the register allocator answered with 72 spilled registers, the prefetched samples first.)
__device__ __forceinline__ void soft_arrive(unsigned *cnt)
{
    // this wave's LDS reads have returned their data (lgkmcnt(0)); then one lane adds 1
    unsigned long long save;
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "ds_add_u32 %1, %2\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save)
                 : "v"((unsigned)(unsigned long long)cnt), "v"(1u)
                 : "memory");
}
__device__ __forceinline__ void soft_wait(unsigned *cnt, unsigned target)
{
    unsigned seen;
    asm volatile("1:\n\t"
                 "ds_read_b32 %0, %1\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_cmp_lt_u32_e32 vcc, %0, %2\n\t"
                 "s_cbranch_vccz 2f\n\t"
                 "s_sleep 1\n\t"
                 "s_branch 1b\n\t"
                 "2:"
                 : "=&v"(seen)
                 : "v"((unsigned)(unsigned long long)cnt), "v"(target)
                 : "vcc", "memory");
}

// s_setprio takes an immediate, so the choice is a scalar branch - inside ONE asm statement (a C++ branch would cut the
// frame into blocks) and on M0, which the kernel loads with 0 (waves 0-3) or 1 (waves 4-7) at its top: an SGPR operand
// per call kept two more scalar registers live through a frame that has none to spare (90 vector registers spilled).
// Nothing else in this kernel uses M0; should that ever change, the priorities would be off, nothing else.
// SCC is saved and restored inside the statement instead of being declared clobbered: with the clobber the same build
// spilled 79 vector registers (with it the compiler... whatever it does, the prefetched samples went to scratch).
__device__ __forceinline__ void set_prio(int stage)  // (a constant wherever it is called: the test folds)
{
#if SDR_R32_PRIO
    unsigned keep;
    if (stage & 1)
        asm volatile("s_cselect_b32 %0, 1, 0\n\ts_cmp_lg_u32 m0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 1\n\ts_branch 2f\n\t1:\n\ts_setprio 0\n\t2:\n\t"
                     "s_cmp_lg_u32 %0, 0"
                     : "=&s"(keep));
    else
        asm volatile("s_cselect_b32 %0, 1, 0\n\ts_cmp_eq_u32 m0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 1\n\ts_branch 2f\n\t1:\n\ts_setprio 0\n\t2:\n\t"
                     "s_cmp_lg_u32 %0, 0"
                     : "=&s"(keep));
#endif
}

template <int E, int P>
__device__ __forceinline__ void ex_write(const double (&x)[32], int t, double *area)
{
    const int base = fft32::map_addr_thread<E, P>(t);
#pragma unroll
    for (int s = 0; s < 32; s++)
        area[base + fft32::map_addr_slot<E>(P, s)] = x[s];
}
template <int E, int P>
__device__ __forceinline__ void ex_read(double (&x)[32], int t, const double *area)
{
    const int base = fft32::map_addr_thread<E, P>(t);
#pragma unroll
    for (int s = 0; s < 32; s++)
        x[s] = area[base + fft32::map_addr_slot<E>(P, s)];
}

__global__ __launch_bounds__(fft32::T, 2) void k_fft_r32(const float *__restrict__ iq_arg, const BatchCursor *__restrict__ cur,
                                                          const cplx *__restrict__ tw, float *__restrict__ psd, int in_stride,
                                                          int out_stride, int n_frames, int fpw, const int *__restrict__ tap_bins,
                                                          float *__restrict__ tap_out, int n_tap, int tap_stride,
                                                          float *__restrict__ tap_wide, int *__restrict__ tap_used)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace fft32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *ex = reinterpret_cast<double *>(smem);
    const unsigned char *tw1_lds = smem + kExchangeBytes;
    const float *__restrict__ iq = cur ? cur->iq : iq_arg;  // graph replay: the batch's input pointer lives in device memory
    const int frame0 = blockIdx.x * fpw;
    const int frame_end = min(frame0 + fpw, n_frames);
    const size_t in_band = (size_t)blockIdx.y * in_stride, out_band = (size_t)blockIdx.y * out_stride;
    const int tid = threadIdx.x;
    const rsrc_t twr = make_rsrc(tw, (unsigned)(kTwTotal * sizeof(cplx)));

    // setSamplesFromIQ's reads, one frame ahead: slot m <- sample tid + 512 * brev5(m) (fft_r32.h: pass 0's thread part of
    // the sample number is the thread id), 8 bytes per lane, 512 contiguous bytes per wave instruction
    u32x2 pf[R];
    // slots [m0, m1) of `frame`; a frame past the workgroup's last one gets a descriptor of zero bytes: the loads return
    // zeros without touching memory, and the frame's code stays free of branches
    auto fetch = [&](int frame, int t, int m0, int m1) {
        const rsrc_t xrs = make_rsrc(iq + (in_band + frame) * (size_t)N * 2, frame < frame_end ? N * 8u : 0u);
        const unsigned voff = (unsigned)thread_sample(t) * 8u;
#pragma unroll
        for (int m = 0; m < R; m++)
            if (m >= m0 && m < m1)
                pf[m] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, slot_sample(m) * 8, SDR_R32_IN_AUX);
    };
    fetch(frame0, tid, 0, R);
    // pass 1's twiddle block -> LDS, once per workgroup
    for (int i = tid; i < kTw1Entries; i += T) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(twr, (unsigned)i * 16u, kTw1 * 16, 0);
        *reinterpret_cast<u32x4 *>(smem + kExchangeBytes + i * 16) = w;
    }
    // the listeners' bins: slot l of the band belongs to thread l (launch_fft_r32 refuses more than 512 slots in use: those
    // banks run the 16-point kernel)
    const bool reg_tap = n_tap > 0 && n_tap <= T;
    const int my_bin = (reg_tap && tid < n_tap) ? tap_bins[(size_t)blockIdx.y * tap_stride + tid] : -1;
    // The WIDE tap (k_peaks.hip k_cum_refine): psd at bin - 1, bin, bin + 1 of every listener, [frame][slot][4] - the exact
    // cumulation is wanted at the signals' bins and their neighbours, and read there it is four contiguous kilobytes per
    // frame instead of a hundred scattered 64-byte sectors per candidate.  Which bins these rows hold is recorded with them
    // (the bank's tap bins may have changed by the time the refinement of this batch runs).
    if (tap_used && blockIdx.x == 0 && tid < tap_stride)
        tap_used[(size_t)blockIdx.y * tap_stride + tid] = my_bin;
    unsigned *soft = reinterpret_cast<unsigned *>(smem + kExchangeBytes + kTw1LdsBytes);  // arrival counters, see soft_wait
    if (tid < kSoftCounters)
        soft[tid] = 0;
    __syncthreads();
    R32Stamps st;
#if defined(SDR_R32_PHASES)
    st.on = false;
    st.v = 0;
#endif
    // A frame's psd row leaves the CU at the top of the NEXT frame: its eight 16-byte runs per thread (and the thread's
    // tap value) wait in registers while the next frame's samples are widened - a phase bound by the conversions' issue
    // rate, with the memory pipeline idle - instead of holding every wave at the end of the frame (first build: 2 700
    // clocks per frame for the CU to take 64 KB of stores).
    u32x4 sv[N / 4 / T];
    float tapv = 0.f;
#pragma unroll
    for (int j = 0; j < N / 4 / T; j++)
        sv[j] = u32x4{0u, 0u, 0u, 0u};
    // (frame < frame0: descriptors of zero bytes, nothing is stored.  The run's offset goes into the VECTOR offset, the
    // scalar offset stays the literal 0: a 16-byte buffer store reads its data registers some cycles after it issues; with
    // an immediate scalar offset hipcc pads a following VALU write of those registers with wait states, with an SGPR
    // offset it assumes no hazard - and on gfx950 there is one: built that way, 0.3 % of the psd words of every launch
    // came out as the next run's LDS address, which the compiler had put into the first data register behind the store.)
    // stores [j0, j1) of the eight (tap: the tap value with the last one)
    auto flush_row = [&](int frame, int t, int j0, int j1) {
        const bool live = frame >= frame0;
        const rsrc_t pdr = make_rsrc(psd + (out_band + (live ? frame : frame0)) * (size_t)N, live ? N * 4u : 0u);
#pragma unroll
        for (int j = 0; j < N / 4 / T; j++)
            if (j >= j0 && j < j1)
                __builtin_amdgcn_raw_buffer_store_b128(sv[j], pdr, (unsigned)(t + T * j) * 16u, 0, 0);
        if (j1 == N / 4 / T) {
            // the tap (rx/receiver.go:393-394: spectrum[SignalBin] per listener and frame; the dB projection is applied
            // where it is consumed, k_listen.hip): slot l's value from thread l; lanes past n_tap fall outside the descriptor
            const rsrc_t tpr = make_rsrc(tap_out + (out_band + (live ? frame : frame0)) * (size_t)tap_stride,
                                         (live && reg_tap) ? (unsigned)n_tap * 4u : 0u);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tapv), tpr, (unsigned)t * 4u, 0, 0);
        }
    };

#if SDR_R32_PRIO
    asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(tid >> 8)) : "m0");  // waves 0-3 / 4-7: a SIMD has one of each
#endif
    int it = 0;  // frames this workgroup has finished (the soft barriers' targets count in it)
#pragma nounroll
    for (int frame = frame0; frame < frame_end; frame++, it++) {
        // (everything derived from the thread id is loop-invariant; hoisted, it would sit in registers the frame needs)
        int t = tid;
        asm volatile("" : "+v"(t));
#if defined(SDR_R32_PHASES)
        st.on = blockIdx.x == SDR_R32_PHASES && blockIdx.y == 0 && frame == frame0 + 1;
#endif
        SDR_R32_STAMP(st, RS_TOP);
#if defined(SDR_R32_PHASES)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        SDR_R32_STAMP(st, RS_LANDED);
        double xr[R], xi[R];
#pragma unroll
        for (int m = 0; m < R; m++) {  // dsp/fft.go:59-69 setSamplesFromIQ: widen, exact
            xr[m] = (double)__uint_as_float(pf[m].x);
            xi[m] = (double)__uint_as_float(pf[m].y);
        }
        // (the next frame's requests go out BEHIND the conversions: hoisted above them - they depend on nothing - both
        // frames' samples would be live at once, 64 registers more than there are; the pins keep the conversions from
        // being sunk below them)
#pragma unroll
        for (int m = 0; m < R; m++)
            asm volatile("" : "+v"(xr[m]), "+v"(xi[m]));
        __builtin_amdgcn_sched_barrier(0);
        // (the previous frame's row goes out a store or two at a time, at the first prefetch points: issued in one go,
        // the CU takes 5 000 clocks to accept a frame's 72 stores, and the waves that come second sit that out)
        auto pf_point = [&](int point) {
            __builtin_amdgcn_sched_barrier(0);
            if (point < kStorePoints)
                flush_row(frame - 1, t, st_begin(point), st_begin(point + 1));
            fetch(frame + 1, t, pf_begin(point), pf_begin(point + 1));
            __builtin_amdgcn_sched_barrier(0);
        };
#if defined(SDR_R32_PHASES)  // (the first point taken apart for its stamps)
        SDR_R32_STAMP(st, RS_CVT);
        __builtin_amdgcn_sched_barrier(0);
        flush_row(frame - 1, t, st_begin(PF_WIDENED), st_begin(PF_WIDENED + 1));
        SDR_R32_STAMP(st, RS_FLUSHED);
        fetch(frame + 1, t, pf_begin(PF_WIDENED), pf_begin(PF_WIDENED + 1));
        __builtin_amdgcn_sched_barrier(0);
#else
        pf_point(PF_WIDENED);
#endif
        SDR_R32_STAMP(st, RS_WIDENED);

        // pass 0; behind its stage 3 everybody must be out of the previous frame's psd row (it shares the area with E0):
        // they said so long ago (SOFT_ROW); the last stage then writes each finished pair's real parts into E0 while
        // the other butterflies compute
        {
            const int wbase = map_addr_thread<0, 0>(t);
            run_pass<5, true, SDR_R32_CHUNK0>(
                xr, xi, [tw](int row, int) { return tw[kTw0 + row]; },
                [&](int q) {
                    set_prio(q);
                    pf_point(PF_PASS0 + q);
                    if (q == 3)
                        soft_wait(soft + SOFT_ROW, (unsigned)(NWAVES * it));
                },
                [&](int a, int b) {
                    ex[wbase + map_addr_slot<0>(0, a)] = xr[a];
                    ex[wbase + map_addr_slot<0>(0, b)] = xr[b];
                });
        }
        SDR_R32_STAMP(st, RS_PASS0);

        // E0: real parts (written above) read, imaginary parts written and read
        pf_point(PF_E0 + 0);
        __syncthreads();
        ex_read<0, 1>(xr, t, ex);
        pf_point(PF_E0 + 1);
        __syncthreads();
        ex_write<0, 0>(xi, t, ex);
        pf_point(PF_E0 + 2);
        __syncthreads();
        ex_read<0, 1>(xi, t, ex);
        pf_point(PF_E0 + 3);
        soft_arrive(soft + SOFT_E0);  // this wave has what it wanted from E0 (E1 writes blocks other waves read here)
        SDR_R32_STAMP(st, RS_E0);
        pf_point(PF_PRE1);

        // pass 1 (twiddles from LDS); its last stage writes the real parts into E1, the wave's own block - once every
        // wave has left E0 (SOFT_E0, waited for behind stage 3)
        {
            const unsigned char *row0 = tw1_lds + tw1_lo(t) * 16;
            const int wbase = map_addr_thread<1, 1>(t);
            run_pass<5, false, SDR_R32_CHUNK1>(
                xr, xi, [row0](int row, int) { return *reinterpret_cast<const cplx *>(row0 + row * 512); },
                [&](int q) {
                    set_prio(q + SDR_R32_PRIO_FLIP1);
                    if (q < 4)
                        pf_point(PF_PASS1 + q);
                    if (q == 3)
                        soft_wait(soft + SOFT_E0, (unsigned)(NWAVES * (it + 1)));
                },
                [&](int a, int b) {
                    ex[wbase + map_addr_slot<1>(1, a)] = xr[a];
                    ex[wbase + map_addr_slot<1>(1, b)] = xr[b];
                });
        }
        SDR_R32_STAMP(st, RS_PASS1);
        // pass 2's twiddles stream from L2; its first two are requested here, in front of E1
        const unsigned tw2_off = (unsigned)tw2_pos(t, 0) * 16u;
        auto tw2 = [twr, tw2_off](int row, int u) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(twr, tw2_off, (kTw2 + row * 1024 + u * 512) * 16, 0);
            cplx r;
            r.x = __hiloint2double((int)w.y, (int)w.x);
            r.y = __hiloint2double((int)w.w, (int)w.z);
            return r;
        };
#if SDR_R32_DEPTH2 == 2
        FirstChunks2<4, SDR_R32_CHUNK2> tw2_first;
        first_chunks2<4, false, SDR_R32_CHUNK2>(tw2, tw2_first);
#else
        FirstChunk<4, SDR_R32_CHUNK2> tw2_first;
        first_chunk<4, false, SDR_R32_CHUNK2>(tw2, tw2_first);
#endif
        // E1: inside the wave's own block
        wave_sync();
        ex_read<1, 2>(xr, t, ex);
        wave_sync();
        ex_write<1, 1>(xi, t, ex);
        wave_sync();
        ex_read<1, 2>(xi, t, ex);
        soft_arrive(soft + SOFT_E1);  // this wave is out of E1 (the psd row is written all over the area)
        SDR_R32_STAMP(st, RS_E1);

#if SDR_R32_DEPTH2 == 2
        run_pass2_with<4, false, SDR_R32_CHUNK2>(xr, xi, tw2_first, tw2, [&](int q) { set_prio(q + SDR_R32_PRIO_FLIP2); });
#else
        run_pass_with<4, false, SDR_R32_CHUNK2>(xr, xi, tw2_first, tw2, [&](int q) { set_prio(q + SDR_R32_PRIO_FLIP2); });
#endif
        pf_point(PF_POST2);
        SDR_R32_STAMP(st, RS_PASS2);

        // Epilogue (dsp/fft.go:54-57 fftshift, :71-73 PSD[float32]): psd[k] = float32(re^2 + im^2), two multiplies and
        // an add in float64, rounded once; into the LDS row at spectrum index k = bin ^ N/2 once every wave is out of E1
        {
            float p[R];
#pragma unroll
            for (int s = 0; s < R; s++)
                p[s] = (float)(xr[s] * xr[s] + xi[s] * xi[s]);
            soft_wait(soft + SOFT_E1, (unsigned)(NWAVES * (it + 1)));
            constexpr int SLOT_MASK = slot_part(2, R - 1);
            static_assert((SLOT_MASK & 0xfc) == 0, "the row swizzle reads thread bits only");
            const int tk = thread_part<2>(t) ^ ((N / 2) & ~SLOT_MASK);
            unsigned char *rowp = smem + row_word(tk) * 4;
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int sk = slot_part(2, s) ^ ((N / 2) & SLOT_MASK);
                *reinterpret_cast<float *>(rowp + sk * 4) = p[s];
            }
        }
        __syncthreads();
        SDR_R32_STAMP(st, RS_ROW);
        pf_point(PF_ROW);
        {
            // this thread's eight 16-byte runs of the row and its listener's bin, into registers; they leave at the top
            // of the next frame (or behind the loop)
#pragma unroll
            for (int j = 0; j < N / 4 / T; j++)
                sv[j] = *reinterpret_cast<const u32x4 *>(smem + row_word(4 * (t + T * j)) * 4);
            const float *row = reinterpret_cast<const float *>(smem);
            // (a free slot, bin -1, reads word 0 and stores 0)
            const float tv = row[row_word(my_bin & (N - 1))];
            tapv = my_bin >= 0 ? tv : 0.0f;
            // the wide tap: (left neighbour, the bin, right neighbour, 0) as ONE 16-byte store per slot, by the waves that
            // hold slots, at once - the memory pipeline is idle here (the prefetch plan leaves this point empty), and what a
            // store costs the frame is its issue: as three dword stores by every wave this was 2.4 % of the kernel.  Lanes
            // without a slot aim past the descriptor (no per-lane branch).  A neighbour that does not exist (bin 0's left,
            // bin N - 1's right) wraps to a word nobody reads.
            if (__builtin_amdgcn_readfirstlane(t) < n_tap && tap_wide) {  // (wave-uniform)
                const rsrc_t wr = make_rsrc(tap_wide + (out_band + frame) * (size_t)(4 * tap_stride), reg_tap ? (unsigned)tap_stride * 16u : 0u);
                u32x4 wv;
                wv.x = __float_as_uint(row[row_word((my_bin - 1) & (N - 1))]);
                wv.y = __float_as_uint(tv);
                wv.z = __float_as_uint(row[row_word((my_bin + 1) & (N - 1))]);
                wv.w = 0u;
                __builtin_amdgcn_raw_buffer_store_b128(wv, wr, t < n_tap ? (unsigned)t * 16u : 0x7ffffff0u, 0, 0);
            }
        }
        soft_arrive(soft + SOFT_ROW);  // this wave is out of the row
        SDR_R32_STAMP(st, RS_STORED);
#if defined(SDR_R32_PHASES)
        if (st.on && (threadIdx.x & 63) < RS_COUNT)
            g_r32_phases[threadIdx.x >> 6][threadIdx.x & 63] = st.v;
#endif
    }
    flush_row(frame_end - 1, tid, 0, N / 4 / T);
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace r32

// Frames per workgroup of the R32 kernel (SDR_FFT_R32_FPW overrides).  A workgroup's first frame is not prefetched and
// its twiddle block is loaded once, so more frames per workgroup are cheaper frames (standalone, 8192 frames: 0.517 ms at
// four, 0.494 at eight, 0.481 at thirty-two) - but the tail stages hold CUs while a launch runs, and a launch that is two
// or three rounds of workgroups over the CUs it gets ends on a round that is nearly empty: in the pipeline about a
// thousand workgroups per launch is what measures best, four frames each at most (config 3, GS/s by frames per batch and
// frames per workgroup: 8192 - 1: 170, 2: 189, 3: 190, 4: 195, 6: 189, 8: 187; 3072 - 3: 170, 4: 155; 2048 - 1: 154, 2: 173,
// 3: 174, 4: 155; 1024 - 1: 141, 2: 163, the sixteen-point kernel 152; 512 - 1: 98, 2: 94, the sixteen-point kernel 104:
// launch_fft sends batches of fewer than 1024 frames there).
static int r32_fpw(long total_frames)
{
    static const int forced = [] {
        const char *e = getenv("SDR_FFT_R32_FPW");
        return e ? std::max(1, std::min(atoi(e), 1024)) : 0;
    }();
    if (forced)
        return forced;
    return total_frames >= 4096 ? 4 : total_frames >= 2560 ? 3 : 2;
}

int r32_twiddle_count() { return fft32::kTwTotal; }
void r32_build_twiddles(const double *wre, const double *wim, fft64::cplx *out) { fft32::build_twiddles(wre, wim, out); }

hipError_t launch_fft_r32(const float *iq, const BatchCursor *cur, const fft64::cplx *tw, float *psd, int n_frames, int n_bands,
                          int in_stride, int out_stride, FftTap tap, hipStream_t stream)
{
    constexpr int kMaxDevices = 64;
    static std::once_flag attr_once[kMaxDevices];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    if (dev < 0 || dev >= kMaxDevices)
        return hipErrorInvalidDevice;
    hipError_t attr_err = hipSuccess;
    std::call_once(attr_once[dev], [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(&r32::k_fft_r32), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       r32::kLdsBytes);
    });
    if (attr_err != hipSuccess)
        return attr_err;
    if (n_frames <= 0 || n_bands <= 0)
        return hipSuccess;
    if (tap.n > fft32::T)
        return hipErrorInvalidValue;  // (launch_fft never asks: one listener slot per thread)
    // a workgroup's frames are consecutive; never fewer workgroups than the chip has CUs
    int fpw = r32_fpw((long)n_frames * n_bands);
    while (fpw > 1 && (long)((n_frames + fpw - 1) / fpw) * n_bands < 256)
        fpw /= 2;
    launch_kernel(r32::k_fft_r32, dim3((n_frames + fpw - 1) / fpw, n_bands), dim3(fft32::T), r32::kLdsBytes, stream, iq, cur, tw, psd,
                  in_stride, out_stride, n_frames, fpw, tap.bins, tap.out, tap.n, tap.stride, tap.wide, tap.used);
    return hipGetLastError();
}

}  // namespace sdr
