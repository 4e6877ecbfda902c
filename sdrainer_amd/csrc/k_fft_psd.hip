// k_fft_psd.hip — the dominant kernel: IQ frame -> float64 radix-2 DIT FFT -> fftshift -> PSD (float32), plus the
// "tap": the PSD values of the bins the band's listeners sit on, gathered into a compact [frame][listener] array.
// The dB projection of dsp/fft.go:79-81 is a pure function of the float32 PSD value, so it is evaluated where it
// is consumed (k_peaks.hip cumulation, k_listen.hip envelope), not here: this kernel stores 4 bytes per sample
// instead of 8 and carries no logarithm.  Compiled with -ffp-contract=off (see gomath.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "fft_f64.h"
#include "fft_r32.h"
#include "gomath.h"
#include "sdr_device.h"

#if !defined(SDR_FFT_PSD_AUX)
#define SDR_FFT_PSD_AUX 0  // cache policy bits of the psd stores (2 = nt)
#endif
#if !defined(SDR_FFT_DMA_AUX)
#define SDR_FFT_DMA_AUX 2  // cache policy bits of the input LDS-DMA: nt - a frame is read once, by one CU (0.198 vs 0.202 ms)
#endif

namespace sdr {

// Development aids (tools/fft_bench.hip; none of this is compiled into the library).
//  SDR_FFT_PHASES=<workgroup>: every wave of that one workgroup reads the shader clock (s_memtime) at each phase
//    boundary into SGPRs - no wait, no store until the wave's last instruction - and writes the stamps out at the end;
//    the other workgroups pay a scalar compare per stamp.  (Round 2 stamped through memory at every boundary, which
//    cost a third of the kernel's speed; this costs under 1 %.)
//  SDR_FFT_CLOCK: per-workgroup spans (first wave's start, last wave's end, where it ran) + the in-kernel clock.
//  SDR_ABLATE=n: timing-only builds with one ingredient removed (results are wrong by construction).
#if defined(SDR_FFT_CLOCK_LIB)
// diagnostic LIBRARY build (tools/build_abl.sh fftclk "-DSDR_FFT_CLOCK -DSDR_FFT_CLOCK_LIB", tools/insitu_fft.py):
// the per-workgroup spans of the FFT launches inside the running pipeline
__device__ unsigned long long g_fft_clock[2];
__device__ unsigned long long g_fft_wg[2048][4];
extern "C" __attribute__((visibility("default"))) int sdr_debug_fft_wg(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fft_wg), sizeof(g_fft_wg));
}
#endif
// "Everything but the n youngest vector memory operations has completed": relies on loads, stores and LDS-DMA retiring
// in issue order (MI355X_MICROARCH.md; multi-frame workgroups only).  -DSDR_SAFE_FENCES waits for all of them instead.
#if defined(SDR_SAFE_FENCES)
#define SDR_WAIT_ALL_BUT(n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define SDR_WAIT_ALL_BUT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#endif
constexpr int kStampCount = 16;
enum StampId { ST_START = 0, ST_LOADED = 1, ST_PASS0 = 2, ST_EX0 = 3, ST_PASS1 = 4, ST_EX1 = 5, ST_PASS2 = 6, ST_EX2 = 7,
               ST_PASS3 = 8, ST_STORED = 10, ST_END = 11, ST_LANDED = 12, ST_ALL_LANDED = 13 };
#if defined(SDR_FFT_PHASES)
struct Stamps {
    unsigned long long v[kStampCount];
    bool on;
};
#define SDR_STAMP(st, k)                                \
    do {                                                \
        if ((st).on)                                    \
            (st).v[k] = __builtin_amdgcn_s_memtime();   \
    } while (0)
#else
struct Stamps {
};
#define SDR_STAMP(st, k) \
    do {                 \
    } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// k_fft_psd  (dsp/fft.go:23-37 IQToSpectrumAndPSD, the psd half; :59-69 setSamplesFromIQ; :54-57 fftshift)
// ---------------------------------------------------------------------------------------------
// Buffer addressing: address = descriptor base + per-thread 32-bit byte offset (a VGPR) + a scalar byte
// offset.  Everything that is the same for all threads - which register slot, which twiddle row - goes into
// the scalar offset, so a load or store costs no vector ALU instruction for its address.  The float64 VALU
// is this kernel's busiest unit; flat 64-bit addressing spent about 150 vector instructions per thread on
// address arithmetic.
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

// Orders one wave's LDS stores before its later LDS loads (and the reverse) without a workgroup
// barrier: a wave's DS instructions execute in issue order, so the fences only pin the compiler.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// does any exchange after E go through LDS?
template <int LOGN>
constexpr bool later_lds_exchange(int e)
{
    for (int k = e + 1; k < fft64::Plan<LOGN>::NPASS - 1; k++)
        if (fft64::make_reg_plan<LOGN>(k).n == 0)
            return true;
    return false;
}

// `lds_free()` is called once, when the frame's last exchange through LDS is over (a following frame's
// input may be staged into the exchange area from then on).
// is the frame's last exchange through LDS the barrier-fenced cross-wave one?
template <int LOGN>
constexpr bool last_lds_exchange_is_cross()
{
    int last = -1;
    for (int e = 0; e < fft64::Plan<LOGN>::NPASS - 1; e++)
        if (fft64::make_reg_plan<LOGN>(e).n == 0)
            last = e;
    return last >= 0 && fft64::Plan<LOGN>::cross_wave(last);
}

// Twiddles of a pass's first stages are requested BEFORE the exchange in front of the pass (SDR_FFT_TWPRE rows of the
// pass's twiddle block: 1 = stage 0, 3 = stages 0-1, 7 = stages 0-2): behind an exchange every wave of the
// workgroup starts its pass at the same moment, and without this each of them sat out an L2 round trip there - the
// fences of the exchange keep the compiler from hoisting the loads itself.  During the exchange only the 64 data
// registers are live, so the rows cost no register the pass does not have anyway.
#if !defined(SDR_FFT_TWPRE)
#define SDR_FFT_TWPRE 1
#endif
#if !defined(SDR_FFT_TW_CHUNK)
#define SDR_FFT_TW_CHUNK 0  // layout B: at most this many twiddle rows requested at a time (0: the compiler decides)
#endif
#if !defined(SDR_FFT_SCALAR_CHUNK)
#define SDR_FFT_SCALAR_CHUNK 4
#endif
#if !defined(SDR_FFT_TWPRE_B)
#define SDR_FFT_TWPRE_B 7  // layout B: rows of pass 2 requested before the next frame's LDS-DMA goes out (see k_fft_psd_b)
#endif
template <int LOGN, int P>
constexpr int tw_pre_rows()
{
    using PL = fft64::Plan<LOGN>;
    // (a short last pass has several groups per thread with different twiddles each, and nothing but register swaps
    // or nothing at all in front of it: left to the compiler)
    if (P <= 0 || P >= PL::NPASS || PL::pass_log(P) != PL::LOGR)
        return 0;
    if (PL::LB && P == 1)
        return 0;  // layout B: pass 1's twiddles depend on the wave only - scalar loads, nothing to pre-issue
    if (PL::LB && P == 2)
        return SDR_FFT_TWPRE_B < 15 ? SDR_FFT_TWPRE_B : 15;
    constexpr int rows = (1 << PL::LOGR) - 1;
    return SDR_FFT_TWPRE < rows ? SDR_FFT_TWPRE : rows;
}
constexpr int kTwPreMax = 15;

__device__ __forceinline__ fft64::cplx load_tw(rsrc_t tw, int lo, int c)
{
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(tw, (unsigned)lo * 16u, c * 16, 0);
    fft64::cplx r;
    r.x = __hiloint2double((int)w.y, (int)w.x);
    r.y = __hiloint2double((int)w.w, (int)w.z);
    return r;
}

template <int LOGN, int P>
__device__ __forceinline__ void prefetch_tw(fft64::cplx (&pre)[kTwPreMax], int t, rsrc_t tw)
{
    using PL = fft64::Plan<LOGN>;
    constexpr int NPRE = tw_pre_rows<LOGN, P>();
    if constexpr (NPRE > 0) {
        constexpr int S = 1 << (P * PL::LOGR);
        const int lo = fft64::tw_pos<LOGN, P>(t, 0);
#pragma unroll
        for (int r = 0; r < NPRE; r++)
            pre[r] = load_tw(tw, lo, PL::tw_offset(P) + r * S);
    }
}

// `hooks(E, after)` is called right before (after == false) and right after (after == true) exchange E: the
// register-staged kernel hangs its tap and its store fence there.
struct NoHooks {
    __device__ __forceinline__ void operator()(int, bool) const {}
};

template <int LOGN, int P, bool FRAME_FOLLOWS, class LdsFree, class Hooks = NoHooks>
__device__ __forceinline__ void run_passes(double (&xr)[fft64::Plan<LOGN>::R], double (&xi)[fft64::Plan<LOGN>::R],
                                           int t, rsrc_t tw, const fft64::cplx *__restrict__ tw_ptr, double *lds,
                                           LdsFree lds_free, const fft64::cplx (&pre)[kTwPreMax], Stamps &st,
                                           Hooks hooks = Hooks{})
{
    using PL = fft64::Plan<LOGN>;
#if !(defined(SDR_ABLATE) && (SDR_ABLATE == 5))
    // (layout B, pass 1: the twiddle depends on the wave only - wave w owns the sub-problem "index bits 0-3 = w")
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    // (layout B keeps the next frame's samples in 32 registers: twiddle rows come four at a time, fft_f64.h)
    // (scalar passes: a fence behind every few rows keeps the scalar loads from all being hoisted to the top of the
    // pass, where their 60 SGPRs do not fit)
    constexpr int CHUNK = !PL::LB ? 0 : (P <= 1 ? SDR_FFT_SCALAR_CHUNK : SDR_FFT_TW_CHUNK);
    fft64::butterfly_pass<LOGN, P, CHUNK>(xr, xi, t, [tw, tw_ptr, &pre, wave_u](int c, int lo) {
        if constexpr (P == 0)
            return tw_ptr[c];  // pass 0: the same entry for every thread, a scalar load
#if !defined(SDR_FFT_P1_VECTOR)
        if constexpr (PL::LB && P == 1)
            return tw_ptr[c + wave_u];  // scalar load (fft_f64.h tw_pos(t, 0) == wave here, checked by tests/emu)
#endif
        constexpr int S = 1 << (P * PL::LOGR);
        const int row = (c - PL::tw_offset(P)) / S;  // (a constant once the pass is unrolled)
        if (row < tw_pre_rows<LOGN, P>())
            return pre[row];
        return load_tw(tw, lo, c);
    });
#endif
    // Layout B: pass 2's rows are requested two passes ahead - behind exchange 0, BEFORE lds_free() sends the next
    // frame's LDS-DMA out (a vector load issued behind the DMA returns behind it, 5 us later) - and ride through pass 1,
    // whose own twiddles are scalar: they arrive in `pre` at pass 1 and are handed on.
    fft64::cplx pre_next[kTwPreMax];
    if constexpr (P < PL::NPASS - 1 && !(PL::LB && P <= 1))
        prefetch_tw<LOGN, P + 1>(pre_next, t, tw);
    if constexpr (PL::LB && P == 1) {
#pragma unroll
        for (int r = 0; r < kTwPreMax; r++)
            pre_next[r] = pre[r];
    }
    SDR_STAMP(st, 2 + 2 * P);
    if constexpr (P < PL::NPASS - 1) {
        hooks(P, false);
        if constexpr (fft64::make_reg_plan<LOGN>(P).n > 0) {
            // done in registers (fft_f64.h exchange_regs: v_permlane16/32_swap, ds_bpermute), no LDS memory
#if !(defined(SDR_ABLATE) && (SDR_ABLATE == 14))
            fft64::exchange_regs<LOGN, P>(xr);
            fft64::exchange_regs<LOGN, P>(xi);
#endif
        } else {
            // A wave-local exchange (fft_f64.h make_layout) only touches LDS words of the wave's own
            // elements: no workgroup barrier, the waves drift apart and one wave's exchange overlaps the
            // others' butterflies.  The single cross-wave exchange is fenced by barriers on both sides.
            constexpr bool CROSS = PL::cross_wave(P);
            // timing-only builds: 10 = cross-wave exchange without its LDS traffic, 11 = without its barriers,
            // 12 = without either, 13 = no wave-local LDS exchange
#if defined(SDR_ABLATE) && (SDR_ABLATE == 11 || SDR_ABLATE == 12)
            constexpr bool BARRIERS = false;
#else
            constexpr bool BARRIERS = CROSS;
#endif
#if defined(SDR_ABLATE) && (SDR_ABLATE == 10 || SDR_ABLATE == 12)
            constexpr bool TRAFFIC = !CROSS;
#elif defined(SDR_ABLATE) && (SDR_ABLATE == 13)
            constexpr bool TRAFFIC = CROSS;
#else
            constexpr bool TRAFFIC = true;
#endif
            auto sync = [] {
                if constexpr (BARRIERS)
                    __syncthreads();
                else
                    wave_sync();
            };
            auto wr = [&](double (&x)[PL::R], double *area) {
                if constexpr (TRAFFIC)
                    fft64::exchange_write<LOGN, P>(x, t, area);
            };
            auto rd = [&](double (&x)[PL::R], double *area) {
                if constexpr (TRAFFIC)
                    fft64::exchange_read<LOGN, P>(x, t, area);
            };
            if constexpr (BARRIERS)
                __syncthreads();  // every wave is done with the words of its previous wave-local exchange
#if defined(SDR_ABLATE) && (SDR_ABLATE == 2)
            if (t < 0)  // timing-only build: no exchanges
#endif
            if constexpr (PL::SPLIT) {
                wr(xr, lds);
                sync();
                rd(xr, lds);
                sync();
                wr(xi, lds);
                sync();
                rd(xi, lds);
            } else {
                wr(xr, lds);
                wr(xi, lds + fft64::kExchangeWords<LOGN>);
                sync();
                rd(xr, lds);
                rd(xi, lds + fft64::kExchangeWords<LOGN>);
            }
            // reads done before a later exchange writes LDS again (other waves' words if CROSS)
            // ... or the next frame's staging does
            if constexpr (later_lds_exchange<LOGN>(P) || FRAME_FOLLOWS)
                sync();
            if constexpr (PL::LB && P == 0)
                prefetch_tw<LOGN, 2>(pre_next, t, tw);
            if constexpr (!later_lds_exchange<LOGN>(P))
                lds_free();
        }
        SDR_STAMP(st, 3 + 2 * P);
        hooks(P, true);
        run_passes<LOGN, P + 1, FRAME_FOLLOWS>(xr, xi, t, tw, tw_ptr, lds, lds_free, pre_next, st, hooks);
    }
}

// Epilogue (dsp/fft.go:54-57 fftshift, :71-73 PSD[float32]): psd[k] = float32(re^2 + im^2), two multiplies and an
// add in float64, no FMA, rounded once.
// LDS_COPY: the row also goes to LDS (float32 at byte 4 k), where the one-frame workgroup's tap picks its bins up.
template <int LOGN, bool LDS_COPY>
__device__ __forceinline__ void store_psd(const double (&xr)[fft64::Plan<LOGN>::R], const double (&xi)[fft64::Plan<LOGN>::R],
                                          int t, float *__restrict__ pd, unsigned char *lds_row, bool lds_copy)
{
    using PL = fft64::Plan<LOGN>;
    const int tp = fft64::thread_part<LOGN, PL::NPASS - 1>(t);
    // fft-shift = flip the top index bit: in the slot part it is a compile-time constant, in the thread part
    // it is applied once; spectrum index k = tk | sk(s), and sk(s) goes into the scalar offset
    constexpr int SLOT_MASK = fft64::slot_part<LOGN, PL::NPASS - 1>(PL::R - 1);
    constexpr int H = PL::N / 2;
    const unsigned tk = (unsigned)(tp ^ (H & ~SLOT_MASK));
    const rsrc_t pdr = make_rsrc(pd, PL::N * 4u);
#pragma unroll
    for (int s = 0; s < PL::R; s++) {
        const int sk = fft64::slot_part<LOGN, PL::NPASS - 1>(s) ^ (H & SLOT_MASK);
        const float p = (float)(xr[s] * xr[s] + xi[s] * xi[s]);
#if defined(SDR_ABLATE) && (SDR_ABLATE == 6 || SDR_ABLATE == 7 || SDR_ABLATE == 16)
        if (p == 1234.5f)  // timing-only build: (almost) no stores
            pd[tk | sk] = p;
#else
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(p), pdr, tk * 4u, sk * 4, SDR_FFT_PSD_AUX);
#endif
        if constexpr (LDS_COPY)
            if (lds_copy)  // (uniform) per-thread address once, the slot part in the instruction's offset field
                *reinterpret_cast<float *>(lds_row + tk * 4u + (unsigned)(sk * 4)) = p;
    }
}

// Where the time of a frame goes (N = 16384, tools/fft_bench.hip with -DSDR_FFT_CLOCK and the -DSDR_ABLATE
// builds, MI355X at 2.3 GHz in-kernel): the phases of a frame run one after the other on its CU - all 16 waves
// wait for the input, then all compute, then all exchange, ... - and each phase is bound by a different unit, so
// their times ADD: nothing of another frame can run beside them, a frame's float64 state is half the CU's
// register file.  Hence:
//  * MULTI: a workgroup takes `fpw` consecutive frames and has the next frame's LDS-DMA in flight while it
//    finishes the current one (SDR_FFT_DMA_AT: 0 = issued when the last exchange through LDS is over, 1 = just
//    before the epilogue).  The wait at the top of the next frame is a COUNTED vmcnt: the DMA is older than the
//    R psd stores that followed it, and vector-memory operations retire in order, so "all but the R youngest"
//    covers the DMA without draining the stores.
//  * no logarithm here (see the file header) - it was a fifth of the kernel.
// Tried and measured no better: starting the first generation of workgroups staggered over a frame time (the
// theory was that 256 CUs reading at the same moment and storing at the same moment make HBM bursts; spreading
// them changed nothing), a persistent variant prefetching into the registers the epilogue frees (5 % slower).
//
// The tap.  `tap_bins[band][tap_stride]` lists the spectrum bins the band's listeners sit on (-1: free slot);
// for each of them the frame's psd value goes to tap_out[band][frame][slot].  The value is re-read from the psd
// row the workgroup has stored (thread l takes slot l), once those stores have certainly reached L2.  A MULTI
// workgroup taps frame f-1 near the end of frame f: by then every wave has waited for twiddles it loaded during
// frame f - younger than its stores of frame f-1, and vector-memory operations retire in order - and has passed
// the barriers of the cross-wave exchange since, so all of frame f-1's stores are complete; the tap then costs
// two instructions per listener and no drain; its last frame is tapped after a final drain.  A one-frame workgroup
// (the default) would pay that drain - a microsecond of store latency with the whole CU held - on every frame, so
// its epilogue also writes the psd row into LDS (the exchange area is free by then; LDS stores cost no vector ALU
// time) and the tap reads its bins from there behind one barrier: no wait on memory at all.
#if !defined(SDR_FFT_DMA_AT)
#define SDR_FFT_DMA_AT 1
#endif
// Wave-private staging (SDR_FFT_PRIVATE_STAGE = 1: by LDS-DMA, 2: straight into registers; default 0 = off).  The
// frame's 128 KB arrive over about 5 us (every CU of a generation asks at once: 6.4 TB/s while it lasts) and the waves'
// rows land in the order the waves were started, four at a time - one per SIMD.  Staged co-operatively (every wave
// reads from every 1 KB row: two workgroup barriers before the first butterfly) the whole workgroup sits that time
// out.  The idea: each wave fetches exactly the samples its own lanes hold in pass 0 - 32-byte runs, the layout's wave
// bits are sample bits 2-5 - into the 8 KB of LDS that are its own in the wave-local exchange that follows (or into
// the registers pass 0 starts from), and starts as soon as ITS samples are there (its own vmcnt, no barrier): the
// waves that land first run passes 0 and 1 while the others' samples are on their way.
// MEASURED (round 4, tools/fft_bench, 2048 x 16384, profiles/r04_fft_experiments.txt): bit-identical and SLOWER - 0.170
// (LDS-DMA) / 0.171 ms (registers; 0.181 with the nt policy) against 0.163 co-operative.  The pipelining is there (the
// first waves have their samples after 5 200 clocks and are through pass 1 at 18 500, where the co-operative build
// starts pass 0 at 13 100) but the last waves' samples land at 24 700 instead of 12 000: four waves share every 128-byte
// line, each asks for it on its own, and the CU's inbound path delivers lines, not bytes (a stand-alone load loop,
// tools/ubench_load, does not show this: there nothing else competes for the path).  A layout whose waves own whole
// lines in pass 0 needs the cross-wave exchange right behind pass 0 - layout B below, which lost for other reasons.
#if !defined(SDR_FFT_PRIVATE_STAGE)
#define SDR_FFT_PRIVATE_STAGE 0
#endif
// logical id of the thread (which index bits its wave id stands for is fft_f64.h make_layout's business; which of the
// workgroup's waves plays which logical wave is free)
template <int LOGN>
__device__ __forceinline__ int logical_thread(int tid)
{
    using PL = fft64::Plan<LOGN>;
    if constexpr (SDR_FFT_PRIVATE_STAGE && PL::WB == 4 && !PL::LB) {
        const int w = tid >> 6;
        return (((w & 3) << 2 | (w >> 2)) << 6) | (tid & 63);
    } else {
        return tid;
    }
}
// bytes of LDS a wave owns across the staging and the first (wave-local) exchange
template <int LOGN>
inline constexpr int kWaveBlockBytes = fft64::make_addr<LOGN>(0).block * 8;
template <int LOGN>
constexpr bool private_stage()
{
    using PL = fft64::Plan<LOGN>;
    // the first exchange must be wave-local (layout A always) and the wave's block must hold its R x 64 samples
    return SDR_FFT_PRIVATE_STAGE && !PL::LB && !(PL::WB > 0 && PL::cross_wave(0)) && kWaveBlockBytes<LOGN> >= PL::R * 512;
}
constexpr int kMaxLdsTap = 4096;  // listeners per band the LDS tap holds bins for (16 KB); more fall back to the drain
// Cache warm-up of a LATER frame (one-frame workgroups; SDR_FFT_PF_DIST > 0, default 0 = off).  The chip runs the
// frames in generations - 256 workgroups start together, fetch their 128 KB together and then leave the memory system
// idle for the rest of the frame; a workgroup with 64 CUs running lives 14.2 us, one of 256 18.4 us (tools/fft_bench 64 /
// 2048 frames, -DSDR_FFT_CLOCK).  The idea: each workgroup touches every 128-byte line of the frame SDR_FFT_PF_DIST
// workgroups ahead - the one that follows it on its CU, same XCD - with ONE LDS-DMA dword per lane into a sink behind
// the tap bins (no registers, nothing waits for it), right before exchange SDR_FFT_PF_AT, so that the lines sit in L2 /
// the Infinity Cache when that workgroup's staging DMA asks for them.
// MEASURED (round 4, profiles/r04_fft_experiments.txt): no gain before exchange 0 (0.162 against 0.162 ms), 0.205 -
// 0.214 ms before exchanges 1 / 2, any distance, nt or not: the warm-up is itself a 32 MB burst of the whole chip, the
// twiddle loads behind it return behind it, and what it leaves in the Infinity Cache comes back no faster than from HBM
// through the same fabric (L2 cannot hold a generation: 32 CUs x 128 KB = its 4 MB); issued at the very end of the
// workgroup (SDR_FFT_PF_AT=3: nothing of it waits behind the warm-up) 0.171 - 0.176 ms - the workgroup cannot leave before
// its LDS-DMA has landed.  The burst is the cost, and only
// spreading the requests over the frame - which needs somewhere on the CU to put them - would remove it.
#if !defined(SDR_FFT_PF_DIST)
#define SDR_FFT_PF_DIST 0
#endif
#if !defined(SDR_FFT_PF_AT)
#define SDR_FFT_PF_AT 1
#endif
#if !defined(SDR_FFT_PF_AUX)
#define SDR_FFT_PF_AUX 0
#endif
constexpr int kPfSinkBytes = SDR_FFT_PF_DIST > 0 ? 4096 : 0;
// (second launch bound = waves per SIMD the register allocation must leave room for: four, i.e. one 1024-thread
// workgroup or two 512-thread ones per CU)
template <int LOGN, bool MULTI>
__global__ __launch_bounds__(fft64::Plan<LOGN>::T, (fft64::Plan<LOGN>::T >= 512 ? 4 : 1)) void k_fft_psd(const float *__restrict__ iq_arg, const BatchCursor *__restrict__ cur,
                                                                  const fft64::cplx *__restrict__ tw,
                                                                  float *__restrict__ psd, int in_stride, int out_stride,
                                                                  int n_frames, int fpw, const int *__restrict__ tap_bins,
                                                                  float *__restrict__ tap_out, int n_tap, int tap_stride)
{
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass needs the signature only; with the body it drops the stub without a diagnostic)
    using PL = fft64::Plan<LOGN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *lds = reinterpret_cast<double *>(smem);
    Stamps st;
#if defined(SDR_FFT_PHASES)
    st.on = blockIdx.x == SDR_FFT_PHASES && blockIdx.y == 0;
#pragma unroll
    for (int k = 0; k < kStampCount; k++)
        st.v[k] = 0;
#endif
#if defined(SDR_FFT_CLOCK)
    unsigned long long ck0 = 0, rt0 = 0;
    if (blockIdx.x == 100)
        ck0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    SDR_STAMP(st, ST_START);
    const float *__restrict__ iq = cur ? cur->iq : iq_arg;  // graph replay: the batch's input pointer lives in device memory
    const int frame0 = MULTI ? blockIdx.x * fpw : blockIdx.x;
    const int frame_end = MULTI ? min(frame0 + fpw, n_frames) : frame0 + 1;
    const size_t in_band = (size_t)blockIdx.y * in_stride, out_band = (size_t)blockIdx.y * out_stride;
    constexpr bool PRIV = private_stage<LOGN>();
    const int ltid = logical_thread<LOGN>((int)threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(ltid >> 6);

    // Frame -> LDS by LDS-DMA.  Co-operative: one contiguous 1 KB row per wave instruction, shaped through the source
    // address (fft_f64.h "Input staging"), the image lives in the exchange area.  Wave-private (see above): instruction j
    // fills the wave's slots 2j (lanes 0-31) and 2j+1 (lanes 32-63), 16 bytes = two consecutive samples = two
    // neighbouring lanes' values per DMA lane; block layout [slot][lane], 8 bytes each.
    auto stage_frame = [&](int frame, int tid, int part = 0) {
#if defined(SDR_ABLATE) && (SDR_ABLATE == 15 || SDR_ABLATE == 16)
        if (frame >= 0)  // timing-only build: no input DMA at all - what a perfectly hidden input would leave
            return;
#endif
        constexpr int ROWS_PER_WAVE = PL::R / 2;
        const int lane = tid & 63;
#if defined(SDR_ABLATE) && (SDR_ABLATE == 8)
        const size_t fr = (in_band + frame) & 15;  // timing-only: 16 frames, L2-resident
#else
        const size_t fr = in_band + frame;
#endif
        // buffer form: row in the scalar offset, granule in one 32-bit VGPR - no 64-bit per-lane addresses
        const rsrc_t xrs = make_rsrc(iq + fr * PL::N * 2, PL::N * 8u);
        if constexpr (PRIV) {
            const unsigned voff = (unsigned)(fft64::input_sample<LOGN>((tid & ~63) | (2 * (lane & 31)), 0) +
                                             ((lane >> 5) ? fft64::input_slot_sample<LOGN>(1) : 0)) * 8u;
#pragma unroll
            for (int j = 0; j < ROWS_PER_WAVE; j++) {
                if ((part == 1 && j >= ROWS_PER_WAVE / 2) || (part == 2 && j < ROWS_PER_WAVE / 2))
                    continue;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(smem + wave * kWaveBlockBytes<LOGN> + j * 1024), 16,
                                                         voff, fft64::input_slot_sample<LOGN>(2 * j) * 8, 0, SDR_FFT_DMA_AUX);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < ROWS_PER_WAVE; j++) {
            if ((part == 1 && j >= ROWS_PER_WAVE / 2) || (part == 2 && j < ROWS_PER_WAVE / 2))
                continue;
            const int r = wave * ROWS_PER_WAVE + j;
            const int g = fft64::in_granule<LOGN>(lane, r);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(smem + r * 1024), 16,
                                                     (unsigned)g * 16u, r * 1024, 0, SDR_FFT_DMA_AUX);
        }
    };
    // tap of one finished frame (its psd stores are known to be complete, see above)
    auto tap_frame = [&](int frame) {
        const float *row = psd + (out_band + frame) * PL::N;
        float *out = tap_out + (out_band + frame) * (size_t)tap_stride;
        const int *bins = tap_bins + (size_t)blockIdx.y * tap_stride;
        for (int l = threadIdx.x; l < n_tap; l += PL::T) {
            const int bin = bins[l];
            out[l] = bin >= 0 ? row[bin] : 0.0f;
        }
    };
    constexpr bool REGS = PRIV && !MULTI && SDR_FFT_PRIVATE_STAGE == 2;  // one-frame workgroup: straight into the registers pass 0 starts from
    if constexpr (!REGS)
        stage_frame(frame0, ltid);
    // one-frame workgroup: its listeners' bins into LDS (behind the exchange area) while the frame is on its way
    int *lds_bins = reinterpret_cast<int *>(smem + fft64::kLdsBytes<LOGN>);
    const bool lds_tap = !MULTI && n_tap > 0 && n_tap <= kMaxLdsTap;
    unsigned char *pf_sink = smem + fft64::kLdsBytes<LOGN> + (lds_tap ? ((n_tap * 4 + 255) & ~255) : 0);
    (void)pf_sink;
    if (lds_tap)
        for (int l = threadIdx.x; l < n_tap; l += PL::T)
            lds_bins[l] = tap_bins[(size_t)blockIdx.y * tap_stride + l];

#pragma nounroll
    for (int frame = frame0; frame < frame_end; frame++) {
        // (with more than one frame per workgroup everything derived from the thread id is loop-invariant and the
        // compiler would hoist - and spill - it: make the thread id opaque per frame)
        int t = ltid;
        if constexpr (MULTI)
            asm volatile("" : "+v"(t));
        // The first frame's DMA is the wave's only traffic: full drain.  A later frame's DMA is older than the
        // previous frame's R psd stores (MI355X_MICROARCH.md: loads, stores and LDS-DMA count together, in issue
        // order), so "all but the R youngest" covers it (tap traffic behind the stores only makes the wait cover
        // some of the stores too).
        double xr[PL::R], xi[PL::R];
        if constexpr (REGS) {
            // dsp/fft.go:59-69 setSamplesFromIQ: slot m <- sample input_sample(t, m): the thread's part of the sample
            // number in the per-lane offset, the slot's in the scalar offset; 8 bytes per lane, four lanes per 32-byte run
            const rsrc_t xrs = make_rsrc(iq + (in_band + frame) * PL::N * 2, PL::N * 8u);
            const unsigned voff = (unsigned)fft64::input_sample<LOGN>(t, 0) * 8u;
            u32x2 raw[PL::R];
#pragma unroll
            for (int m = 0; m < PL::R; m++)
                raw[m] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, fft64::input_slot_sample<LOGN>(m) * 8, SDR_FFT_DMA_AUX);
#pragma unroll
            for (int m = 0; m < PL::R; m++) {
                xr[m] = (double)__uint_as_float(raw[m].x);
                xi[m] = (double)__uint_as_float(raw[m].y);
            }
            SDR_STAMP(st, ST_LANDED);
        } else {
        if (!MULTI || frame == frame0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            SDR_WAIT_ALL_BUT(PL::R);
        SDR_STAMP(st, ST_LANDED);  // this wave's rows have landed
        if constexpr (!PRIV)
            __syncthreads();
        SDR_STAMP(st, ST_ALL_LANDED);  // everybody's have

        const int n_thread = fft64::input_sample<LOGN>(t, 0);
        const int thread_byte = PRIV ? wave * kWaveBlockBytes<LOGN> + (t & 63) * 8 : fft64::in_lds_byte<LOGN>(n_thread);
#pragma unroll
        for (int m = 0; m < PL::R; m++) {
#if defined(SDR_ABLATE) && (SDR_ABLATE == 1 || SDR_ABLATE == 7)
            const float2 v = make_float2(1e-3f * (float)(t + m), 0.5f);  // timing-only build: no input
#else
            // co-operative image: sample number -> image address is linear over GF(2): thread part and slot part
            // combine by XOR, and the slot part is a compile-time constant; wave-private block: [slot][lane]
            const int slot_byte = PRIV ? m * 512 : fft64::in_lds_byte<LOGN>(fft64::input_sample<LOGN>(0, m));
            const float2 v = *reinterpret_cast<const float2 *>(smem + (PRIV ? thread_byte + slot_byte : (thread_byte ^ slot_byte)));
#endif
            xr[m] = (double)v.x;
            xi[m] = (double)v.y;
        }
        if constexpr (PRIV)
            wave_sync();  // the wave has its samples: its block belongs to its first exchange now
        else
            __syncthreads();  // everyone has its samples: the exchange area may be written again
        }
        SDR_STAMP(st, ST_LOADED);
        const bool more = MULTI && frame + 1 < frame_end;
        // (no scheduling pin around the DMA: the compiler keeps it behind the exchanges' LDS accesses and behind the
        // twiddle loads already issued, which it waits for with counted vmcnt; a "memory" pin cost 46 spills)
        // (LDS is written again after the last exchange in both variants - the next frame's staging or the tap's
        // copy of the psd row - so the exchange ends with its fence: a barrier when it crossed waves)
        const fft64::cplx no_pre[kTwPreMax] = {};
        run_passes<LOGN, 0, true>(xr, xi, t, make_rsrc(tw, (unsigned)(PL::TW_TOTAL * sizeof(fft64::cplx))), tw, lds, [&] {
            if constexpr (MULTI && (SDR_FFT_DMA_AT == 0 || SDR_FFT_DMA_AT == 2))
                if (more)
                    stage_frame(frame + 1, t, SDR_FFT_DMA_AT == 2 ? 1 : 0);
        }, no_pre, st, [&](int e, bool after) {
            if constexpr (!MULTI && SDR_FFT_PF_DIST > 0) {
                if (e == SDR_FFT_PF_AT && !after && frame + SDR_FFT_PF_DIST < n_frames) {
                    const rsrc_t nrs = make_rsrc(iq + (in_band + frame + SDR_FFT_PF_DIST) * PL::N * 2, PL::N * 8u);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(nrs, (__attribute__((address_space(3))) void *)(pf_sink + wave * 256), 4,
                                                             (unsigned)(t & 63) * 128u, wave * 8192, 0, SDR_FFT_PF_AUX);
                }
            }
        });
        if constexpr (MULTI && (SDR_FFT_DMA_AT == 1 || SDR_FFT_DMA_AT == 2))
            if (more)
                stage_frame(frame + 1, t, SDR_FFT_DMA_AT == 2 ? 2 : 0);
        if constexpr (!MULTI && !last_lds_exchange_is_cross<LOGN>())
            if (lds_tap)
                __syncthreads();  // a wave-local last exchange fences only its own wave; the row goes everywhere
        store_psd<LOGN, !MULTI>(xr, xi, t, psd + (out_band + frame) * PL::N, smem, lds_tap);
        if constexpr (!MULTI && SDR_FFT_PF_DIST > 0 && SDR_FFT_PF_AT == 3) {
            // (behind the frame's last vector loads: nothing of this workgroup waits behind the warm-up any more)
            if (frame + SDR_FFT_PF_DIST < n_frames) {
                const rsrc_t nrs = make_rsrc(iq + (in_band + frame + SDR_FFT_PF_DIST) * PL::N * 2, PL::N * 8u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(nrs, (__attribute__((address_space(3))) void *)(pf_sink + wave * 256), 4,
                                                         (unsigned)(t & 63) * 128u, wave * 8192, 0, SDR_FFT_PF_AUX);
            }
        }
        // (behind the DMA and the stores, so that its two dependent loads delay neither: the oldest waves - the
        // ones that tap - reach the end of a frame microseconds before the youngest)
        if constexpr (MULTI) {
            static_assert(PL::NPASS >= 2, "the tap relies on pass-1 twiddle loads");
            if (n_tap > 0 && frame > frame0)
                tap_frame(frame - 1);
        }
        SDR_STAMP(st, ST_STORED);
    }
    if (lds_tap) {
        __syncthreads();  // the row is in LDS (and lds_bins has been for a long time)
        const float *row = reinterpret_cast<const float *>(smem);
        float *out = tap_out + (out_band + frame0) * (size_t)tap_stride;
        for (int l = threadIdx.x; l < n_tap; l += PL::T) {
            const int bin = lds_bins[l];
            out[l] = bin >= 0 ? row[bin] : 0.0f;
        }
    } else if (n_tap > 0 && frame_end > frame0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tap_frame(frame_end - 1);
    }
    SDR_STAMP(st, ST_END);
#if defined(SDR_FFT_PHASES)
    if (st.on && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kStampCount; k++)
            g_fft_phases[threadIdx.x >> 6][k] = st.v[k];
    }
#endif
#if defined(SDR_FFT_CLOCK)
    if (blockIdx.x == 100 && threadIdx.x == 0) {
        g_fft_clock[0] = __builtin_amdgcn_s_memtime() - ck0;
        g_fft_clock[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    // every workgroup: first wave's start, each wave's end (the host takes the latest), where it ran
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            g_fft_wg[blockIdx.x][0] = rt0;
            g_fft_wg[blockIdx.x][1] = now;
            // HW_REG_HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13 (gfx9); XCC_ID is a register of its own
            g_fft_wg[blockIdx.x][3] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)) |
                                      ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11)) << 32);
        }
        atomicMax(&g_fft_wg[blockIdx.x][2], now);
    }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

// Epilogue of layout B (dsp/fft.go:54-57 fftshift, :71-73 PSD[float32]).  Wave w holds the bins = w (mod 16): stored
// from the registers, a wave instruction would put 64 four-byte words into 64 different 64-byte segments of the row
// (measured: 0.087 of the kernel's 0.21 ms).  So the row goes through a 32 KB tile of LDS - the part the next frame's
// staging image leaves free - half a row at a time: every thread drops its eight values of the half (ds_write_b32, the
// slot part of the address in the offset field; an XOR swizzle of address bits 0-3 with bits 5-8 keeps both sides
// free of bank conflicts), and behind a barrier picks up two runs of four consecutive bins (ds_read_b128; the swizzle
// permutes the four words by a per-thread constant, undone with v_cndmask) and stores them 16 bytes per lane, 1 KB
// contiguous per wave instruction.
template <int LOGN>
__device__ __forceinline__ void store_psd_b(const double (&xr)[fft64::Plan<LOGN>::R], const double (&xi)[fft64::Plan<LOGN>::R], int t,
                                            float *__restrict__ pd, unsigned char *tile)
{
    using PL = fft64::Plan<LOGN>;
    constexpr int H = PL::N / 2;
    constexpr int LAST = PL::NPASS - 1;
    static_assert(PL::T * 8 == PL::N / 2 && PL::N * 2 == 32768, "tile = half a row = 32 KB, eight bins per thread");
    const int tp = fft64::thread_part<LOGN, LAST>(t);  // (bits 0..9 here: wave and lanes)
    const unsigned x = ((unsigned)tp >> 5) & 15u;
    const unsigned wbase = (((unsigned)tp & (unsigned)(H - 1)) ^ x) * 4u;
    const rsrc_t pdr = make_rsrc(pd, PL::N * 4u);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h)
            __syncthreads();  // everybody has read the first half out of the tile
#pragma unroll
        for (int s = 0; s < PL::R; s++) {
            const int sp = fft64::slot_part<LOGN, LAST>(s);
            const int k_hi = ((sp ^ H) >> (LOGN - 1)) & 1;  // which half of the shifted row this slot's bin lands in
            if (k_hi != h)
                continue;
            const float p = (float)(xr[s] * xr[s] + xi[s] * xi[s]);
            *reinterpret_cast<float *>(tile + wbase + (unsigned)((sp & (H - 1)) * 4)) = p;
        }
        __syncthreads();
#pragma unroll
        for (int part = 0; part < 2; part++) {
            const unsigned a0 = 4u * (unsigned)t + (unsigned)(part * (H / 2));  // first of this thread's four bins (within the half)
            const unsigned xr4 = (a0 >> 5) & 15u;
            const float4 u = *reinterpret_cast<const float4 *>(tile + (((a0 ^ xr4) & ~3u) * 4u));
            // word i of the read holds bin a0 + (i ^ (xr4 & 3))
            const bool s1 = xr4 & 1u, s2 = xr4 & 2u;
            const float a = s1 ? u.y : u.x, b = s1 ? u.x : u.y, c = s1 ? u.w : u.z, d = s1 ? u.z : u.w;
            u32x4 v;
            v.x = __float_as_uint(s2 ? c : a);
            v.y = __float_as_uint(s2 ? d : b);
            v.z = __float_as_uint(s2 ? a : c);
            v.w = __float_as_uint(s2 ? b : d);
#if defined(SDR_ABLATE) && (SDR_ABLATE == 6 || SDR_ABLATE == 7 || SDR_ABLATE == 16)
            if (v.x == 0x449a5000u)  // timing-only build: (almost) no stores
#endif
            __builtin_amdgcn_raw_buffer_store_b128(v, pdr, a0 * 4u, h * (H * 4), SDR_FFT_PSD_AUX);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_fft_psd_b - layout B (N = 16384).  What it is built around: a CU takes in about 25 GB/s, so a frame's 128 KB need
// 5 us to arrive, a third of the 15 us the frame's arithmetic and exchanges take - and in layout A that third is
// spent waiting (tools/fft_bench, SDR_ABLATE=15: 0.115 ms per 2048 frames with the input wait removed, 0.168 with
// it).  Two things keep layout A from fetching the next frame under the current one: its 128 KB of LDS are busy until
// the last exchange, and a twiddle load issued behind an LDS-DMA returns behind it (vector memory operations retire
// in order), so a DMA sent out in the second half of a frame stalls that half's twiddles.  Here
//   * the cross-wave exchange comes first, right behind pass 0, and everything behind it stays in registers
//     (v_permlane swaps and ds_bpermute lane rotations, fft_f64.h make_reg_plan): LDS is free for the next frame from a
//     third of the way into the current one;
//   * pass 1, which follows, loads its twiddles through the scalar cache (wave w owns the sub-problem "index bits 0-3
//     = w"), and the first eleven rows of pass 2's twiddles are requested BEFORE the DMA goes out and ride through
//     pass 1 in the registers scalar twiddles leave free: the first vector load behind the DMA is issued 5.5 us
//     after it.
// The staging image is wave-private: wave w fetches exactly the 1024 samples it will read (16 slots x 64 consecutive
// samples: 512 contiguous bytes each), so no barrier stands between the DMA and the reads - only the wave's own
// counted vmcnt.  A workgroup takes `fpw` consecutive frames.  Tap: frame f-1's bins are read back from its psd row
// around the register exchange behind pass 1 of frame f (loads before it, stores behind it); the row is complete by
// then: every wave waited for its own psd stores of frame f-1 (vmcnt at the top of frame f covers everything older
// than them; the hook in front of the cross-wave exchange waits for the stores themselves) before the barriers of
// frame f's cross-wave exchange, which lie in between.  The workgroup's last frame is tapped after a drain.
// ---------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(fft64::Plan<LOGN>::T, 4) void k_fft_psd_b(const float *__restrict__ iq_arg, const BatchCursor *__restrict__ cur,
                                                                        const fft64::cplx *__restrict__ tw, float *__restrict__ psd,
                                                                        int in_stride, int out_stride, int n_frames, int fpw,
                                                                        const int *__restrict__ tap_bins, float *__restrict__ tap_out,
                                                                        int n_tap, int tap_stride)
{
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass needs the kernel's signature only; something in this body made it drop the
                                     // stub without a diagnostic)
    using PL = fft64::Plan<LOGN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *lds = reinterpret_cast<double *>(smem);
    Stamps st;
#if defined(SDR_FFT_PHASES)
    st.on = blockIdx.x == SDR_FFT_PHASES && blockIdx.y == 0;
#pragma unroll
    for (int k = 0; k < kStampCount; k++)
        st.v[k] = 0;
#endif
#if defined(SDR_FFT_CLOCK)
    unsigned long long ck0 = 0, rt0 = 0;
    if (blockIdx.x == 100)
        ck0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    SDR_STAMP(st, ST_START);
    const float *__restrict__ iq = cur ? cur->iq : iq_arg;
    const int frame0 = blockIdx.x * fpw;
    const int frame_end = min(frame0 + fpw, n_frames);
    const size_t in_band = (size_t)blockIdx.y * in_stride, out_band = (size_t)blockIdx.y * out_stride;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kBlock = PL::R * 64 * 8;  // bytes of staging per wave: its 16 slots x 64 samples
    constexpr int kStageBytes = PL::N * 8;  // the staging image; the epilogue's 32 KB tile sits behind it
    // Frame -> this wave's staging block by LDS-DMA.  Instruction j fills slots 2j (lanes 0-31) and 2j+1 (lanes
    // 32-63), 16 bytes = two consecutive samples per lane; block layout [slot][lane], 8 bytes each.
    auto stage_frame = [&](int frame) {
#if defined(SDR_ABLATE) && (SDR_ABLATE == 15 || SDR_ABLATE == 16)
        if (frame >= 0)  // timing-only build: no input DMA at all
            return;
#endif
#if defined(SDR_ABLATE) && (SDR_ABLATE == 8)
        const size_t fr = (in_band + frame) & 15;  // timing-only: 16 frames, L2-resident
#else
        const size_t fr = in_band + frame;
#endif
        const rsrc_t xrs = make_rsrc(iq + fr * PL::N * 2, PL::N * 8u);
        const int lane = tid & 63;
        // sample of (lane 2g, this wave, slot h): the slot's odd bit h = lane >> 5 selects sample bit 13
        const unsigned voff = (unsigned)(fft64::input_sample<LOGN>((tid & ~63) | (2 * (lane & 31)), 0) + ((lane >> 5) ? fft64::input_slot_sample<LOGN>(1) : 0)) * 8u;
#pragma unroll
        for (int j = 0; j < PL::R / 2; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(smem + wave * kBlock + j * 1024), 16,
                                                     voff, fft64::input_slot_sample<LOGN>(2 * j) * 8, 0, SDR_FFT_DMA_AUX);
    };
    const rsrc_t twr = make_rsrc(tw, (unsigned)(PL::TW_TOTAL * sizeof(fft64::cplx)));
    const int *bins = tap_bins + (size_t)blockIdx.y * tap_stride;
    const bool reg_tap = n_tap > 0 && n_tap <= PL::T;  // one listener per thread in registers; more take the plain loop
    const int my_bin = (reg_tap && tid < n_tap) ? bins[tid] : -1;
    float tap_val = 0.f;
    auto tap_frame_slow = [&](int frame) {
        const float *row = psd + (out_band + frame) * PL::N;
        float *out = tap_out + (out_band + frame) * (size_t)tap_stride;
        for (int l = tid; l < n_tap; l += PL::T) {
            const int bin = bins[l];
            out[l] = bin >= 0 ? row[bin] : 0.0f;
        }
    };
    if (frame0 < frame_end)
        stage_frame(frame0);
#pragma nounroll
    for (int frame = frame0; frame < frame_end; frame++) {
        // (everything derived from the thread id - twiddle positions, store offsets - is loop-invariant and the
        // compiler would hoist it out of the frame loop into registers it does not have: opaque per frame)
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        // this wave's DMA: the first frame's is its only traffic; a later one is older than the previous frame's R psd
        // stores, and vector memory operations retire in order
        if (frame == frame0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            SDR_WAIT_ALL_BUT(PL::R);
        SDR_STAMP(st, ST_LANDED);
        double xr[PL::R], xi[PL::R];
        {
            const unsigned char *blk = smem + wave * kBlock + (t & 63) * 8;
#pragma unroll
            for (int m = 0; m < PL::R; m++) {  // dsp/fft.go:59-69 setSamplesFromIQ
#if defined(SDR_ABLATE) && (SDR_ABLATE == 1 || SDR_ABLATE == 7)
                const float2 v = make_float2(1e-3f * (float)(t + m), 0.5f);  // timing-only build: no input
#else
                const float2 v = *reinterpret_cast<const float2 *>(blk + m * 512);
#endif
                xr[m] = (double)v.x;
                xi[m] = (double)v.y;
            }
        }
        SDR_STAMP(st, ST_LOADED);
        const bool more = frame + 1 < frame_end;
        const bool tap_prev = n_tap > 0 && frame > frame0;
        const fft64::cplx no_pre[kTwPreMax] = {};
        run_passes<LOGN, 0, true>(
            xr, xi, t, twr, tw, lds,
            [&] {  // the cross-wave exchange is over (its last barrier passed): LDS belongs to the next frame
                if (more)
                    stage_frame(frame + 1);
            },
            no_pre, st,
            [&](int e, bool after) {
                if (e == 0 && !after) {
                    // this wave's psd stores of the previous frame have completed before the barrier that follows
                    if (frame > frame0)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (e == 1 && tap_prev && reg_tap) {
                    if (!after) {
                        const float *row = psd + (out_band + frame - 1) * PL::N;
                        tap_val = my_bin >= 0 ? __builtin_nontemporal_load(row + my_bin) : 0.0f;
                    } else if (tid < n_tap) {
                        tap_out[(out_band + frame - 1) * (size_t)tap_stride + tid] = tap_val;
                    }
                }
            });
#if defined(SDR_FFT_B_LDS_EXCH)
        __syncthreads();  // (the epilogue's tile overlaps the tail of the padded exchange area: every wave is out of its last exchange)
#endif
        store_psd_b<LOGN>(xr, xi, t, psd + (out_band + frame) * PL::N, smem + kStageBytes);
        if (tap_prev && !reg_tap)
            tap_frame_slow(frame - 1);
        SDR_STAMP(st, ST_STORED);
    }
    if (n_tap > 0 && frame_end > frame0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tap_frame_slow(frame_end - 1);
    }
    SDR_STAMP(st, ST_END);
#if defined(SDR_FFT_PHASES)
    if (st.on && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kStampCount; k++)
            g_fft_phases[threadIdx.x >> 6][k] = st.v[k];
    }
#endif
#if defined(SDR_FFT_CLOCK)
    if (blockIdx.x == 100 && threadIdx.x == 0) {
        g_fft_clock[0] = __builtin_amdgcn_s_memtime() - ck0;
        g_fft_clock[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            g_fft_wg[blockIdx.x][0] = rt0;
            g_fft_wg[blockIdx.x][1] = now;
            g_fft_wg[blockIdx.x][3] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)) |
                                      ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11)) << 32);
        }
        atomicMax(&g_fft_wg[blockIdx.x][2], now);
    }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

// Tuning knob, read once per process: SDR_FFT_FPW = frames per workgroup.
// (LDS behind the exchange area: the one-frame workgroup's copy of its listeners' bins)
constexpr int kDefaultFpw = 1;  // in the pipeline short-lived workgroups win: 0.250 (1) / 0.253 (2) / 0.291 (4) / 0.294 ms (8) per step, standalone the other way round (0.174 / 0.166 / 0.165 / 0.164 ms)
constexpr int kMaxDevices = 64;
static int fft_fpw()
{
    static const int v = [] {
        if (const char *e = getenv("SDR_FFT_FPW"))
            return std::max(1, std::min(atoi(e), 64));
        return kDefaultFpw;
    }();
    return v;
}

// LDS of the layout B kernel: the cross-wave exchange's (padded) area, or the staging image plus the epilogue's tile
template <int LOGN>
inline constexpr int kLdsBytesB = fft64::kLdsBytes<LOGN> > fft64::Plan<LOGN>::N * 8 + fft64::Plan<LOGN>::N * 2
                                      ? fft64::kLdsBytes<LOGN>
                                      : fft64::Plan<LOGN>::N * 8 + fft64::Plan<LOGN>::N * 2;

// frames per workgroup of the layout B kernel: SDR_FFT_FPW overrides
constexpr int kDefaultFpwB = 8;
static int fft_fpw_b()
{
    static const int v = [] {
        if (const char *e = getenv("SDR_FFT_FPW"))
            return std::max(1, std::min(atoi(e), 64));
        return kDefaultFpwB;
    }();
    return v;
}

template <int LOGN>
static hipError_t launch_fft_t(const float *iq, const BatchCursor *cur, const fft64::cplx *tw, float *psd, int n_frames,
                               int n_bands, int in_stride, int out_stride, FftTap tap, hipStream_t stream)
{
    using PL = fft64::Plan<LOGN>;
    if constexpr (PL::LB) {
        static std::once_flag b_attr_once[kMaxDevices];
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess)
            return e;
        if (dev < 0 || dev >= kMaxDevices)
            return hipErrorInvalidDevice;
        hipError_t attr_err = hipSuccess;
        std::call_once(b_attr_once[dev], [&] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_psd_b<LOGN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kLdsBytesB<LOGN>);
        });
        if (attr_err != hipSuccess)
            return attr_err;
        if (n_frames <= 0 || n_bands <= 0)
            return hipSuccess;
        // a workgroup's frames are consecutive (the next one is prefetched under the current one's first passes);
        // never fewer workgroups than the chip has CUs
        int fpw = fft_fpw_b();
        while (fpw > 1 && (long)((n_frames + fpw - 1) / fpw) * n_bands < 256)
            fpw /= 2;
        launch_kernel((k_fft_psd_b<LOGN>), dim3((n_frames + fpw - 1) / fpw, n_bands), dim3(PL::T), kLdsBytesB<LOGN>, stream, iq, cur, tw, psd,
                      in_stride, out_stride, n_frames, fpw, tap.bins, tap.out, tap.n, tap.stride);
        return hipGetLastError();
    } else {
    // the > 64 KB dynamic LDS attribute is per device: set it once on each device a bank launches on
    static std::once_flag attr_once[kMaxDevices];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    if (dev < 0 || dev >= kMaxDevices)
        return hipErrorInvalidDevice;
    hipError_t attr_err = hipSuccess;
    std::call_once(attr_once[dev], [&] {
        for (const void *k : {reinterpret_cast<const void *>(&k_fft_psd<LOGN, false>),
                              reinterpret_cast<const void *>(&k_fft_psd<LOGN, true>)}) {
            const hipError_t ae = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, fft64::kLdsBytes<LOGN> + kMaxLdsTap * 4 + kPfSinkBytes);
            if (ae != hipSuccess)
                attr_err = ae;
        }
    });
    if (attr_err != hipSuccess)
        return attr_err;
    if (n_frames <= 0 || n_bands <= 0)
        return hipSuccess;
    // a workgroup's frames are consecutive; never fewer workgroups than CUs can take (a short batch keeps one
    // frame per workgroup)
    int fpw = fft_fpw();
    while (fpw > 1 && (long)((n_frames + fpw - 1) / fpw) * n_bands < 256)
        fpw /= 2;
    if (fpw > 1)
        launch_kernel((k_fft_psd<LOGN, true>), dim3((n_frames + fpw - 1) / fpw, n_bands), dim3(PL::T), fft64::kLdsBytes<LOGN>, stream,
                           iq, cur, tw, psd, in_stride, out_stride, n_frames, fpw, tap.bins, tap.out, tap.n, tap.stride);
    else
        launch_kernel((k_fft_psd<LOGN, false>), dim3(n_frames, n_bands), dim3(PL::T),
                           fft64::kLdsBytes<LOGN> + (tap.n > 0 && tap.n <= kMaxLdsTap ? ((tap.n * 4 + 255) & ~255) : 0) + kPfSinkBytes, stream, iq, cur, tw, psd, in_stride,
                           out_stride, n_frames, 1, tap.bins, tap.out, tap.n, tap.stride);
    return hipGetLastError();
    }
}

// N = 16384 has two kernels: this file's 16-point one and k_fft_r32.hip (512 threads x 32 points, the next frame
// prefetched into registers), whose workgroups take several frames each.  SDR_FFT_R32 = 0 / 1 forces one of them (tests);
// by default the 32-point kernel runs from 1024 frames per launch on (k_fft_r32.hip r32_fpw: measured by batch size).  The bank's
// twiddle buffer holds both kernels' tables, the 32-point kernel's behind the other.
static int r32_mode()
{
    static const int v = [] {
        if (const char *e = getenv("SDR_FFT_R32"))
            return atoi(e) ? 1 : 0;
        return -1;
    }();
    return v;
}

static bool use_r32(int logn, int n_frames, int n_bands, int tap_n)
{
    const int mode = r32_mode();
    return logn == 14 && tap_n <= fft32::T && (mode == 1 || (mode < 0 && (long)n_frames * n_bands >= 1024));
}
bool fft_writes_wide_tap(int logn, int n_frames, int n_bands, int tap_n) { return tap_n > 0 && use_r32(logn, n_frames, n_bands, tap_n); }

hipError_t launch_fft(int logn, const float *iq, const BatchCursor *cur, const fft64::cplx *tw, float *psd, int n_frames,
                      int n_bands, int in_stride, int out_stride, FftTap tap, hipStream_t stream)
{
    if (logn == 14) {
        if (use_r32(logn, n_frames, n_bands, tap.n))
            return launch_fft_r32(iq, cur, tw + fft64::Plan<14>::TW_TOTAL, psd, n_frames, n_bands, in_stride, out_stride, tap, stream);
    }
    switch (logn) {
    case 9: return launch_fft_t<9>(iq, cur, tw, psd, n_frames, n_bands, in_stride, out_stride, tap, stream);
    case 10: return launch_fft_t<10>(iq, cur, tw, psd, n_frames, n_bands, in_stride, out_stride, tap, stream);
    case 11: return launch_fft_t<11>(iq, cur, tw, psd, n_frames, n_bands, in_stride, out_stride, tap, stream);
    case 12: return launch_fft_t<12>(iq, cur, tw, psd, n_frames, n_bands, in_stride, out_stride, tap, stream);
    case 13: return launch_fft_t<13>(iq, cur, tw, psd, n_frames, n_bands, in_stride, out_stride, tap, stream);
    case 14: return launch_fft_t<14>(iq, cur, tw, psd, n_frames, n_bands, in_stride, out_stride, tap, stream);
    default: return hipErrorInvalidValue;
    }
}

int twiddle_count(int logn)
{
    switch (logn) {
    case 9: return fft64::Plan<9>::TW_TOTAL;
    case 10: return fft64::Plan<10>::TW_TOTAL;
    case 11: return fft64::Plan<11>::TW_TOTAL;
    case 12: return fft64::Plan<12>::TW_TOTAL;
    case 13: return fft64::Plan<13>::TW_TOTAL;
    case 14: return fft64::Plan<14>::TW_TOTAL + r32_twiddle_count();
    default: return 0;
    }
}

void build_twiddles(int logn, const double *wre, const double *wim, fft64::cplx *out)
{
    switch (logn) {
    case 9: fft64::build_pass_twiddles<9>(wre, wim, out); break;
    case 10: fft64::build_pass_twiddles<10>(wre, wim, out); break;
    case 11: fft64::build_pass_twiddles<11>(wre, wim, out); break;
    case 12: fft64::build_pass_twiddles<12>(wre, wim, out); break;
    case 13: fft64::build_pass_twiddles<13>(wre, wim, out); break;
    case 14:
        fft64::build_pass_twiddles<14>(wre, wim, out);
        r32_build_twiddles(wre, wim, out + fft64::Plan<14>::TW_TOTAL);
        break;
    default: break;
    }
}

}  // namespace sdr
