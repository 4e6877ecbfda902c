// k_fft_project.hip — the dominant kernel: IQ frame -> float64 radix-2 DIT FFT -> fftshift -> PSD / dB projection.
// Compiled with -ffp-contract=off (see gomath.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "fft_f64.h"
#include "gomath.h"
#include "sdr_device.h"

#if !defined(SDR_FFT_PSD_AUX)
#define SDR_FFT_PSD_AUX 0  // cache policy bits of the psd / spectrum stores (2 = nt)
#endif
#if !defined(SDR_FFT_SPEC_AUX)
#define SDR_FFT_SPEC_AUX 0
#endif
#if !defined(SDR_FFT_DMA_AUX)
#define SDR_FFT_DMA_AUX 2  // cache policy bits of the input LDS-DMA: nt - a frame is read once, by one CU (0.198 vs 0.202 ms)
#endif

namespace sdr {

// Development aids (tools/fft_trace.hip).  SDR_FFT_TRACE: per-wave time stamps of one workgroup's phases (the
// stamps cost a third of the kernel's speed: every one waits for the scalar-memory counter LDS shares).
// SDR_FFT_STOP: every wave ends at phase `g_fft_stop_at` (a uniform value read at run time), so the launch time
// of "everything up to phase k" can be measured on otherwise identical code.
#if defined(SDR_FFT_TRACE)
__shared__ int s_fft_trace_frame;  // which of the workgroup's frames is being stamped
#define SDR_STAMP(k)                                                                                  \
    do {                                                                                              \
        if (blockIdx.x == SDR_FFT_TRACE && (threadIdx.x & 63) == 0)                                   \
            g_fft_trace[s_fft_trace_frame & 1][threadIdx.x >> 6][k] = wall_clock64();                 \
    } while (0)
#elif defined(SDR_FFT_STOP)
__device__ int g_fft_stop_at;
#define SDR_STAMP(k)                                                 \
    do {                                                             \
        if (__builtin_amdgcn_readfirstlane(g_fft_stop_at) == (k))    \
            __builtin_amdgcn_endpgm();                               \
    } while (0)
#else
#define SDR_STAMP(k) \
    do {             \
    } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// k_fft_project  (dsp/fft.go:23-37 IQToSpectrumAndPSD + rx/receiver.go:376-378 projection closure)
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
        if (!fft64::make_swap_plan<LOGN>(k).ok)
            return true;
    return false;
}

// `lds_free()` is called once, when the frame's last exchange through LDS is over (a following frame's
// input may be staged into the exchange area from then on).
template <int LOGN, int P, bool FRAME_FOLLOWS, class LdsFree>
__device__ __forceinline__ void run_passes(double (&xr)[fft64::Plan<LOGN>::R], double (&xi)[fft64::Plan<LOGN>::R],
                                           int t, rsrc_t tw, const fft64::cplx *__restrict__ tw_ptr, double *lds,
                                           LdsFree lds_free)
{
    using PL = fft64::Plan<LOGN>;
#if !(defined(SDR_ABLATE) && (SDR_ABLATE == 5))
    fft64::butterfly_pass<LOGN, P>(xr, xi, t, [tw, tw_ptr](int c, int lo) {
        if constexpr (P == 0)
            return tw_ptr[c];  // pass 0: the same entry for every thread, a scalar load
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(tw, (unsigned)lo * 16u, c * 16, 0);
        fft64::cplx r;
        r.x = __hiloint2double((int)w.y, (int)w.x);
        r.y = __hiloint2double((int)w.w, (int)w.z);
        return r;
    });
#endif
    SDR_STAMP(2 + 2 * P);
    if constexpr (P < PL::NPASS - 1) {
        if constexpr (fft64::make_swap_plan<LOGN>(P).ok) {
            // slot bits <-> lane bits 4/5 only: done in registers (fft_f64.h exchange_swap), no LDS
#if !(defined(SDR_ABLATE) && (SDR_ABLATE == 14))
            fft64::exchange_swap<LOGN, P>(xr);
            fft64::exchange_swap<LOGN, P>(xi);
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
                wr(xi, lds + PL::N);
                sync();
                rd(xr, lds);
                rd(xi, lds + PL::N);
            }
            // reads done before a later exchange writes LDS again (other waves' words if CROSS)
            // ... or the next frame's staging does
            if constexpr (later_lds_exchange<LOGN>(P) || FRAME_FOLLOWS)
                sync();
            if constexpr (!later_lds_exchange<LOGN>(P))
                lds_free();
        }
        SDR_STAMP(3 + 2 * P);
        run_passes<LOGN, P + 1, FRAME_FOLLOWS>(xr, xi, t, tw, tw_ptr, lds, lds_free);
    }
}

// Epilogue (dsp/fft.go:54-57 fftshift, :71-73 PSD[float32], :79-81 MagnitudeIndB, rx/receiver.go:377
// +dBmShift).  The certified shortcut (gomath.h) settles all but about 2 values in 10^4; the rest are
// redone with the literal Go algorithm in a rolled loop that re-reads the PSD value just stored, so the
// long literal path exists once and holds no registers while the slots stream through.
// `after_slot(s)` lets the persistent kernel slip its prefetch between slots.
template <int LOGN, typename F>
__device__ __forceinline__ void project_and_store(const double (&xr)[fft64::Plan<LOGN>::R],
                                                  const double (&xi)[fft64::Plan<LOGN>::R], int t, float *__restrict__ sp,
                                                  float *__restrict__ pd, double inv_n2,
                                                  const gomath::LogTabEntry *ltab, F after_slot)
{
    using PL = fft64::Plan<LOGN>;
    unsigned redo = 0;
    const int tp = fft64::thread_part<LOGN, PL::NPASS - 1>(t);
    // fft-shift = flip the top index bit: in the slot part it is a compile-time constant, in the thread part
    // it is applied once; spectrum index k = tk | sk(s), and sk(s) goes into the scalar base pointer
    constexpr int SLOT_MASK = fft64::slot_part<LOGN, PL::NPASS - 1>(PL::R - 1);
    constexpr int H = PL::N / 2;
    const unsigned tk = (unsigned)(tp ^ (H & ~SLOT_MASK));
    const rsrc_t pdr = make_rsrc(pd, PL::N * 4u), spr = make_rsrc(sp, PL::N * 4u);
#pragma unroll
    for (int s = 0; s < PL::R; s++) {
        const int sk = fft64::slot_part<LOGN, PL::NPASS - 1>(s) ^ (H & SLOT_MASK);
        const float p = (float)(xr[s] * xr[s] + xi[s] * xi[s]);
#if defined(SDR_ABLATE) && (SDR_ABLATE == 15)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(p), pdr, tk * 4u, sk * 4, SDR_FFT_PSD_AUX);  // timing-only: psd only
#elif defined(SDR_ABLATE) && (SDR_ABLATE == 6 || SDR_ABLATE == 7)
        float db = 0.0f;  // timing-only build: (almost) no stores
        if (!gomath::psd_value_in_db_fast(p, inv_n2, ltab, &db))
            redo |= 1u << s;
        if (db == 1234.5f) {
            pd[tk | sk] = p;
            sp[tk | sk] = db;
        }
#else
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(p), pdr, tk * 4u, sk * 4, SDR_FFT_PSD_AUX);
        float db = 0.0f;
        if (!gomath::psd_value_in_db_fast(p, inv_n2, ltab, &db))
            redo |= 1u << s;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db + 120.0f), spr, tk * 4u, sk * 4, SDR_FFT_SPEC_AUX);
#endif
        after_slot(s);
    }
    while (redo) {
        const int s = __builtin_ctz(redo);
        redo &= redo - 1;
        int sl = 0;
        constexpr fft64::Layout L = fft64::make_layout<LOGN>(PL::NPASS - 1);
#pragma unroll
        for (int j = 0; j < PL::LOGR; j++)
            sl |= ((s >> j) & 1) << L.sbit[j];
        const int k = ((tp | sl) + PL::N / 2) & (PL::N - 1);
        const float p = __builtin_nontemporal_load(pd + k);  // this thread's own store, re-read
        sp[k] = gomath::psd_value_in_db(p, inv_n2) + 120.0f;
    }
}

// (A persistent variant - the grid sized to the chip, each workgroup walking over frames, the next frame's
// samples prefetched into the REGISTERS the projection frees, workgroups started staggered or not - measured
// 5 % slower at N = 16384: 0.223 against 0.212 ms for 2048 frames.)
//
// MULTI: a workgroup takes `fpw` consecutive frames.  The next frame's LDS-DMA is issued as soon as the current
// frame's last exchange through LDS is over (SDR_FFT_DMA_AT 0; LDS is idle from then on and the DMA needs no
// registers) or just before the projection (SDR_FFT_DMA_AT 1), so its HBM latency is covered by the current
// frame's remaining passes and projection, and there is one dispatch gap per `fpw` frames.
// `stagger_ticks` (100 MHz ticks): the first generation of workgroups (one per CU) starts in four phases a
// quarter of this apart.  Workgroups of one launch all do the same work, so without it every CU reads its
// frame at the same moment and stores its spectrum at the same moment: HBM sees 33 MB bursts with idle time
// between them, and a prefetch issued by all CUs at once lands in the middle of everybody's store burst.
#if !defined(SDR_FFT_DMA_AT)
#define SDR_FFT_DMA_AT 0
#endif
template <int LOGN, bool MULTI>
__global__ __launch_bounds__(fft64::Plan<LOGN>::T) void k_fft_project(const float *__restrict__ iq,
                                                                      const fft64::cplx *__restrict__ tw,
                                                                      float *__restrict__ spectrum,
                                                                      float *__restrict__ psd, double inv_n2,
                                                                      int in_stride, int out_stride, int n_frames,
                                                                      int fpw, int stagger_ticks)
{
    using PL = fft64::Plan<LOGN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *lds = reinterpret_cast<double *>(smem);
    // the 64-entry table of the certified fast dB path (gomath.h) follows the twiddles in HBM and sits
    // behind the exchange area in LDS; the staging barriers publish it long before the epilogue
    gomath::LogTabEntry *ltab = reinterpret_cast<gomath::LogTabEntry *>(smem + PL::LDS_BYTES);
#if defined(SDR_FFT_TRACE)
    if ((threadIdx.x & 63) == 0)
        s_fft_trace_frame = 0;
#endif
    if constexpr (MULTI) {
        if (stagger_ticks > 0 && blockIdx.x < 256u) {
            const unsigned phase = (blockIdx.x >> 3) & 3u;  // blocks b and b + 8 share an XCD: spread within each
            const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(phase * (unsigned)stagger_ticks / 4u);
            while (__builtin_amdgcn_s_memrealtime() < until)
                __builtin_amdgcn_s_sleep(8);
        }
    }
#if defined(SDR_FFT_CLOCK)
    unsigned long long ck0 = 0, rt0 = 0;
    if (blockIdx.x == 100)
        ck0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    SDR_STAMP(0);
    if (threadIdx.x < gomath::kLogTabSize)
        ltab[threadIdx.x] = reinterpret_cast<const gomath::LogTabEntry *>(tw + PL::TW_TOTAL)[threadIdx.x];
    const int frame0 = MULTI ? blockIdx.x * fpw : blockIdx.x;
    const int frame_end = MULTI ? min(frame0 + fpw, n_frames) : frame0 + 1;
    const size_t in_band = (size_t)blockIdx.y * in_stride, out_band = (size_t)blockIdx.y * out_stride;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

    // Frame -> LDS by LDS-DMA, one contiguous 1 KB row per wave instruction, shaped through the source address
    // (fft_f64.h "Input staging").  The staging image lives in the exchange area.
    auto stage_frame = [&](int frame, int tid) {
        constexpr int ROWS_PER_WAVE = PL::R / 2;
        const int lane = tid & 63;
#if defined(SDR_ABLATE) && (SDR_ABLATE == 8)
        const size_t fr = (in_band + frame) & 15;  // timing-only: 16 frames, L2-resident
#else
        const size_t fr = in_band + frame;
#endif
        // buffer form: row in the scalar offset, granule in one 32-bit VGPR - no 64-bit per-lane addresses
        const rsrc_t xrs = make_rsrc(iq + fr * PL::N * 2, PL::N * 8u);
#pragma unroll
        for (int j = 0; j < ROWS_PER_WAVE; j++) {
            const int r = wave * ROWS_PER_WAVE + j;
            const int g = fft64::in_granule<LOGN>(lane, r);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(smem + r * 1024), 16,
                                                     (unsigned)g * 16u, r * 1024, 0, SDR_FFT_DMA_AUX);
        }
    };
    stage_frame(frame0, threadIdx.x);

#pragma nounroll
    for (int frame = frame0; frame < frame_end; frame++) {
#if defined(SDR_FFT_TRACE)
        if ((threadIdx.x & 63) == 0)
            s_fft_trace_frame = frame - frame0;  // (every wave writes the same value; its own lane 0 reads it back)
#endif
        // (with more than one frame per workgroup everything derived from the thread id is loop-invariant and the
        // compiler would hoist - and spill - it: make the thread id opaque per frame)
        int t = threadIdx.x;
        if constexpr (MULTI)
            asm volatile("" : "+v"(t));
        // Vector-memory operations retire in issue order for the counter (MI355X_MICROARCH.md: loads, stores and
        // LDS-DMA count together, in issue order).  The first frame's DMA is the wave's only traffic: full drain.
        // A later frame's DMA was issued BEFORE the previous frame's projection stores (2 per slot, plus whatever
        // the rare redo loop added), so "all but the 2R youngest" covers it without also waiting for those stores
        // to reach memory; with redo traffic on top it merely waits for a few of the stores as well.
        if (!MULTI || frame == frame0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PL::R) : "memory");
        SDR_STAMP(12);  // this wave's rows have landed
        __syncthreads();
        SDR_STAMP(13);  // everybody's have

        double xr[PL::R], xi[PL::R];
        const int n_thread = fft64::input_sample<LOGN>(t, 0);
        const int thread_byte = fft64::in_lds_byte<LOGN>(n_thread);
#pragma unroll
        for (int m = 0; m < PL::R; m++) {
#if defined(SDR_ABLATE) && (SDR_ABLATE == 1 || SDR_ABLATE == 7)
            const float2 v = make_float2(1e-3f * (float)(t + m), 0.5f);  // timing-only build: no input
#else
            // sample number -> image address is linear over GF(2): thread part and slot part combine by XOR,
            // and the slot part is a compile-time constant
            const int slot_byte = fft64::in_lds_byte<LOGN>(fft64::input_sample<LOGN>(0, m));
            const float2 v = *reinterpret_cast<const float2 *>(smem + (thread_byte ^ slot_byte));
#endif
            xr[m] = (double)v.x;
            xi[m] = (double)v.y;
        }
        __syncthreads();  // everyone has its samples: the exchange area may be written again
        SDR_STAMP(1);
        const bool more = MULTI && frame + 1 < frame_end;
        // (no scheduling pin around the DMA: the compiler keeps it behind the exchanges' LDS accesses and behind the
        // twiddle loads already issued, which it waits for with counted vmcnt; a "memory" pin cost 46 spills)
        run_passes<LOGN, 0, MULTI>(xr, xi, t, make_rsrc(tw, (unsigned)(PL::TW_TOTAL * sizeof(fft64::cplx))), tw, lds, [&] {
            if constexpr (MULTI && SDR_FFT_DMA_AT == 0)
                if (more)
                    stage_frame(frame + 1, t);
        });
        if constexpr (MULTI && SDR_FFT_DMA_AT == 1)
            if (more)
                stage_frame(frame + 1, t);
        float *sp = spectrum + (out_band + frame) * PL::N;
        float *pd = psd + (out_band + frame) * PL::N;
#if defined(SDR_ABLATE) && (SDR_ABLATE == 4)
#pragma unroll
        for (int s = 0; s < PL::R; s++) {  // timing-only build: no projection
            const int kk = fft64::output_bin<LOGN>(t, s);
            pd[kk] = (float)xr[s];
            sp[kk] = (float)xi[s];
        }
#else
        project_and_store<LOGN>(xr, xi, t, sp, pd, inv_n2, ltab, [](int) {});
#endif
        SDR_STAMP(10);
    }
#if defined(SDR_FFT_TRACE)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SDR_STAMP(11);
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
}

constexpr int kLogTabBytes = gomath::kLogTabSize * (int)sizeof(gomath::LogTabEntry);

// Tuning knobs, read once per process: SDR_FFT_FPW = frames per workgroup (default kDefaultFpw),
// SDR_FFT_STAGGER_US = spread of the first workgroups' start times in microseconds.
constexpr int kDefaultFpw = 1;
constexpr int kMaxDevices = 64;
struct FftKnobs {
    int fpw = kDefaultFpw;
    int stagger_ticks = 0;
};
static const FftKnobs &fft_knobs()
{
    static const FftKnobs k = [] {
        FftKnobs v;
        if (const char *e = getenv("SDR_FFT_FPW"))
            v.fpw = std::max(1, std::min(atoi(e), 64));
        if (const char *e = getenv("SDR_FFT_STAGGER_US"))
            v.stagger_ticks = std::max(0, std::min((int)(atof(e) * 100.0), 100000));
        return v;
    }();
    return k;
}

template <int LOGN>
static hipError_t launch_fft_t(const float *iq, const fft64::cplx *tw, float *spectrum, float *psd, int n_frames,
                               int n_bands, int in_stride, int out_stride, hipStream_t stream)
{
    using PL = fft64::Plan<LOGN>;
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
        for (const void *k : {reinterpret_cast<const void *>(&k_fft_project<LOGN, false>),
                              reinterpret_cast<const void *>(&k_fft_project<LOGN, true>)}) {
            const hipError_t ae = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, PL::LDS_BYTES + kLogTabBytes);
            if (ae != hipSuccess)
                attr_err = ae;
        }
    });
    if (attr_err != hipSuccess)
        return attr_err;
    const FftKnobs &kn = fft_knobs();
    const double inv_n2 = 1.0 / ((double)PL::N * (double)PL::N);
    if (n_frames <= 0 || n_bands <= 0)
        return hipSuccess;
    if (kn.fpw > 1)
        hipLaunchKernelGGL((k_fft_project<LOGN, true>), dim3((n_frames + kn.fpw - 1) / kn.fpw, n_bands), dim3(PL::T),
                           PL::LDS_BYTES + kLogTabBytes, stream, iq, tw, spectrum, psd, inv_n2, in_stride, out_stride, n_frames,
                           kn.fpw, kn.stagger_ticks);
    else
        hipLaunchKernelGGL((k_fft_project<LOGN, false>), dim3(n_frames, n_bands), dim3(PL::T), PL::LDS_BYTES + kLogTabBytes, stream,
                           iq, tw, spectrum, psd, inv_n2, in_stride, out_stride, n_frames, 1, 0);
    return hipGetLastError();
}

hipError_t launch_fft(int logn, const float *iq, const fft64::cplx *tw, float *spectrum, float *psd, int n_frames,
                      int n_bands, int in_stride, int out_stride, hipStream_t stream)
{
    switch (logn) {
    case 9: return launch_fft_t<9>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 10: return launch_fft_t<10>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 11: return launch_fft_t<11>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 12: return launch_fft_t<12>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 13: return launch_fft_t<13>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 14: return launch_fft_t<14>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
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
    case 14: return fft64::Plan<14>::TW_TOTAL;
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
    case 14: fft64::build_pass_twiddles<14>(wre, wim, out); break;
    default: break;
    }
}

}  // namespace sdr
