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
// The next frame's 32 loads per thread are dealt over the frame (loads per program point; fifteen points: behind the
// widening, behind each of pass 0's five stages, at the four steps of E0, in front of pass 1, behind pass 1's stages 0-3).
// Issued in one go - 256 wave instructions, 128 KB per CU - they do not overlap anything: a wave cannot issue a vector
// memory instruction while the CU's memory pipeline is backed up, and it takes the pipeline 5 us to work a frame off
// (first build: 13 000 of a frame's 44 000 clocks went into issuing them).  Nothing after pass 1's stage 3: the last
// group needs time to land before pass 2's twiddle loads, which return behind it (vector memory retires in order).
#if !defined(SDR_R32_PF_PLAN)
#define SDR_R32_PF_PLAN {4, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2}
#endif
#if !defined(SDR_R32_CHUNK1)
#define SDR_R32_CHUNK1 4  // twiddles fetched per chunk in pass 1 (0: the compiler decides - and spills)
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
enum R32Stamp { RS_TOP = 0, RS_LANDED, RS_WIDENED, RS_PASS0, RS_E0, RS_PASS1, RS_E1, RS_PASS2, RS_ROW, RS_STORED, RS_COUNT };
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

constexpr int kPfPoints = 15;
constexpr int kPfPlan[kPfPoints] = SDR_R32_PF_PLAN;
constexpr int pf_begin(int point)
{
    int b = 0;
    for (int i = 0; i < point; i++)
        b += kPfPlan[i];
    return b;
}
static_assert(pf_begin(kPfPoints) == fft32::R, "the prefetch plan must cover the frame's 32 slots");
enum PfPoint { PF_WIDENED = 0, PF_PASS0 = 1 /* +q */, PF_E0 = 6 /* +step */, PF_PRE1 = 10, PF_PASS1 = 11 /* +q */ };

constexpr int kTw1LdsBytes = fft32::kTw1Entries * 16;
constexpr int kLdsBytes = fft32::kExchangeBytes + kTw1LdsBytes;
static_assert(kLdsBytes <= 160 * 1024, "LDS of one workgroup");
static_assert(fft32::kExchangeBytes % 16 == 0, "the twiddle block is read with ds_read_b128");
static_assert(fft32::N * 4 <= fft32::kExchangeBytes, "the psd row lives in the exchange area");

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
                                                          float *__restrict__ tap_out, int n_tap, int tap_stride)
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
    // the listeners' bins: slot l of the band belongs to thread l (more than 512: the loop at the end of a frame)
    const int *bins = tap_bins + (size_t)blockIdx.y * tap_stride;
    const bool reg_tap = n_tap > 0 && n_tap <= T;
    const int my_bin = (reg_tap && tid < n_tap) ? bins[tid] : -1;
    __syncthreads();
    R32Stamps st;
#if defined(SDR_R32_PHASES)
    st.on = false;
    st.v = 0;
#endif

#pragma nounroll
    for (int frame = frame0; frame < frame_end; frame++) {
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
        // (the requests go out BEHIND the conversions: hoisted above them - they depend on nothing - both frames' samples
        // would be live at once, 64 registers more than there are; the pins keep the conversions from being sunk below
        // the branch around the requests)
#pragma unroll
        for (int m = 0; m < R; m++)
            asm volatile("" : "+v"(xr[m]), "+v"(xi[m]));
        auto pf_point = [&](int point) {
            __builtin_amdgcn_sched_barrier(0);
            fetch(frame + 1, t, pf_begin(point), pf_begin(point + 1));
            __builtin_amdgcn_sched_barrier(0);
        };
        pf_point(PF_WIDENED);
        SDR_R32_STAMP(st, RS_WIDENED);

        run_pass<5, true, SDR_R32_CHUNK0>(xr, xi, [tw](int row, int) { return tw[kTw0 + row]; }, [&](int q) { pf_point(PF_PASS0 + q); });
        SDR_R32_STAMP(st, RS_PASS0);

        // E0: everybody is out of the previous frame's psd row (it shares the area), then real / imaginary rounds
        __syncthreads();
        ex_write<0, 0>(xr, t, ex);
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
        __syncthreads();  // E1 writes the wave's own block, which other waves have just read from
        SDR_R32_STAMP(st, RS_E0);
        pf_point(PF_PRE1);

        {
            const unsigned char *row0 = tw1_lds + tw1_lo(t) * 16;
            run_pass<5, false, SDR_R32_CHUNK1>(
                xr, xi, [row0](int row, int) { return *reinterpret_cast<const cplx *>(row0 + row * 512); },
                [&](int q) {
                    if (q < 4)
                        pf_point(PF_PASS1 + q);
                });
        }

        SDR_R32_STAMP(st, RS_PASS1);
        // E1: inside the wave's own block
        ex_write<1, 1>(xr, t, ex);
        wave_sync();
        ex_read<1, 2>(xr, t, ex);
        wave_sync();
        ex_write<1, 1>(xi, t, ex);
        wave_sync();
        ex_read<1, 2>(xi, t, ex);
        SDR_R32_STAMP(st, RS_E1);

        {
            const unsigned p0 = (unsigned)tw2_pos(t, 0) * 16u;
            run_pass<4, false, SDR_R32_CHUNK2>(xr, xi, [twr, p0](int row, int u) {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(twr, p0, (kTw2 + row * 1024 + u * 512) * 16, 0);
                cplx r;
                r.x = __hiloint2double((int)w.y, (int)w.x);
                r.y = __hiloint2double((int)w.w, (int)w.z);
                return r;
            });
        }

        SDR_R32_STAMP(st, RS_PASS2);
        // Epilogue (dsp/fft.go:54-57 fftshift, :71-73 PSD[float32]): psd[k] = float32(re^2 + im^2), two multiplies and
        // an add in float64, rounded once; into the LDS row at spectrum index k = bin ^ N/2
        __syncthreads();  // every wave is out of E1
        {
            constexpr int SLOT_MASK = slot_part(2, R - 1);
            static_assert((SLOT_MASK & 0xfc) == 0, "the row swizzle reads thread bits only");
            const int tk = thread_part<2>(t) ^ ((N / 2) & ~SLOT_MASK);
            unsigned char *rowp = smem + row_word(tk) * 4;
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int sk = slot_part(2, s) ^ ((N / 2) & SLOT_MASK);
                const float p = (float)(xr[s] * xr[s] + xi[s] * xi[s]);
                *reinterpret_cast<float *>(rowp + sk * 4) = p;
            }
        }
        __syncthreads();
        SDR_R32_STAMP(st, RS_ROW);
        {
            const rsrc_t pdr = make_rsrc(psd + (out_band + frame) * (size_t)N, N * 4u);
#pragma unroll
            for (int j = 0; j < N / 4 / T; j++) {
                const int c = t + T * j;  // 16-byte chunk of the row
                const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + row_word(4 * c) * 4);
                // (the chunk's offset goes into the VECTOR offset, the scalar offset stays the literal 0.  A 16-byte buffer
                // store reads its data registers some cycles after it issues; with an immediate scalar offset hipcc pads
                // a following VALU write of those registers with wait states, with an SGPR offset it assumes no hazard -
                // and on gfx950 there is one: built that way, 0.3 % of the psd words of every launch came out as the
                // next chunk's LDS address, which the compiler had put into the first data register right behind the
                // store.)
                __builtin_amdgcn_raw_buffer_store_b128(v, pdr, (unsigned)c * 16u, 0, 0);
            }
            // the tap (rx/receiver.go:393-394: spectrum[SignalBin] per listener and frame; the dB projection is applied
            // where it is consumed, k_listen.hip)
            const float *row = reinterpret_cast<const float *>(smem);
            float *out = tap_out + (out_band + frame) * (size_t)tap_stride;
            if (reg_tap) {
                if (tid < n_tap)
                    out[tid] = my_bin >= 0 ? row[row_word(my_bin)] : 0.0f;
            } else {
                for (int l = tid; l < n_tap; l += T) {
                    const int bin = bins[l];
                    out[l] = bin >= 0 ? row[row_word(bin)] : 0.0f;
                }
            }
        }
        SDR_R32_STAMP(st, RS_STORED);
#if defined(SDR_R32_PHASES)
        if (st.on && (threadIdx.x & 63) < RS_COUNT)
            g_r32_phases[threadIdx.x >> 6][threadIdx.x & 63] = st.v;
#endif
    }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace r32

// frames per workgroup of the R32 kernel (SDR_FFT_R32_FPW overrides)
static int r32_fpw()
{
    static const int v = [] {
        if (const char *e = getenv("SDR_FFT_R32_FPW"))
            return std::max(1, std::min(atoi(e), 1024));
        return 4;
    }();
    return v;
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
    // a workgroup's frames are consecutive; never fewer workgroups than the chip has CUs
    int fpw = r32_fpw();
    while (fpw > 1 && (long)((n_frames + fpw - 1) / fpw) * n_bands < 256)
        fpw /= 2;
    launch_kernel(r32::k_fft_r32, dim3((n_frames + fpw - 1) / fpw, n_bands), dim3(fft32::T), r32::kLdsBytes, stream, iq, cur, tw, psd,
                  in_stride, out_stride, n_frames, fpw, tap.bins, tap.out, tap.n, tap.stride);
    return hipGetLastError();
}

}  // namespace sdr
