// k_fft_project.hip — the dominant kernel: IQ frame -> float64 radix-2 DIT FFT -> fftshift -> PSD / dB projection.
// Compiled with -ffp-contract=off (see gomath.h).
#include <hip/hip_runtime.h>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "fft_f64.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

// ---------------------------------------------------------------------------------------------
// k_fft_project  (dsp/fft.go:23-37 IQToSpectrumAndPSD + rx/receiver.go:376-378 projection closure)
// ---------------------------------------------------------------------------------------------
template <int LOGN, int P>
__device__ __forceinline__ void run_passes(double (&xr)[fft64::Plan<LOGN>::R], double (&xi)[fft64::Plan<LOGN>::R],
                                           int t, const fft64::cplx *__restrict__ tw, double *lds)
{
    using PL = fft64::Plan<LOGN>;
    fft64::butterfly_pass<LOGN, P>(xr, xi, t, tw);
    if constexpr (P < PL::NPASS - 1) {
        if constexpr (P > 0)
            __syncthreads();  // everyone is done reading the previous exchange
        if constexpr (PL::SPLIT) {
            fft64::exchange_write<LOGN, P>(xr, t, lds);
            __syncthreads();
            fft64::exchange_read<LOGN, P>(xr, t, lds);
            __syncthreads();
            fft64::exchange_write<LOGN, P>(xi, t, lds);
            __syncthreads();
            fft64::exchange_read<LOGN, P>(xi, t, lds);
        } else {
            fft64::exchange_write<LOGN, P>(xr, t, lds);
            fft64::exchange_write<LOGN, P>(xi, t, lds + PL::N);
            __syncthreads();
            fft64::exchange_read<LOGN, P>(xr, t, lds);
            fft64::exchange_read<LOGN, P>(xi, t, lds + PL::N);
        }
        run_passes<LOGN, P + 1>(xr, xi, t, tw, lds);
    }
}

template <int LOGN>
__global__ __launch_bounds__(fft64::Plan<LOGN>::T) void k_fft_project(const float *__restrict__ iq,
                                                                      const fft64::cplx *__restrict__ tw,
                                                                      float *__restrict__ spectrum,
                                                                      float *__restrict__ psd, double inv_n2,
                                                                      int in_stride, int out_stride)
{
    using PL = fft64::Plan<LOGN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *lds = reinterpret_cast<double *>(smem);
    // the 64-entry table of the certified fast dB path (gomath.h) follows the twiddles in HBM and sits
    // behind the exchange area in LDS; the exchanges' barriers publish it long before the epilogue
    gomath::LogTabEntry *ltab = reinterpret_cast<gomath::LogTabEntry *>(smem + PL::LDS_BYTES);
    const int t = threadIdx.x;
    if (t < gomath::kLogTabSize)
        ltab[t] = reinterpret_cast<const gomath::LogTabEntry *>(tw + PL::TW_TOTAL)[t];
    const size_t in_frame = (size_t)blockIdx.y * in_stride + blockIdx.x;
    const size_t out_frame = (size_t)blockIdx.y * out_stride + blockIdx.x;
    const float2 *x = reinterpret_cast<const float2 *>(iq) + in_frame * PL::N;

    double xr[PL::R], xi[PL::R];
#pragma unroll
    for (int m = 0; m < PL::R; m++) {
        // slot m <- x[bitrev(m) * T + t]: 512-byte contiguous segments per wave (fft_f64.h load_input)
        const int k = (int)fft64::brev_bits((unsigned)m, PL::LOGR);
        const float2 v = x[k * PL::T + t];
        xr[m] = (double)v.x;
        xi[m] = (double)v.y;
    }
    run_passes<LOGN, 0>(xr, xi, t, tw, lds);

    float *sp = spectrum + out_frame * PL::N;
    float *pd = psd + out_frame * PL::N;
#pragma unroll
    for (int s = 0; s < PL::R; s++) {
        const int i = fft64::output_bin<LOGN>(t, s);
        const int k = (i + PL::N / 2) & (PL::N - 1);                  // dsp/fft.go:54-57
        const float p = (float)(xr[s] * xr[s] + xi[s] * xi[s]);       // dsp/fft.go:71-73 PSD[float32]
        pd[k] = p;
        float db;                                                     // dsp/fft.go:79-81 MagnitudeIndB
        if (!gomath::psd_value_in_db_fast(p, inv_n2, ltab, &db))      // certified shortcut, else the literal
            db = gomath::psd_value_in_db(p, inv_n2);                  // Go algorithm (about 2 values in 10^4)
        sp[k] = db + 120.0f;                                          // + dBmShift (rx/receiver.go:377)
    }
}

constexpr int kLogTabBytes = gomath::kLogTabSize * (int)sizeof(gomath::LogTabEntry);

template <int LOGN>
static hipError_t launch_fft_t(const float *iq, const fft64::cplx *tw, float *spectrum, float *psd, int n_frames,
                               int n_bands, int in_stride, int out_stride, hipStream_t stream)
{
    using PL = fft64::Plan<LOGN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_project<LOGN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PL::LDS_BYTES + kLogTabBytes);
        if (e != hipSuccess)
            return e;
        attr_set = true;
    }
    const double inv_n2 = 1.0 / ((double)PL::N * (double)PL::N);
    hipLaunchKernelGGL(k_fft_project<LOGN>, dim3(n_frames, n_bands), dim3(PL::T), PL::LDS_BYTES + kLogTabBytes, stream, iq, tw,
                       spectrum, psd, inv_n2, in_stride, out_stride);
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
