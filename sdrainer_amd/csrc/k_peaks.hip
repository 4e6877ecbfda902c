// k_peaks.hip — ordered float32 cumulation over 100 frames and the run-length peak scan of each completed cumulation.
#include <hip/hip_runtime.h>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "fft_f64.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

// ---------------------------------------------------------------------------------------------
// k_cumulate — cumulation[i] += spectrum[i] (rx/receiver.go:404-407): a float32 sum in frame order.
// Slot 0 continues the cumulation carried over from the previous batch; a slot that reaches 100
// frames is written out for the peak scan, an incomplete last slot becomes the next carry.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cumulate(const float *__restrict__ spectrum, const float *__restrict__ carry_in,
                                                  float *__restrict__ carry_out, float *__restrict__ cum_out,
                                                  CumGeom g)
{
    const int i4 = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 bins
    if (i4 * 4 >= g.n)
        return;
    const int slot = blockIdx.y, band = blockIdx.z;
    // frames of this slot: slot 0 takes (100 - count0) frames, later slots 100 each
    const int first_len = SDR_CUMULATION_SIZE - g.count0;
    const int begin = slot == 0 ? 0 : first_len + (slot - 1) * SDR_CUMULATION_SIZE;
    const int len = slot == 0 ? first_len : SDR_CUMULATION_SIZE;
    const int end = min(begin + len, g.n_frames);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot == 0 && g.count0 > 0)
        acc = reinterpret_cast<const float4 *>(carry_in + (size_t)band * g.n)[i4];
    const float4 *sp = reinterpret_cast<const float4 *>(spectrum + (size_t)band * g.stride * g.n) + i4;
    const size_t fstride = g.n / 4;
    int f = begin;
    for (; f + 10 <= end; f += 10) {  // ten independent 16-byte loads in flight, then ten ORDERED adds
        float4 v[10];
#pragma unroll
        for (int k = 0; k < 10; k++)
            v[k] = sp[(size_t)(f + k) * fstride];
#pragma unroll
        for (int k = 0; k < 10; k++) {
            acc.x += v[k].x;
            acc.y += v[k].y;
            acc.z += v[k].z;
            acc.w += v[k].w;
        }
    }
    for (; f < end; f++) {
        const float4 v = sp[(size_t)f * fstride];
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
    }
    const bool complete = (begin + len) <= g.n_frames;
    if (complete) {
        // completed chunk index == slot (slot 0 completes first if it completes at all)
        reinterpret_cast<float4 *>(cum_out + ((size_t)band * g.max_chunks + slot) * g.n)[i4] = acc;
    } else {
        reinterpret_cast<float4 *>(carry_out + (size_t)band * g.n)[i4] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// k_find_peaks — dsp.FindPeaks (dsp/fft.go:254-285) on one completed cumulation: flag bins whose
// value/100 exceeds the threshold of the completing frame, number the runs with a workgroup prefix
// sum, and let the thread that owns a run start walk it (first maximum wins, strict `<`, :270).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_find_peaks(const float *__restrict__ cum, const sdr_frame_rec *__restrict__ recs,
                                                    DevPeak *__restrict__ peaks, int *__restrict__ counts, PeakGeom g)
{
    __shared__ int s_scan[256];
    const int chunk = blockIdx.x, band = blockIdx.y, tid = threadIdx.x;
    const int first_len = SDR_CUMULATION_SIZE - g.count0;
    const int end_frame = first_len + chunk * SDR_CUMULATION_SIZE - 1;  // frame that completes this chunk
    const float thr = recs[(size_t)band * g.stride + end_frame].peak_thr;
    const float *c = cum + ((size_t)band * g.max_chunks + chunk) * g.n;
    const float size = (float)SDR_CUMULATION_SIZE;
    const int per = g.n / 256;
    const int base = tid * per;
    bool prev = base > 0 ? (__fdiv_rn(c[base - 1], size) > thr) : false;
    int starts = 0;
    for (int i = base; i < base + per; i++) {
        const bool fl = __fdiv_rn(c[i], size) > thr;
        if (fl && !prev)
            starts++;
        prev = fl;
    }
    s_scan[tid] = starts;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
        int v = 0;
        if (tid >= off)
            v = s_scan[tid - off];
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int idx = s_scan[tid] - starts;
    if (tid == 255)
        counts[(size_t)band * g.max_chunks + chunk] = s_scan[255];
    prev = base > 0 ? (__fdiv_rn(c[base - 1], size) > thr) : false;
    for (int i = base; i < base + per; i++) {
        const float value = __fdiv_rn(c[i], size);
        const bool fl = value > thr;
        if (fl && !prev) {
            float best = value;
            int best_bin = i;
            int j = i + 1;
            for (; j < g.n; j++) {
                const float vj = __fdiv_rn(c[j], size);
                if (!(vj > thr))
                    break;
                if (best < vj) {
                    best = vj;
                    best_bin = j;
                }
            }
            if (idx < g.max_peaks) {
                DevPeak p;
                p.from = i;
                p.to = j - 1;  // also N-1 for a run still open at the last bin (:276-282)
                p.signal_bin = best_bin;
                p.signal_value = best;
                p.y1 = best_bin > 0 ? c[best_bin - 1] : 0.f;
                p.y2 = c[best_bin];
                p.y3 = best_bin < g.n - 1 ? c[best_bin + 1] : 0.f;
                peaks[((size_t)band * g.max_chunks + chunk) * g.max_peaks + idx] = p;
            }
            idx++;
        }
        prev = fl;
    }
}

hipError_t launch_cumulate(const float *spectrum, const float *carry_in, float *carry_out, float *cum_out, CumGeom g,
                           int n_slots, int n_bands, hipStream_t stream)
{
    hipLaunchKernelGGL(k_cumulate, dim3((g.n / 4 + 255) / 256, n_slots, n_bands), dim3(256), 0, stream, spectrum,
                       carry_in, carry_out, cum_out, g);
    return hipGetLastError();
}

hipError_t launch_find_peaks(const float *cum, const sdr_frame_rec *recs, DevPeak *peaks, int *counts, PeakGeom g,
                             int n_chunks, int n_bands, hipStream_t stream)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_find_peaks, dim3(n_chunks, n_bands), dim3(256), 0, stream, cum, recs, peaks, counts, g);
    return hipGetLastError();
}

}  // namespace sdr
