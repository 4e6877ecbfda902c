// k_noise.hip — dsp.FindNoiseFloor (dsp/fft.go:215-252) and the rolling thresholds of Receiver.run
// (rx/receiver.go:381-385).  Compiled with -ffp-contract=off.
//
// FindNoiseFloor is two SEQUENTIAL float64 accumulations per frame (window sums, then the variance
// about the winning window's mean): float64 addition is not associative, so to reproduce the
// reference's bits each chain keeps its order — one lane per chain.  What is parallel is everything
// around the chain: a wave owns 64 chains of 64 consecutive frames, its lanes fetch each chain's next
// 64 values with one fully coalesced 256-byte load per chain, transpose them through LDS, and only then
// does every lane walk its own row.  The next tile's loads are in flight while the current one is
// consumed.
#include <hip/hip_runtime.h>

#include "../../include/sdrainer_hip.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

constexpr int TILE = 64;

// Stage tile `t` (64 columns) of 64 equally long rows that start `row_stride` floats apart into regs.
__device__ __forceinline__ void tile_load(const float *__restrict__ base, size_t row_stride, int rows, int col0,
                                          int n_cols, int lane, float (&regs)[TILE])
{
    const int col = col0 + lane;
#pragma unroll
    for (int r = 0; r < TILE; r++)
        regs[r] = (r < rows && col < n_cols) ? base[(size_t)r * row_stride + col] : 0.f;
}

__device__ __forceinline__ void tile_store(float (*tile)[TILE + 1], int lane, const float (&regs)[TILE])
{
#pragma unroll
    for (int r = 0; r < TILE; r++)
        tile[r][lane] = regs[r];
}

// pass 1: mean of window w of frame f = sequential float64 sum of psd[edge + w*W .. +W) / W (:239-241,:230)
__global__ __launch_bounds__(64) void k_window_means(const float *__restrict__ psd, double *__restrict__ win_mean,
                                                     NoiseGeom g, int n_frames, int stride)
{
    __shared__ float tile[TILE][TILE + 1];
    const int lane = threadIdx.x;
    const int f0 = blockIdx.x * TILE, w = blockIdx.y, band = blockIdx.z;
    const int rows = min(TILE, n_frames - f0);
    const size_t frame0 = (size_t)band * stride + f0;
    const float *base = psd + frame0 * g.n + g.edge + (size_t)w * g.window;
    const int n_tiles = (g.window + TILE - 1) / TILE;
    float regs[TILE];
    tile_load(base, g.n, rows, 0, g.window, lane, regs);
    double sum = 0;
    for (int t = 0; t < n_tiles; t++) {
        tile_store(tile, lane, regs);
        __syncthreads();
        if (t + 1 < n_tiles)
            tile_load(base, g.n, rows, (t + 1) * TILE, g.window, lane, regs);
        const int lim = min(TILE, g.window - t * TILE);
        for (int j = 0; j < lim; j++)
            sum += (double)tile[lane][j];
        __syncthreads();
    }
    if (lane < rows)
        win_mean[(frame0 + lane) * 10 + w] = sum / (double)g.window;
}

// pass 2: pick the minimum window in the reference's order, then the variance chain over
// psd[edge .. resultTo] (inclusive) about that mean, divided by windowSize (:244-249, App. C1);
// finally the two dB inputs of the rolling means (rx/receiver.go:383-384).
__global__ __launch_bounds__(64) void k_noise_stats(const float *__restrict__ psd, const double *__restrict__ win_mean,
                                                    sdr_frame_rec *__restrict__ recs, NoiseGeom g, int n_frames,
                                                    int stride)
{
    __shared__ float tile[TILE][TILE + 1];
    const int lane = threadIdx.x;
    const int f0 = blockIdx.x * TILE, band = blockIdx.y;
    const int rows = min(TILE, n_frames - f0);
    const size_t frame0 = (size_t)band * stride + f0;
    const bool valid = lane < rows;
    const size_t frame = frame0 + (valid ? lane : 0);
    const float *p = psd + frame * g.n;

    double minValue = (double)p[0];
    bool first = true;
    double resultMean = 0;
    int n_terms = 1;  // resultTo - resultFrom + 1 when no window qualifies (resultFrom = resultTo = 0)
    bool from_edge = false;
    for (int w = 0; w < g.n_windows; w++) {
        const double mean = win_mean[frame * 10 + w];
        if (mean < minValue || first) {  // :232
            minValue = mean;
            first = false;
            resultMean = mean;
            from_edge = true;  // resultFrom = edge: `from` is only assigned on the first iteration (App. C1)
            n_terms = (w + 1) * g.window + 1;  // resultTo = edge + (w+1)*W
        }
    }
    if (!valid)
        n_terms = 0;
    // chain length of this wave = the longest lane; lanes past their own end just skip the add
    int max_terms = n_terms;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        max_terms = max(max_terms, __shfl_xor(max_terms, o));
    const float *base = psd + frame0 * g.n + (from_edge ? g.edge : 0);
    // (from_edge is wave-uniform whenever n_windows >= 1, which sdr_create guarantees)
    const int n_cols = g.n - g.edge;  // never read past the frame
    const int n_tiles = (max_terms + TILE - 1) / TILE;
    float regs[TILE];
    tile_load(base, g.n, rows, 0, n_cols, lane, regs);
    double sum = 0;
    for (int t = 0; t < n_tiles; t++) {
        tile_store(tile, lane, regs);
        __syncthreads();
        if (t + 1 < n_tiles)
            tile_load(base, g.n, rows, (t + 1) * TILE, n_cols, lane, regs);
        const int lim = min(TILE, n_terms - t * TILE);
        for (int j = 0; j < lim; j++) {
            const double d = (double)tile[lane][j] - resultMean;
            sum += d * d;  // math.Pow(d, 2)
        }
        __syncthreads();
    }
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
// carried over from the previous batch), so only the two float32 running sums form a serial chain:
// lanes 0 and 1 of wave 0 run them side by side, everything else is data-parallel.
// ---------------------------------------------------------------------------------------------
constexpr int THR_CHUNK = 1024;

__global__ __launch_bounds__(256) void k_thresholds(sdr_frame_rec *__restrict__ recs, BandState *__restrict__ st,
                                                    int n_frames, int stride)
{
    __shared__ float s_in[2][THR_CHUNK];
    __shared__ float s_old[2][THR_CHUNK];
    __shared__ float s_sum[2][THR_CHUNK];
    __shared__ float s_ring[2][SDR_NOISE_WINDOW];
    __shared__ float s_carry[2];
    const int band = blockIdx.x;
    const int tid = threadIdx.x;
    BandState *s = &st[band];
    sdr_frame_rec *r = recs + (size_t)band * stride;
    const int next0 = s->next;
    const float peak_threshold = s->peak_threshold;
    if (tid < SDR_NOISE_WINDOW) {
        s_ring[0][tid] = s->nf_ring[tid];
        s_ring[1][tid] = s->dev_ring[tid];
    }
    if (tid == 0) {
        s_carry[0] = s->nf_sum;
        s_carry[1] = s->dev_sum;
    }
    __syncthreads();

    for (int base = 0; base < n_frames; base += THR_CHUNK) {
        const int cnt = min(THR_CHUNK, n_frames - base);
        for (int j = tid; j < cnt; j += blockDim.x) {
            const int f = base + j;
            s_in[0][j] = r[f].nf_in;
            s_in[1][j] = r[f].dev_in;
            if (f >= SDR_NOISE_WINDOW) {
                s_old[0][j] = r[f - SDR_NOISE_WINDOW].nf_in;
                s_old[1][j] = r[f - SDR_NOISE_WINDOW].dev_in;
            } else {
                const int slot = (next0 + f) % SDR_NOISE_WINDOW;
                s_old[0][j] = s_ring[0][slot];
                s_old[1][j] = s_ring[1][slot];
            }
        }
        __syncthreads();
        if (tid < 2) {
            float sum = s_carry[tid];
            const float *in = s_in[tid], *old = s_old[tid];
            float *out = s_sum[tid];
            for (int j = 0; j < cnt; j++) {
                sum = sum - old[j];  // v.sumForMean -= v.values[v.next]
                sum = sum + in[j];   // v.sumForMean += v.values[v.next]
                out[j] = sum;
            }
            s_carry[tid] = sum;
        }
        __syncthreads();
        for (int j = tid; j < cnt; j += blockDim.x) {
            const int f = base + j;
            const float noiseFloor = __fdiv_rn(s_sum[0][j], (float)SDR_NOISE_WINDOW);
            const float noiseDeviation = __fdiv_rn(s_sum[1][j], (float)SDR_NOISE_WINDOW);
            r[f].noise_floor = noiseFloor;
            r[f].noise_dev = noiseDeviation;
            r[f].peak_thr = peak_threshold + noiseFloor;  // rx/receiver.go:385
            r[f].listen_thr = noiseFloor + noiseDeviation;  // rx/receiver.go:394
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
    if (tid == 0) {
        s->nf_sum = s_carry[0];
        s->dev_sum = s_carry[1];
        s->next = (next0 + n_frames) % SDR_NOISE_WINDOW;
    }
}


hipError_t launch_window_means(const float *psd, double *win_mean, NoiseGeom g, int n_frames, int n_bands, int stride,
                               hipStream_t stream)
{
    hipLaunchKernelGGL(k_window_means, dim3((n_frames + TILE - 1) / TILE, g.n_windows, n_bands), dim3(64), 0, stream,
                       psd, win_mean, g, n_frames, stride);
    return hipGetLastError();
}

hipError_t launch_noise_stats(const float *psd, const double *win_mean, sdr_frame_rec *recs, NoiseGeom g, int n_frames,
                              int n_bands, int stride, hipStream_t stream)
{
    hipLaunchKernelGGL(k_noise_stats, dim3((n_frames + TILE - 1) / TILE, n_bands), dim3(64), 0, stream, psd, win_mean,
                       recs, g, n_frames, stride);
    return hipGetLastError();
}

hipError_t launch_thresholds(sdr_frame_rec *recs, BandState *st, int n_frames, int n_bands, int stride,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(k_thresholds, dim3(n_bands), dim3(256), 0, stream, recs, st, n_frames, stride);
    return hipGetLastError();
}


}  // namespace sdr
