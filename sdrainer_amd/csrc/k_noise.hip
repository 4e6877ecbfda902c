// k_noise.hip — dsp.FindNoiseFloor (dsp/fft.go:215-252) and the rolling thresholds of Receiver.run
// (rx/receiver.go:381-385).  Compiled with -ffp-contract=off.
//
// FindNoiseFloor is two SEQUENTIAL float64 accumulations per frame (window sums, then the variance
// about the winning window's mean): float64 addition is not associative, so to reproduce the
// reference's bits each chain keeps its order — one lane per chain.  What is parallel is everything
// around the chain: a workgroup owns 64 chains (64 consecutive frames); fifteen producer waves take
// turns fetching each chain's next 64 values (one fully coalesced 256-byte load per chain), widen /
// subtract / square them in float64 and lay them down transposed in a four-slot LDS ring guarded by
// flags; the consumer wave's lane i then only reads row i and adds — the strictly serial part is one
// ds_read_b64 + one v_add_f64 per term.
#include <hip/hip_runtime.h>

#include "../../include/sdrainer_hip.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

constexpr int TILE = 64;
constexpr int HALF = TILE / 2;
constexpr int N_PRODUCERS = 15;                        // wave 0 = consumer (the chains), waves 1..15 = producers
constexpr int CHAIN_THREADS = 64 * (1 + N_PRODUCERS);  // 1024 threads, 128 VGPRs per lane

// SLOTS = LDS tiles between producers and consumer: 4 (133 KB, one workgroup per CU) for the long
// variance chains, 2 (67 KB, two workgroups per CU) for the short window sums.
template <int SLOTS>
struct ChainShared {
    static constexpr int RING_SLOTS = SLOTS;
    double term[SLOTS][TILE][TILE + 1];  // [slot][chain][column]; row stride 65 doubles: conflict-free both ways
    double mean[TILE];                        // per chain: value subtracted before squaring (variance pass)
    int n_terms[TILE];                        // per chain: number of leading terms that count
    int ready[SLOTS][2];                      // ready[t % SLOTS][h] == t + 1  <=>  rows 32h..32h+31 of tile t are published
    int consumed;                             // tiles the consumer has finished with
};

__device__ __forceinline__ int lds_flag_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_flag_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

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
                                               int rows, int n_cols, int n_tiles, int min_terms, int p, int lane)
{
    // lane r keeps chain r's mean / length; rows read them with v_readlane (wave-uniform, no LDS traffic)
    const double mean_of_lane = sh.mean[lane];
    const int terms_of_lane = sh.n_terms[lane];
    for (int u = p; u < 2 * n_tiles; u += N_PRODUCERS) {
        const int t = u >> 1, h = u & 1;
        const unsigned col = (unsigned)(t * TILE + lane);
        // unconditional loads (a predicated load compiles to branch + load + vmcnt(0): serial round
        // trips): out-of-range rows / columns are clamped to a valid address and their values are
        // discarded below by the chain-length test (such chains have n_terms <= col)
        const unsigned ccol = min(col, (unsigned)(n_cols - 1));
        float v[HALF];
#pragma unroll
        for (int i = 0; i < HALF; i++)
            v[i] = base[(unsigned)min(h * HALF + i, rows - 1) * row_stride + ccol];
        constexpr int RING_SLOTS = Shared::RING_SLOTS;
        const int slot = t % RING_SLOTS;
        if (t >= RING_SLOTS)
            while (lds_flag_load(&sh.consumed) < t - RING_SLOTS + 1)
                __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const bool full = (t + 1) * TILE <= min_terms;  // every chain still covers the whole tile
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < HALF; i++) {
                double x = (double)v[i];
                if (VARIANCE) {
                    const double d = x - readlane_f64(mean_of_lane, i);
                    x = d * d;  // math.Pow(d, 2)
                }
                if (!full && (int)col >= readlane_i32(terms_of_lane, i))
                    x = 0.0;
                sh.term[slot][i][lane] = x;
            }
        } else {
#pragma unroll
            for (int i = 0; i < HALF; i++) {
                double x = (double)v[i];
                if (VARIANCE) {
                    const double d = x - readlane_f64(mean_of_lane, HALF + i);
                    x = d * d;
                }
                if (!full && (int)col >= readlane_i32(terms_of_lane, HALF + i))
                    x = 0.0;
                sh.term[slot][HALF + i][lane] = x;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            lds_flag_store(&sh.ready[slot][h], t + 1);
    }
}

// Consumer: lane = chain; per tile 64 strictly ordered float64 additions (ds_read_b64 + v_add_f64 each).
template <class Shared>
__device__ __forceinline__ double chain_consumer(Shared &sh, int n_tiles, int lane)
{
    constexpr int RING_SLOTS = Shared::RING_SLOTS;
    double sum = 0;
    __builtin_amdgcn_s_setprio(3);
    for (int t = 0; t < n_tiles; t++) {
        const int slot = t % RING_SLOTS;
        while (lds_flag_load(&sh.ready[slot][0]) != t + 1 || lds_flag_load(&sh.ready[slot][1]) != t + 1)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll 1
        for (int j0 = 0; j0 < TILE; j0 += 16) {
#pragma unroll
            for (int j = 0; j < 16; j++)
                sum += sh.term[slot][lane][j0 + j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            lds_flag_store(&sh.consumed, t + 1);
    }
    __builtin_amdgcn_s_setprio(0);
    return sum;
}

// Runs 64 chains (lane i of wave 0 owns chain i).  `my_terms` / `my_mean` are the consumer lane's chain
// length and mean; returns the chain's sum in the consumer lanes.  All waits are on waves of the same
// workgroup (co-resident by construction), so every spin terminates.
template <bool VARIANCE, class Shared>
__device__ __forceinline__ double chain_run(Shared &sh, const float *__restrict__ base, size_t row_stride,
                                            int rows, int n_cols, int my_terms, double my_mean)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
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
    const int n_tiles = (max_terms + TILE - 1) / TILE;
    double sum = 0;
    if (wave == 0)
        sum = chain_consumer(sh, n_tiles, lane);
    else
        chain_producer<VARIANCE>(sh, base, (unsigned)row_stride, rows, n_cols, n_tiles, min_terms, wave - 1, lane);
    __syncthreads();  // everyone is done with the ring before a caller re-initialises it
    return sum;
}

// pass 1: mean of window w of frame f = sequential float64 sum of psd[edge + w*W .. +W) / W (:239-241,:230).
// A workgroup walks `windows_per_block` windows of its 64 frames back to back: with many bands there are
// thousands of (frame group, window) chains, and one long-lived workgroup per CU beats ten short ones.
__global__ __launch_bounds__(CHAIN_THREADS) void k_window_means(const float *__restrict__ psd,
                                                                double *__restrict__ win_mean, NoiseGeom g,
                                                                int n_frames, int stride, int windows_per_block)
{
    __shared__ ChainShared<2> sh;
    const int lane = threadIdx.x & 63;
    const int f0 = blockIdx.x * TILE, band = blockIdx.z;
    const int rows = min(TILE, n_frames - f0);
    const size_t frame0 = (size_t)band * stride + f0;
    const int w_end = min(g.n_windows, (int)(blockIdx.y + 1) * windows_per_block);
    for (int w = blockIdx.y * windows_per_block; w < w_end; w++) {
        const float *base = psd + frame0 * g.n + g.edge + (size_t)w * g.window;
        const double sum = chain_run<false>(sh, base, g.n, rows, g.window, lane < rows ? g.window : 0, 0.0);
        if (threadIdx.x < rows)
            win_mean[(frame0 + lane) * 10 + w] = sum / (double)g.window;
    }
}

// pass 2: pick the minimum window in the reference's order, then the variance chain over
// psd[edge .. resultTo] (inclusive) about that mean, divided by windowSize (:244-249, App. C1);
// finally the two dB inputs of the rolling means (rx/receiver.go:383-384).
__global__ __launch_bounds__(CHAIN_THREADS) void k_noise_stats(const float *__restrict__ psd,
                                                               const double *__restrict__ win_mean,
                                                               sdr_frame_rec *__restrict__ recs, NoiseGeom g,
                                                               int n_frames, int stride)
{
    __shared__ ChainShared<4> sh;
    const int lane = threadIdx.x & 63;
    const bool consumer = threadIdx.x < TILE;
    const int f0 = blockIdx.x * TILE, band = blockIdx.y;
    const int rows = min(TILE, n_frames - f0);
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
    const double sum = chain_run<true>(sh, base, g.n, rows, g.n - g.edge, n_terms, resultMean);
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
    // enough workgroups to cover the chip (two fit per CU), but no more than needed
    const int groups = ((n_frames + TILE - 1) / TILE) * n_bands;
    int wpb = (groups * g.n_windows) / 512;
    wpb = wpb < 1 ? 1 : (wpb > g.n_windows ? g.n_windows : wpb);
    hipLaunchKernelGGL(k_window_means, dim3((n_frames + TILE - 1) / TILE, (g.n_windows + wpb - 1) / wpb, n_bands),
                       dim3(CHAIN_THREADS), 0, stream, psd, win_mean, g, n_frames, stride, wpb);
    return hipGetLastError();
}

hipError_t launch_noise_stats(const float *psd, const double *win_mean, sdr_frame_rec *recs, NoiseGeom g, int n_frames,
                              int n_bands, int stride, hipStream_t stream)
{
    hipLaunchKernelGGL(k_noise_stats, dim3((n_frames + TILE - 1) / TILE, n_bands), dim3(CHAIN_THREADS), 0, stream, psd, win_mean,
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
