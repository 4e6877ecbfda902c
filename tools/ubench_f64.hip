// Micro-benchmark (development aid, not part of the product): dependent-chain latency of v_add_f64 /
// v_mul_f64 / v_fma-free complex butterflies, and of ds_read_b64 + v_add_f64, on one wave.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_add_chain(double *out, double x, int n, long long *cycles)
{
    double s = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            s += x;
    }
    long long t1 = clock64();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

__global__ void k_mul_chain(double *out, double x, int n, long long *cycles)
{
    double s = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            s *= x;
    }
    long long t1 = clock64();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

__global__ void k_add_indep(double *out, double x, int n, long long *cycles)
{
    double s[8];
    for (int k = 0; k < 8; k++) s[k] = out[threadIdx.x] + k;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            s[k & 7] += x;
    }
    long long t1 = clock64();
    double r = 0;
    for (int k = 0; k < 8; k++) r += s[k];
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

__global__ void k_lds_add_chain(double *out, int n, long long *cycles)
{
    __shared__ double t[64][65];
    for (int j = 0; j < 64; j++) t[threadIdx.x][j] = 1e-3 * j;
    __syncthreads();
    double s = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++)
            s += t[threadIdx.x][j];
    }
    long long t1 = clock64();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

// Chip-wide float64 issue rate: every CU full of waves, 8 independent chains per lane, alternating
// v_mul_f64 / v_add_f64 (the butterfly mix), timed with HIP events so DVFS under load is included.
__global__ __launch_bounds__(1024) void k_f64_throughput(double *out, double a, double b, int n, long long *cycles)
{
    double s[8];
    for (int k = 0; k < 8; k++) s[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            s[k] = s[k] * a;
#pragma unroll
        for (int k = 0; k < 8; k++)
            s[k] = s[k] + b;
    }
    long long t1 = clock64();
    double r = 0;
    for (int k = 0; k < 8; k++) r += s[k];
    if (r == 12345.678) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

static void throughput(double *out, long long *cyc, int blocks, int threads)
{
    const int n = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_f64_throughput, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9, n, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        long long h = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
        const double ops = (double)blocks * threads * n * 16.0;
        const double waves_per_simd = (double)threads / 64 / 4;
        printf("%4d blocks x %4d threads: %.3f ms, %.2f T f64-ops/s (lane ops), block 0: %.2f clk per wave-instruction per SIMD, "
               "shader clock %.2f GHz\n",
               blocks, threads, ms, ops / (ms * 1e-3) / 1e12, (double)h / (n * 16.0 * waves_per_simd),
               (double)h / (ms * 1e-3) / 1e9);
    }
}

// dependent-chain latency in nanoseconds (wall_clock64 ticks at 100 MHz whatever the shader clock does)
__global__ void k_add_chain_ns(double *out, double x, int n, long long *ticks)
{
    double s = out[threadIdx.x];
    long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            s += x;
    }
    long long t1 = wall_clock64();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}

int main()
{
    double *out;
    long long *cyc, h;
    hipMalloc(&out, 64 * sizeof(double));
    hipMemset(out, 0, 64 * sizeof(double));
    hipMalloc(&cyc, sizeof(long long));
    const int n = 4096;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_add_chain_ns, dim3(1), dim3(64), 0, 0, out, 1e-9, 65536, cyc);
        hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("dependent v_add_f64, one wave: %.2f ns/op (= %.1f clocks at 2.4 GHz)\n", (double)h * 10.0 / (65536.0 * 16),
               (double)h * 10.0 / (65536.0 * 16) * 2.4);
    }
    throughput(out, cyc, 256, 1024);
    throughput(out, cyc, 256, 512);
    throughput(out, cyc, 256, 256);
    throughput(out, cyc, 64, 1024);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_add_chain, dim3(1), dim3(64), 0, 0, out, 1e-9, n, cyc);
        hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("dependent v_add_f64: %.2f clk/op\n", (double)h / (n * 16));
        hipLaunchKernelGGL(k_mul_chain, dim3(1), dim3(64), 0, 0, out, 1.0000001, n, cyc);
        hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("dependent v_mul_f64: %.2f clk/op\n", (double)h / (n * 16));
        hipLaunchKernelGGL(k_add_indep, dim3(1), dim3(64), 0, 0, out, 1e-9, n, cyc);
        hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("independent v_add_f64 (8 chains): %.2f clk/op\n", (double)h / (n * 16));
        hipLaunchKernelGGL(k_lds_add_chain, dim3(1), dim3(64), 0, 0, out, 256, cyc);
        hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("ds_read_b64 + dependent v_add_f64: %.2f clk/term\n", (double)h / (256 * 64));
    }
    return 0;
}
