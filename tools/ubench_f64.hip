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

int main()
{
    double *out;
    long long *cyc, h;
    hipMalloc(&out, 64 * sizeof(double));
    hipMemset(out, 0, 64 * sizeof(double));
    hipMalloc(&cyc, sizeof(long long));
    const int n = 4096;
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
