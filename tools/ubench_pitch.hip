// Development aid: does a 64 KB row pitch (N = 16384 floats) camp on a few HBM channels when a workgroup
// walks 64 rows column block by column block, as the noise-floor producers do?
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void k_rows(const float *__restrict__ base, size_t pitch, int n_cols, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *rows = base + (size_t)blockIdx.x * 64 * pitch;
    float acc = 0;
    for (int u = wave; u < 2 * (n_cols / 64); u += 16) {
        const int t = u >> 1, h = u & 1;
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; i++)
            v[i] = rows[(size_t)(h * 32 + i) * pitch + t * 64 + lane];
#pragma unroll
        for (int i = 0; i < 32; i++)
            acc += v[i];
    }
    if (acc == 123.456f)
        out[0] = acc;
}

int main()
{
    const int n_cols = 11904, frames = 2048;
    float *buf, *out;
    const size_t max_pitch = 16384 + 1024;
    hipMalloc(&buf, (size_t)frames * max_pitch * 4);
    hipMemset(buf, 0, (size_t)frames * max_pitch * 4);
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (size_t pitch : {(size_t)16384, (size_t)16384 + 32, (size_t)16384 + 64, (size_t)16384 + 256, (size_t)16384 + 1024}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_rows, dim3(frames / 64), dim3(1024), 0, 0, buf, pitch, n_cols, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best)
                best = ms;
        }
        printf("row pitch %6zu floats: %.3f ms, %.2f TB/s\n", pitch, best, (double)frames * n_cols * 4 / (best * 1e-3) / 1e12);
    }
    return 0;
}
