// Development aid: how long does a 1024-thread workgroup (one per CU, as k_fft_psd at N=16384) take
// to pull its 128 KB frame into registers, by load pattern?  Grid 2048, 129 KB LDS to force 1 WG/CU.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k_load(const float2 *__restrict__ x, float *__restrict__ out, int spin)
{
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    const float2 *f = x + (size_t)blockIdx.x * 16384;
    float acc = 0;
    if (MODE == 0) {  // slot m <- f[m * 1024 + t]: 512 contiguous bytes per wave instruction
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const float2 v = f[m * 1024 + t];
            acc += v.x * v.y;
        }
    } else if (MODE == 1) {  // 32-byte runs: lanes 0-3 consecutive, then stride 64 samples; waves interleave
        const int lane = t & 63, w = t >> 6;
        const int base = (lane & 3) | (w << 2) | ((lane >> 2) << 6);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const float2 v = f[m * 1024 + base];
            acc += v.x * v.y;
        }
    } else if (MODE == 2) {  // 16 bytes per lane, 1 KB contiguous per wave instruction (8 instructions)
        const float4 *g = reinterpret_cast<const float4 *>(f);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const float4 v = g[m * 1024 + t];
            acc += v.x * v.y + v.z * v.w;
        }
    } else if (MODE == 3) {  // as 0, nontemporal
#pragma unroll
        for (int m = 0; m < 16; m++) {
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(&f[m * 1024 + t]));
            acc += v.x * v.y;
        }
    }
    // stand-in for the compute phase so workgroups overlap as in the real kernel
    double s = acc;
    for (int i = 0; i < spin; i++)
        s = s * 1.0000001 + 1e-9;
    if (s == 123.456)
        lds[t] = s;
    if (t == 0 || s == 123.456)
        out[blockIdx.x] = (float)s;
}

template <int MODE>
static void run(const char *name, const float2 *x, float *out, int spin)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_load<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 129 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_load<MODE>, dim3(2048), dim3(1024), 129 * 1024, 0, x, out, spin);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best)
            best = ms;
    }
    printf("%-44s spin %5d: %.3f ms  (%.2f us per workgroup slot, %.2f TB/s)\n", name, spin, best, best * 1e3 / 8,
           2048.0 * 131072 / (best * 1e-3) / 1e12);
}

int main()
{
    float2 *x;
    float *out;
    hipMalloc(&x, (size_t)2048 * 16384 * 8);
    hipMemset(x, 0, (size_t)2048 * 16384 * 8);
    hipMalloc(&out, 2048 * 4);
    for (int spin : {0, 2000}) {
        run<0>("512 B per wave instruction (dwordx2)", x, out, spin);
        run<1>("32 B runs, 16 per wave instruction (dwordx2)", x, out, spin);
        run<2>("1 KB per wave instruction (dwordx4)", x, out, spin);
        run<3>("512 B per wave instruction, nontemporal", x, out, spin);
    }
    return 0;
}
