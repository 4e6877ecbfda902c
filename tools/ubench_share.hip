// Development aid: which instruction classes do waves on DIFFERENT SIMDs of one CU take from each other?
// A workgroup of W waves (one per SIMD for W <= 4) runs the same loop in every wave; the time of W = 1 against W = 4.
// usage: ubench_share
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(double *out, int iters, int lanes)
{
    if ((int)(threadIdx.x & 63) >= lanes)
        return;
    double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5;
    int s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), t = 3, vs = threadIdx.x, vt = 7;
    unsigned long long m = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // dependent float64 adds
#pragma unroll
            for (int j = 0; j < 16; j++)
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (MODE == 1) {  // dependent float64 fma + rsq (transcendental)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                asm volatile("v_rsq_f64 %0, %1" : "=v"(c) : "v"(a));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(c), "v"(b));
            }
        } else if (MODE == 2) {  // scalar ALU chain
#pragma unroll
            for (int j = 0; j < 16; j++)
                asm volatile("s_add_u32 %0, %0, %1\n\ts_and_b64 %2, %2, exec" : "+s"(t), "+s"(s), "+s"(m) : : "scc");
        } else if (MODE == 3) {  // compare -> saveexec -> restore (the structurizer's pattern), no taken branch
#pragma unroll
            for (int j = 0; j < 8; j++)
                asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\ts_and_saveexec_b64 %0, vcc\n\tv_add_f64 %1, %1, %2\n\ts_or_b64 exec, exec, %0"
                             : "=s"(m), "+v"(a) : "v"(b) : "vcc", "scc");
        } else if (MODE == 4) {  // taken branches
#pragma unroll
            for (int j = 0; j < 8; j++)
                asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n1:\n\tv_add_f64 %1, %1, %2" : "+s"(t), "+v"(a) : "v"(b) : "scc");
        } else if (MODE == 5) {  // v_cndmask / integer VALU chain
#pragma unroll
            for (int j = 0; j < 16; j++)
                asm volatile("v_add_u32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(vs) : "v"(vt) : "vcc");
        } else if (MODE == 6) {  // float64 division as the compiler expands it
            a = b / (a + 1.5) + a;
            asm volatile("" : "+v"(a));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + s + t + vs + (double)m + c;
}

template <int MODE>
int run(const char *name, double *d)
{
    const int iters = 20000;
    const int waves[4] = {1, 2, 4, 8}, lanes[5] = {1, 4, 16, 32, 64};
    printf("%s: ms for 20000 iterations; rows = active lanes per wave, columns = waves per workgroup (1 workgroup per CU, 8 CUs)\n", name);
    for (int l = 0; l < 5; l++) {
        printf("  %2d lanes:", lanes[l]);
        for (int w = 0; w < 4; w++) {
            hipEvent_t e0, e1;
            float ms;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k<MODE>, dim3(8), dim3(64 * waves[w]), 0, 0, d, 100, lanes[l]);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<MODE>, dim3(8), dim3(64 * waves[w]), 0, 0, d, iters, lanes[l]);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %d waves %.3f", waves[w], ms);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

int main()
{
    double *d;
    CHECK(hipMalloc(&d, 8 * 1024 * sizeof(double)));
    run<0>("dependent v_add_f64 (16 per iteration)", d);
    run<5>("v_add_u32 + v_cndmask chain (16 pairs per iteration)", d);
    run<2>("scalar ALU chain", d);
    run<3>("v_cmp -> s_and_saveexec -> v_add_f64 -> s_or exec (8 per iteration)", d);
    run<6>("float64 division (compiler expansion)", d);
    return 0;
}
