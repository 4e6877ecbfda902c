// Development aid: where do the waves of small workgroups land?  usage: hwid <waves per workgroup> <workgroups> [vgprs]
// Prints, per workgroup, the (XCC, SE, CU, SIMD) of each of its waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(unsigned *out, int spin)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID (id 4), all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | (0 << 6) | 20);  // HW_REG_XCC_ID bits 3:0
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 4, wgs = argc > 2 ? atoi(argv[2]) : 16;
    unsigned *d;
    hipMalloc(&d, (size_t)waves * wgs * 8);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), 0, 0, d, 100000);
    std::vector<unsigned> h((size_t)waves * wgs * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int g = 0; g < wgs; g++) {
        printf("wg %3d:", g);
        for (int w = 0; w < waves; w++) {
            const unsigned hw = h[(g * waves + w) * 2], xcc = h[(g * waves + w) * 2 + 1] & 15;
            // HW_ID: wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13
            printf("  x%u se%u sh%u cu%2u simd%u w%u", xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
        }
        printf("\n");
    }
    return 0;
}
