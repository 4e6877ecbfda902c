// Micro-benchmark (development aid, not part of the product): what a float32 <-> float64 conversion costs on gfx950 next
// to a float64 add, chip-wide (every CU full, two or four waves per SIMD), and what the same widening costs when it is
// done with integer instructions on the bits.  The FFT kernel widens 64 floats per thread and frame and narrows 32.
//   usage: ubench_cvt
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                               \
        }                                                                           \
    } while (0)

// widening on the bits: sign | (exponent + 896) << 20 | mantissa >> 3, mantissa << 29 - exact for normal finite values;
// zero, subnormals, infinities and NaNs need the fix-up that `exact` adds
__device__ __forceinline__ double widen_bits(unsigned b, bool exact)
{
    const unsigned lo = b << 29;
    unsigned hi = ((b & 0x7fffffffu) >> 3) + 0x38000000u;
    hi |= b & 0x80000000u;
    if (exact) {
        const unsigned e = (b >> 23) & 0xffu;
        if (e == 0u || e == 255u)
            return (double)__uint_as_float(b);
    }
    return __hiloint2double((int)hi, (int)lo);
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_rate(const unsigned *in, double *out, int n)
{
    unsigned x[8];
    for (int k = 0; k < 8; k++)
        x[k] = in[(threadIdx.x + 64 * k) & 1023];
    double acc[8];
    for (int k = 0; k < 8; k++)
        acc[k] = 0.0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE == 0) {  // float64 add only (the yardstick)
                acc[k] += 1.5;
            } else if (MODE == 1) {  // v_cvt_f64_f32 + add
                acc[k] += (double)__uint_as_float(x[k]);
            } else if (MODE == 2) {  // integer widening + add
                acc[k] += widen_bits(x[k], false);
            } else if (MODE == 3) {  // integer widening with the special-value fix-up + add
                acc[k] += widen_bits(x[k], true);
            } else if (MODE == 4) {  // v_cvt_f32_f64 + (integer) add
                x[k] += __float_as_uint((float)acc[k]);
            }
            // keep the inputs changing so that nothing is hoisted
            if (MODE >= 1 && MODE <= 3)
                x[k] += 0x00000100u;
        }
    }
    double s = 0;
    for (int k = 0; k < 8; k++)
        s += acc[k] + (double)x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    unsigned *in;
    double *out;
    CHECK(hipMalloc(&in, 1024 * 4));
    CHECK(hipMalloc(&out, 256 * 1024 * 8));
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; i++) {
        const float f = 1e-3f * (float)(i + 1);
        memcpy(&h[i], &f, 4);
    }
    CHECK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
    const char *names[] = {"v_add_f64 alone", "v_cvt_f64_f32 + v_add_f64 + v_add_u32", "integer widening + v_add_f64 + v_add_u32",
                           "integer widening, special values handled, + v_add_f64 + v_add_u32", "v_cvt_f32_f64 + v_add_u32"};
    const int n = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int threads : {512, 1024}) {
        double base_ns = 0;
        for (int mode = 0; mode < 5; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                CHECK(hipEventRecord(e0));
                switch (mode) {
                case 0: hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(threads), 0, 0, in, out, n); break;
                case 1: hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(threads), 0, 0, in, out, n); break;
                case 2: hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(threads), 0, 0, in, out, n); break;
                case 3: hipLaunchKernelGGL(k_rate<3>, dim3(256), dim3(threads), 0, 0, in, out, n); break;
                default: hipLaunchKernelGGL(k_rate<4>, dim3(256), dim3(threads), 0, 0, in, out, n); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best)
                    best = ms;
            }
            // per wave-level group of (op + add): time x SIMDs / (waves per SIMD x iterations x 8)
            const double waves_per_simd = threads / 256.0;
            const double ns = best * 1e6 / (n * 8.0 * waves_per_simd);
            if (mode == 0)
                base_ns = ns;
            printf("256 x %4d threads  %-70s %7.3f ms  %6.2f ns per wave-level group per SIMD  (x %.2f of an add)\n", threads, names[mode], best, ns,
                   ns / base_ns);
        }
    }
    return 0;
}
