// Development aid (not part of the product): per-wave phase timeline of one k_fft_project workgroup.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/fft_trace tools/fft_trace.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

// -DSDR_ABLATE=n builds time the kernel with one ingredient removed (no stamps then)
#if !defined(SDR_ABLATE)
#define SDR_FFT_TRACE 1000
#endif
__device__ unsigned long long g_fft_trace[2][16][16];
#include "../sdrainer_amd/csrc/k_fft_project.hip"
#include "../sdrainer_amd/csrc/twiddles.h"

int main()
{
    const int logn = 14, N = 1 << logn, frames = 2048;
    std::vector<double> wre, wim;
    fft64::radix2_factors(N, wre, wim);
    const size_t ntw = (size_t)sdr::twiddle_count(logn);
    std::vector<fft64::cplx> h(ntw + gomath::kLogTabSize * sizeof(gomath::LogTabEntry) / sizeof(fft64::cplx) + 1);
    sdr::build_twiddles(logn, wre.data(), wim.data(), h.data());
    gomath::build_log_table(reinterpret_cast<gomath::LogTabEntry *>(h.data() + ntw));
    fft64::cplx *tw;
    float *iq, *sp, *pd;
    hipMalloc(&tw, h.size() * sizeof(fft64::cplx));
    hipMemcpy(tw, h.data(), h.size() * sizeof(fft64::cplx), hipMemcpyHostToDevice);
    hipMalloc(&iq, (size_t)frames * N * 8);
    hipMalloc(&sp, (size_t)frames * N * 4);
    hipMalloc(&pd, (size_t)frames * N * 4);
    std::vector<float> x((size_t)frames * N * 2);
    unsigned s = 1;
    for (auto &v : x) {
        s = s * 1664525u + 1013904223u;
        v = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23);
    }
    hipMemcpy(iq, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0, 0);
        sdr::launch_fft(logn, iq, tw, sp, pd, frames, 1, frames, frames, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms\n", rep, ms);
    }
#if defined(SDR_ABLATE)
    return 0;
#endif
    unsigned long long tr2[2][16][16];
    hipMemcpyFromSymbol(tr2, HIP_SYMBOL(g_fft_trace), sizeof tr2);
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < 16; w++)
        if (tr2[0][w][0] && tr2[0][w][0] < t0)
            t0 = tr2[0][w][0];
    const char *names[14] = {"start", "loaded", "pass0", "ex0", "pass1", "ex1", "pass2", "ex2", "pass3", "-", "stored", "drained", "landed", "all"};
    const int order[] = {0, 12, 13, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11};
    const int frames_per_wg = getenv("SDR_FFT_FPW") && atoi(getenv("SDR_FFT_FPW")) == 2 ? 2 : 1;
    for (int f = 0; f < frames_per_wg; f++) {
        printf("frame %d of the workgroup (us since its first wave started; 100 MHz clock)\nwave ", f);
        for (int k : order)
            printf("%8s", names[k]);
        printf("\n");
        for (int w = 0; w < 16; w++) {
            printf("%4d ", w);
            for (int k : order) {
                const unsigned long long v = tr2[f][w][k];
                if (v >= t0)
                    printf("%8.2f", (double)(v - t0) / 100.0);
                else
                    printf("%8s", "-");
            }
            printf("\n");
        }
    }
    return 0;
}
