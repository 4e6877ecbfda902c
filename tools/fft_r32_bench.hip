// Development aid (not part of the product): k_fft_r32 (512 threads x 32 points, next frame prefetched into registers)
// beside the 16-point kernel k_fft_psd<14>, standalone, on the same input:
//   - the two psd arrays and the two tap arrays must be equal word for word (the 16-point kernel's bits are pinned against
//     the oracle by the GPU tests)
//   - timing of each: single launches (min / median of 30) and 100 launches back to back
//   - -DSDR_R32_PHASES=<workgroup>: the phase timeline of that workgroup's second frame, every wave
// usage: fft_r32_bench [frames [bands]]      env SDR_FFT_R32_FPW = frames per workgroup, SDR_TAP = listeners
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#ifndef SDR_TOOL_FLAGS
#define SDR_TOOL_FLAGS "(unrecorded)"
#endif
#ifndef SDR_SRC_HASH
#define SDR_SRC_HASH "(unrecorded)"
#endif
#if defined(SDR_R32_PHASES)
__device__ unsigned g_r32_phases[8][16];
#endif
#include "../sdrainer_amd/csrc/k_fft_psd.hip"
#include "../sdrainer_amd/csrc/k_fft_r32.hip"
#include "../sdrainer_amd/csrc/twiddles.h"

static unsigned long long fnv(const void *p, size_t n)
{
    const unsigned char *b = static_cast<const unsigned char *>(p);
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++)
        h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

#define CK(x)                                              \
    do {                                                   \
        hipError_t e_ = (x);                               \
        if (e_ != hipSuccess) {                            \
            printf("%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                      \
        }                                                  \
    } while (0)

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 2048;
    const int bands = argc > 2 ? atoi(argv[2]) : 1;
    const int logn = 14, N = 1 << logn;
    printf("# fft_r32_bench: %d band(s) x %d frames of %d points\n# flags: %s\n# kernel sources sha256: %s\n", bands, frames, N,
           SDR_TOOL_FLAGS, SDR_SRC_HASH);
    if (const char *e = getenv("SDR_FFT_R32_FPW"))
        printf("# SDR_FFT_R32_FPW=%s\n", e);
    std::vector<double> wre, wim;
    fft64::radix2_factors(N, wre, wim);
    // (the library's table for N = 16384 holds both kernels' twiddles; the tool launches each kernel on its own)
    std::vector<fft64::cplx> h16((size_t)sdr::twiddle_count(logn)), h32((size_t)sdr::r32_twiddle_count());
    sdr::build_twiddles(logn, wre.data(), wim.data(), h16.data());
    sdr::r32_build_twiddles(wre.data(), wim.data(), h32.data());
    setenv("SDR_FFT_R32", "0", 1);  // sdr::launch_fft below = the 16-point kernel
    fft64::cplx *tw16, *tw32;
    float *iq, *pd16, *pd32;
    const size_t total = (size_t)frames * bands;
    CK(hipMalloc(&tw16, h16.size() * sizeof(fft64::cplx)));
    CK(hipMalloc(&tw32, h32.size() * sizeof(fft64::cplx)));
    CK(hipMemcpy(tw16, h16.data(), h16.size() * sizeof(fft64::cplx), hipMemcpyHostToDevice));
    CK(hipMemcpy(tw32, h32.data(), h32.size() * sizeof(fft64::cplx), hipMemcpyHostToDevice));
    CK(hipMalloc(&iq, total * N * 8));
    CK(hipMalloc(&pd16, total * N * 4));
    CK(hipMalloc(&pd32, total * N * 4));
    CK(hipMemset(pd16, 0xff, total * N * 4));
    CK(hipMemset(pd32, 0xee, total * N * 4));
    const int n_tap = getenv("SDR_TAP") ? atoi(getenv("SDR_TAP")) : 256;
    sdr::FftTap tap16{nullptr, nullptr, n_tap, n_tap > 0 ? n_tap : 1}, tap32 = tap16;
    std::vector<int32_t> host_bins;
    float *dout16 = nullptr, *dout32 = nullptr;
    if (n_tap > 0) {
        std::vector<int32_t> bins((size_t)n_tap * bands);
        for (int b = 0; b < bands; b++)
            for (int i = 0; i < n_tap; i++)
                bins[(size_t)b * n_tap + i] = (i % 7 == 3) ? -1 : (N / 8 + i * ((3 * N / 4) / n_tap) + b) % N;
        int32_t *dbins;
        CK(hipMalloc(&dbins, bins.size() * 4));
        CK(hipMemcpy(dbins, bins.data(), bins.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dout16, total * n_tap * 4));
        CK(hipMalloc(&dout32, total * n_tap * 4));
        CK(hipMemset(dout16, 0xff, total * n_tap * 4));
        CK(hipMemset(dout32, 0xee, total * n_tap * 4));
        tap16.bins = tap32.bins = dbins;
        tap16.out = dout16;
        tap32.out = dout32;
        if (!(getenv("SDR_WIDE") && atoi(getenv("SDR_WIDE")) == 0)) {  // the wide tap (k_fft_r32 only)
            CK(hipMalloc(&tap32.wide, total * 4 * n_tap * 4));
            CK(hipMemset(tap32.wide, 0xdd, total * 4 * n_tap * 4));
            CK(hipMalloc(&tap32.used, (size_t)bands * n_tap * 4));
            CK(hipMemset(tap32.used, 0x77, (size_t)bands * n_tap * 4));
        }
        host_bins = bins;
    }
    {
        // wide dynamic range: a strong on-bin carrier, weak noise, a few zeros and subnormals
        std::vector<float> x(total * N * 2);
        unsigned s = 1;
        for (size_t i = 0; i < x.size(); i++) {
            s = s * 1664525u + 1013904223u;
            float v = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23) * 1e-3f;
            const size_t n = (i / 2) % N;
            v += 0.1f * ((i & 1) ? sinf(6.283185307f * 1237.f * n / N) : cosf(6.283185307f * 1237.f * n / N));
            if ((s >> 27) == 0)
                v = 0.f;
            if ((s >> 27) == 1)
                v = 1e-41f;
            x[i] = v;
        }
        CK(hipMemcpy(iq, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    }
    auto launch16 = [&](hipStream_t st) { return sdr::launch_fft(logn, iq, nullptr, tw16, pd16, frames, bands, frames, frames, tap16, st); };
    auto launch32 = [&](hipStream_t st) { return sdr::launch_fft_r32(iq, nullptr, tw32, pd32, frames, bands, frames, frames, tap32, st); };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(launch16(0));
    CK(launch32(0));
    CK(hipDeviceSynchronize());
    int rc = 0;
    {
        std::vector<float> a(total * N), b(total * N);
        CK(hipMemcpy(a.data(), pd16, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), pd32, b.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < a.size(); i++)
            if (memcmp(&a[i], &b[i], 4)) {
                if (!bad)
                    first = i;
                bad++;
            }
        printf("psd: 16-point hash %016llx, r32 hash %016llx, %zu of %zu words differ", fnv(a.data(), a.size() * 4),
               fnv(b.data(), b.size() * 4), bad, a.size());
        if (bad)
            printf(" (first at frame %zu bin %zu: %a vs %a)", first / N, first % N, a[first], b[first]);
        printf("\n");
        rc |= bad != 0;
        if (n_tap > 0) {
            std::vector<float> ta(total * n_tap), tb(total * n_tap);
            CK(hipMemcpy(ta.data(), dout16, ta.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(tb.data(), dout32, tb.size() * 4, hipMemcpyDeviceToHost));
            size_t tbad = 0;
            for (size_t i = 0; i < ta.size(); i++)
                tbad += memcmp(&ta[i], &tb[i], 4) != 0;
            printf("tap: %zu of %zu words differ\n", tbad, ta.size());
            rc |= tbad != 0;
            if (tap32.wide) {
                // psd at bin - 1, bin, bin + 1 of every slot in use, [frame][slot][4], and the bins they were taken at
                std::vector<float> w(total * 4 * n_tap);
                std::vector<int32_t> used((size_t)bands * n_tap);
                CK(hipMemcpy(w.data(), tap32.wide, w.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(used.data(), tap32.used, used.size() * 4, hipMemcpyDeviceToHost));
                size_t wbad = 0, ubad = 0, checked = 0;
                for (size_t i = 0; i < used.size(); i++)
                    ubad += used[i] != host_bins[i];
                for (size_t f = 0; f < total; f++) {
                    const size_t band = f / (size_t)frames;
                    for (int i = 0; i < n_tap; i++) {
                        const int bin = host_bins[band * n_tap + i];
                        if (bin < 0)
                            continue;
                        for (int c = 0; c < 3; c++) {
                            const int nb = bin + c - 1;
                            if (nb < 0 || nb >= N)
                                continue;
                            checked++;
                            wbad += memcmp(&w[(f * n_tap + i) * 4 + c], &b[f * N + nb], 4) != 0;
                        }
                    }
                }
                printf("wide tap: %zu of %zu words differ, %zu of %zu recorded bins differ\n", wbad, checked, ubad, used.size());
                rc |= wbad != 0 || ubad != 0;
            }
        }
    }
    auto time_it = [&](const char *name, auto launch) -> int {
        // warm-up: 0.3 s of launches
        float ms = 0;
        CK(hipEventRecord(e0, 0));
        while (ms < 300.f) {
            for (int i = 0; i < 50; i++)
                CK(launch(0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<float> single;
        for (int rep = 0; rep < 30; rep++) {
            CK(hipEventRecord(e0, 0));
            CK(launch(0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            single.push_back(ms);
        }
        std::sort(single.begin(), single.end());
        const int reps = 100;
        CK(hipEventRecord(e0, 0));
        for (int rep = 0; rep < reps; rep++)
            CK(launch(0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / reps;
        printf("%-8s single: min %.4f median %.4f ms | back to back: %.4f ms per launch = %.1f GS/s = %.3f of the 8 B/sample HBM roofline\n",
               name, single[0], single[15], per, (double)total * N / per / 1e6, (double)total * N * 8 / (per * 1e-3) / 8e12);
        return 0;
    };
    if (!getenv("SDR_R32_ONLY"))
        if (time_it("16-point", launch16))
            return 1;
    if (time_it("r32", launch32))
        return 1;
    if (!getenv("SDR_R32_ONLY"))
        if (time_it("16-point", launch16))
            return 1;
    if (time_it("r32", launch32))
        return 1;
#if defined(SDR_R32_PHASES)
    {
        unsigned ph[8][16];
        CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_r32_phases), sizeof ph));
        unsigned t0 = ~0u;
        for (int w = 0; w < 8; w++)
            if (ph[w][0])
                t0 = std::min(t0, ph[w][0]);
        const char *names[] = {"top", "landed", "cvt", "flushed", "widened", "pass0", "E0", "pass1", "E1", "pass2", "row", "stored"};
        printf("phase timeline of workgroup %d's second frame, shader-clock cycles since its first wave reached the frame's top (stamp = the phase named has just ended)\nwave",
               (int)SDR_R32_PHASES);
        for (int k = 0; k < sdr::r32::RS_COUNT; k++)
            printf("%8s", names[k]);
        printf("\n");
        for (int w = 0; w < 8; w++) {
            printf("%4d", w);
            for (int k = 0; k < sdr::r32::RS_COUNT; k++)
                printf("%8u", ph[w][k] ? ph[w][k] - t0 : 0u);
            printf("\n");
        }
    }
#endif
    return rc;
}
