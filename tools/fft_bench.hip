// Development aid (not part of the product): the production k_fft_psd, standalone.
//   - timing: single launches (min / median of 30, each bracketed by HIP events) and 100 launches back to back
//   - a hash of the psd and tap outputs for a fixed pseudo-random input: variants of the kernel must print the hash
//     of the production build (whose bits the GPU tests pin against the oracle)
//   - -DSDR_FFT_PHASES=<workgroup>: the phase timeline of one workgroup, every wave (stamps held in SGPRs)
//   - -DSDR_FFT_CLOCK: per-workgroup spans over one launch and the in-kernel clock
//   - -DSDR_ABLATE=n: timing-only builds with one ingredient removed
// Built by tools/build_tools.sh, which passes the flags and the hash of the kernel sources in as strings: every
// output starts with them, so a profile file says what it was measured on.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#ifndef SDR_TOOL_FLAGS
#define SDR_TOOL_FLAGS "(unrecorded)"
#endif
#ifndef SDR_SRC_HASH
#define SDR_SRC_HASH "(unrecorded)"
#endif
#if defined(SDR_FFT_PHASES)
__device__ unsigned long long g_fft_phases[16][16];
#endif
#if defined(SDR_FFT_CLOCK)
namespace sdr {
__device__ unsigned long long g_fft_clock[2];
__device__ unsigned long long g_fft_wg[2048][4];
}  // namespace sdr
#endif
#include "../sdrainer_amd/csrc/k_fft_psd.hip"
#include "../sdrainer_amd/csrc/k_fft_r32.hip"  // (launch_fft routes N = 16384 to it; SDR_FFT_R32=0 keeps the 16-point kernel)
#include "../sdrainer_amd/csrc/twiddles.h"

static unsigned long long fnv(const void *p, size_t n)
{
    const unsigned char *b = static_cast<const unsigned char *>(p);
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++)
        h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                             \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 2048;
    const int logn = argc > 2 ? atoi(argv[2]) : 14;
    const int bands = argc > 3 ? atoi(argv[3]) : 1;
    const int N = 1 << logn;
    printf("# fft_bench: k_fft_psd standalone, %d band(s) x %d frames of %d points\n# flags: %s\n# kernel sources sha256: %s\n",
           bands, frames, N, SDR_TOOL_FLAGS, SDR_SRC_HASH);
    if (const char *e = getenv("SDR_FFT_FPW"))
        printf("# SDR_FFT_FPW=%s\n", e);
    std::vector<double> wre, wim;
    fft64::radix2_factors(N, wre, wim);
    const size_t ntw = (size_t)sdr::twiddle_count(logn);
    std::vector<fft64::cplx> h(ntw);
    sdr::build_twiddles(logn, wre.data(), wim.data(), h.data());
    fft64::cplx *tw;
    float *iq, *pd;
    const size_t total = (size_t)frames * bands;
    CK(hipMalloc(&tw, h.size() * sizeof(fft64::cplx)));
    CK(hipMemcpy(tw, h.data(), h.size() * sizeof(fft64::cplx), hipMemcpyHostToDevice));
    CK(hipMalloc(&iq, total * N * 8));
    CK(hipMalloc(&pd, total * N * 4));
    const int n_tap = getenv("SDR_TAP") ? atoi(getenv("SDR_TAP")) : 256;
    sdr::FftTap tap{nullptr, nullptr, n_tap, n_tap > 0 ? n_tap : 1};
    float *dout = nullptr;
    if (n_tap > 0) {
        std::vector<int32_t> bins((size_t)n_tap * bands);
        for (int b = 0; b < bands; b++)
            for (int i = 0; i < n_tap; i++)
                bins[(size_t)b * n_tap + i] = (N / 8 + i * ((3 * N / 4) / n_tap) + b) % N;
        int32_t *dbins;
        CK(hipMalloc(&dbins, bins.size() * 4));
        CK(hipMemcpy(dbins, bins.data(), bins.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dout, total * n_tap * 4));
        tap.bins = dbins;
        tap.out = dout;
    }
    {
        std::vector<float> x(total * N * 2);
        unsigned s = 1;
        for (auto &v : x) {
            s = s * 1664525u + 1013904223u;
            v = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23);
        }
        CK(hipMemcpy(iq, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    }
    auto launch = [&](hipStream_t st) { return sdr::launch_fft(logn, iq, nullptr, tw, pd, frames, bands, frames, frames, tap, st); };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // warm-up: half a second of launches (the clocks leave their idle state)
    {
        CK(launch(0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        float ms = 0;
        int n = 0;
        while (ms < 500.f) {
            for (int i = 0; i < 50; i++, n++)
                CK(launch(0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
    }
#if !defined(SDR_ABLATE)
    {
        std::vector<float> out(total * N);
        CK(hipMemcpy(out.data(), pd, out.size() * 4, hipMemcpyDeviceToHost));
        printf("psd hash %016llx", fnv(out.data(), out.size() * 4));
        if (const char *dump = getenv("SDR_FB_DUMP")) {  // the psd array as raw float32 (variants are compared word by word)
            if (FILE *f = fopen(dump, "wb")) {
                fwrite(out.data(), 4, out.size(), f);
                fclose(f);
            }
        }
        if (dout) {
            std::vector<float> t(total * n_tap);
            CK(hipMemcpy(t.data(), dout, t.size() * 4, hipMemcpyDeviceToHost));
            printf("  tap hash %016llx", fnv(t.data(), t.size() * 4));
        }
        printf("\n");
    }
#endif
    std::vector<float> single;
    for (int rep = 0; rep < 30; rep++) {
        CK(hipEventRecord(e0, 0));
        CK(launch(0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        single.push_back(ms);
    }
    std::sort(single.begin(), single.end());
    printf("single launch: min %.4f ms  median %.4f ms  (30 launches, events around each)\n", single[0], single[15]);
    {
        const int reps = 100;
        CK(hipEventRecord(e0, 0));
        for (int rep = 0; rep < reps; rep++)
            CK(launch(0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / reps;
        printf("back to back: %.4f ms per launch (%d launches on one stream) = %.1f GS/s = %.3f of the 8 B/sample HBM roofline\n", per, reps,
               (double)total * N / per / 1e6, (double)total * N * 8 / (per * 1e-3) / 8e12);
    }
    if (getenv("SDR_TOOL_SHORT"))
        return 0;
#if defined(SDR_FFT_CLOCK)
    {
        unsigned long long ck[2];
        CK(hipMemcpyFromSymbol(ck, HIP_SYMBOL(sdr::g_fft_clock), sizeof ck));
        printf("in-kernel clock: %llu shader cycles in %llu ticks of 10 ns = %.3f GHz (workgroup 100's lifetime %.2f us)\n", ck[0], ck[1],
               (double)ck[0] / (double)ck[1] / 10.0, (double)ck[1] / 100.0);
        static unsigned long long wg[2048][4];
        memset(wg, 0, sizeof wg);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyToSymbol(HIP_SYMBOL(sdr::g_fft_wg), wg, sizeof wg));
        CK(launch(0));
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(wg, HIP_SYMBOL(sdr::g_fft_wg), sizeof wg));
        const int fpw = getenv("SDR_FFT_FPW") ? atoi(getenv("SDR_FFT_FPW")) : 1;
        const int nwg = std::min(2048, (frames + fpw - 1) / fpw);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nwg; i++) {
            t0 = std::min(t0, wg[i][0]);
            t1 = std::max(t1, wg[i][2]);
        }
        printf("launch span (first workgroup's start -> last one's end): %.2f us, %d workgroups\n", (double)(t1 - t0) / 100.0, nwg);
        std::vector<double> life;
        for (int i = 0; i < nwg; i++)
            life.push_back((double)(wg[i][2] - wg[i][0]) / 100.0);
        std::sort(life.begin(), life.end());
        printf("workgroup lifetime us: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f\n", life[0], life[nwg / 10], life[nwg / 2],
               life[nwg * 9 / 10], life[nwg - 1]);
        std::map<unsigned long long, std::vector<int>> by_cu;
        for (int i = 0; i < nwg; i++)
            by_cu[((wg[i][3] >> 32) << 16) | (wg[i][3] & 0xff00)].push_back(i);
        std::vector<double> gaps, per_xcc_end(8, 0.0);
        for (auto &kv : by_cu) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end(), [&](int a, int b) { return wg[a][0] < wg[b][0]; });
            for (size_t k = 1; k < v.size(); k++)
                gaps.push_back(((double)wg[v[k]][0] - (double)wg[v[k - 1]][2]) / 100.0);
            const int xcc = (int)(kv.first >> 16) & 7;
            per_xcc_end[xcc] = std::max(per_xcc_end[xcc], (double)(wg[v.back()][2] - t0) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end());
        if (!gaps.empty())
            printf("%zu distinct CUs; gap between a workgroup's end and the next one's start on its CU: min %.2f median %.2f p90 %.2f max %.2f us\n",
                   by_cu.size(), gaps[0], gaps[gaps.size() / 2], gaps[gaps.size() * 9 / 10], gaps.back());
        printf("last end per XCC (us):");
        for (double e : per_xcc_end)
            printf(" %.1f", e);
        printf("\n");
    }
#endif
#if defined(SDR_FFT_PHASES)
    {
        // one launch from a drained chip would show the first generation (every CU starting at once); the stamped
        // workgroup sits in the middle of the grid, so the back-to-back run above has left its steady-state timeline
        unsigned long long ph[16][16];
        CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_fft_phases), sizeof ph));
        const int waves = std::min(16, N / 16 / 64 > 0 ? N / 16 / 64 : 1);
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < waves; w++)
            if (ph[w][0])
                t0 = std::min(t0, ph[w][0]);
        const char *names[16] = {"start", "loaded", "pass0", "ex0", "pass1", "ex1", "pass2", "ex2", "pass3", "-", "stored", "end", "landed", "all", "-", "-"};
        const int order[] = {0, 12, 13, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11};
        printf("phase timeline of workgroup %d, shader-clock cycles since its first wave started (stamp = the phase named has just ended)\nwave", (int)SDR_FFT_PHASES);
        for (int k : order)
            printf("%8s", names[k]);
        printf("\n");
        for (int w = 0; w < waves; w++) {
            printf("%4d", w);
            for (int k : order) {
                if (ph[w][k] >= t0 && ph[w][k])
                    printf("%8llu", ph[w][k] - t0);
                else
                    printf("%8s", "-");
            }
            printf("\n");
        }
        // durations: median over the waves of each phase
        printf("median over waves, cycles per phase:");
        int prev = 0;
        for (size_t i = 1; i < sizeof(order) / sizeof(order[0]); i++) {
            std::vector<long long> d;
            for (int w = 0; w < waves; w++)
                if (ph[w][order[i]] && ph[w][order[prev]])
                    d.push_back((long long)(ph[w][order[i]] - ph[w][order[prev]]));
            if (d.empty())
                continue;
            std::sort(d.begin(), d.end());
            printf("  %s %lld", names[order[i]], d[d.size() / 2]);
            prev = (int)i;
        }
        printf("\n");
    }
#endif
    return 0;
}
