// Development aid (not part of the product): per-wave phase timeline of one k_fft_psd workgroup.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/fft_trace tools/fft_trace.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

// -DSDR_ABLATE=n builds time the kernel with one ingredient removed (no stamps then)
#if !defined(SDR_ABLATE)
#define SDR_FFT_TRACE 100
#endif
__device__ unsigned long long g_fft_trace[2][16][16];
#if defined(SDR_FFT_CLOCK)
namespace sdr { __device__ unsigned long long g_fft_clock[2]; __device__ unsigned long long g_fft_wg[2048][4]; }
#endif
#include "../sdrainer_amd/csrc/k_fft_psd.hip"
#include "../sdrainer_amd/csrc/twiddles.h"

int main(int argc, char **argv)
{
    const int logn = 14, N = 1 << logn, frames = argc > 1 ? atoi(argv[1]) : 2048;
    std::vector<double> wre, wim;
    fft64::radix2_factors(N, wre, wim);
    const size_t ntw = (size_t)sdr::twiddle_count(logn);
    std::vector<fft64::cplx> h(ntw);
    sdr::build_twiddles(logn, wre.data(), wim.data(), h.data());
    fft64::cplx *tw;
    float *iq, *pd;
    hipMalloc(&tw, h.size() * sizeof(fft64::cplx));
    hipMemcpy(tw, h.data(), h.size() * sizeof(fft64::cplx), hipMemcpyHostToDevice);
    hipMalloc(&iq, (size_t)frames * N * 8);
    hipMalloc(&pd, (size_t)frames * N * 4);
    // the tap: SDR_TAP listeners (default 256) on evenly spread bins
    const int n_tap = getenv("SDR_TAP") ? atoi(getenv("SDR_TAP")) : 256;
    sdr::FftTap tap{nullptr, nullptr, n_tap, n_tap > 0 ? n_tap : 1};
    if (n_tap > 0) {
        std::vector<int32_t> bins(n_tap);
        for (int i = 0; i < n_tap; i++)
            bins[i] = 2300 + i * 40;
        int32_t *dbins;
        hipMalloc(&dbins, n_tap * 4);
        hipMemcpy(dbins, bins.data(), n_tap * 4, hipMemcpyHostToDevice);
        float *dout;
        hipMalloc(&dout, (size_t)frames * n_tap * 4);
        tap.bins = dbins;
        tap.out = dout;
    }
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
    float best = 1e9f;
    for (int rep = 0; rep < 8; rep++) {
        hipEventRecord(e0, 0);
        sdr::launch_fft(logn, iq, nullptr, tw, pd, frames, 1, frames, frames, tap, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best)
            best = ms;
    }
    printf("single launch (best of 6): %.4f ms\n", best);
    if (getenv("SDR_TOOL_SHORT"))
        return 0;
#if defined(SDR_FFT_CLOCK)
    {
        // 2 s of back-to-back launches, then the shader clock over one workgroup's lifetime
        for (int rep = 0; rep < 8000; rep++)
            sdr::launch_fft(logn, iq, nullptr, tw, pd, frames, 1, frames, frames, tap, 0);
        hipDeviceSynchronize();
        unsigned long long ck[2];
        hipMemcpyFromSymbol(ck, HIP_SYMBOL(sdr::g_fft_clock), sizeof ck);
        printf("in-kernel clock: %llu shader cycles in %llu ticks of 10 ns = %.3f GHz (workgroup lifetime %.2f us)\n", ck[0], ck[1],
               (double)ck[0] / (double)ck[1] / 10.0, (double)ck[1] / 100.0);
        // one more launch from an idle chip, then every workgroup's span
        static unsigned long long wg[2048][4];
        memset(wg, 0, sizeof wg);
        hipMemcpyToSymbol(HIP_SYMBOL(sdr::g_fft_wg), wg, sizeof wg);
        sdr::launch_fft(logn, iq, nullptr, tw, pd, frames, 1, frames, frames, tap, 0);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(wg, HIP_SYMBOL(sdr::g_fft_wg), sizeof wg);
        const int fpw = getenv("SDR_FFT_FPW") ? atoi(getenv("SDR_FFT_FPW")) : 1;
        const int nwg = (frames + fpw - 1) / fpw;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nwg; i++) {
            if (wg[i][0] < t0) t0 = wg[i][0];
            if (wg[i][2] > t1) t1 = wg[i][2];
        }
        printf("launch span (first start -> last end): %.2f us, %d workgroups\n", (double)(t1 - t0) / 100.0, nwg);
        std::vector<double> life, skew;
        for (int i = 0; i < nwg; i++) {
            life.push_back((double)(wg[i][2] - wg[i][0]) / 100.0);
            skew.push_back((double)(wg[i][2] - wg[i][1]) / 100.0);
        }
        std::vector<double> l2 = life, s2 = skew;
        std::sort(l2.begin(), l2.end());
        std::sort(s2.begin(), s2.end());
        printf("workgroup lifetime us: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f ; last wave ends after wave 0 by median %.2f max %.2f us\n",
               l2[0], l2[nwg / 10], l2[nwg / 2], l2[nwg * 9 / 10], l2[nwg - 1], s2[nwg / 2], s2[nwg - 1]);
        // per (xcc, se, sh, cu): chain of workgroups -> gaps between one's end and the next one's start
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
            printf("%zu distinct CUs; gap between a workgroup's last wave ending and the next one starting on the same CU: min %.2f median %.2f p90 %.2f max %.2f us\n",
                   by_cu.size(), gaps[0], gaps[gaps.size() / 2], gaps[gaps.size() * 9 / 10], gaps.back());
        else
            printf("%zu distinct CUs\n", by_cu.size());
        printf("last end per XCC (us):");
        for (double e : per_xcc_end)
            printf(" %.1f", e);
        printf("\nfirst-generation starts (us): ");
        std::vector<double> st;
        for (int i = 0; i < nwg; i++)
            st.push_back((double)(wg[i][0] - t0) / 100.0);
        std::sort(st.begin(), st.end());
        printf("p0 %.2f p50(first 256) %.2f p100(first 256) %.2f\n", st[0], st[std::min(nwg, 256) / 2], st[std::min(nwg, 256) - 1]);
    }
#endif
    // launches back to back: on one stream (a barrier between kernels) and alternating between two streams
    // (the next launch's workgroups fill the CUs the previous one's tail leaves idle)
    hipStream_t st[2];
    hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
    for (int ns = 1; ns <= 2; ns++) {
        const int reps = 40;
        hipDeviceSynchronize();
        hipEventRecord(e0, st[0]);
        if (ns == 2)
            hipStreamWaitEvent(st[1], e0, 0);
        for (int rep = 0; rep < reps; rep++)
            sdr::launch_fft(logn, iq, nullptr, tw, pd, frames, 1, frames, frames, tap, st[rep % ns]);
        hipEvent_t ej;
        hipEventCreate(&ej);
        if (ns == 2) {
            hipEventRecord(ej, st[1]);
            hipStreamWaitEvent(st[0], ej, 0);
        }
        hipEventRecord(e1, st[0]);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%d launches on %d stream(s): %.4f ms per launch\n", reps, ns, ms / reps);
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
    const int frames_per_wg = getenv("SDR_FFT_FPW") && atoi(getenv("SDR_FFT_FPW")) >= 2 ? 2 : 1;
    if (getenv("SDR_TRACE_QUIET"))
        return 0;
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
