// CPU emulation of the 32-points-per-thread FFT plan (sdrainer_amd/csrc/fft_r32.h), thread by thread: validates the
// layouts, the two LDS exchange maps, the psd row map and the three twiddle blocks without a GPU, and audits every LDS
// access pattern against the MI355X banking rules.  TEST ONLY (not a CPU fallback; nothing in the product links it).
// Compared bit for bit with the oracle's stage-by-stage radix-2 FFT (oracle/sdr_oracle.c: orc_iq_fft).
//
// usage: emu_fft_r32 <liborc.so>     (exit code 0 = bit-identical and conflict-free)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../sdrainer_amd/csrc/fft_r32.h"
#include "../../sdrainer_amd/csrc/twiddles.h"

using namespace fft32;

typedef void (*orc_iq_fft_t)(int, const float *, double *, double *);

static int audit_maps()
{
    int rc = 0;
    // injectivity + extent
    for (int E = 0; E < 2; E++) {
        const int A = E;
        const int words = E == 0 ? kE0Words : kE1Words;
        std::vector<char> seen((size_t)words, 0);
        for (int i = 0; i < N; i++) {
            const int a = map_addr(A, i);
            if (a < 0 || a >= words || seen[a]) {
                printf("E%d: map not injective / out of range at index %d -> %d\n", E, i, a);
                return 1;
            }
            seen[a] = 1;
        }
    }
    // E1 stays inside the wave's block, and the wave id means the same index bits on both sides
    for (int t = 0; t < T; t++)
        for (int s = 0; s < R; s++) {
            const int aw = map_addr(1, thread_part<1>(t) | slot_part(1, s));
            const int ar = map_addr(1, thread_part<2>(t) | slot_part(2, s));
            if (aw / kE1Block != t / 64 || ar / kE1Block != t / 64) {
                printf("E1: thread %d slot %d leaves its wave's block\n", t, s);
                return 1;
            }
        }
    // thread part + slot part == whole
    for (int t = 0; t < T; t += 7)
        for (int s = 0; s < R; s++) {
            if (map_addr_thread<0, 0>(t) + map_addr_slot<0>(0, s) != map_addr(0, thread_part<0>(t) | slot_part(0, s)) ||
                map_addr_thread<0, 1>(t) + map_addr_slot<0>(1, s) != map_addr(0, thread_part<1>(t) | slot_part(1, s)) ||
                map_addr_thread<1, 1>(t) + map_addr_slot<1>(1, s) != map_addr(1, thread_part<1>(t) | slot_part(1, s)) ||
                map_addr_thread<1, 2>(t) + map_addr_slot<1>(2, s) != map_addr(1, thread_part<2>(t) | slot_part(2, s))) {
                printf("address split broken at t=%d s=%d\n", t, s);
                return 1;
            }
        }
    // banking: ds_write_b64 groups of 16 lanes / 16 words, ds_read_b64 groups of 32 lanes / 32 words
    for (int E = 0; E < 2; E++) {
        const int A = E;
        int ww = 0, wr = 0;
        for (int s = 0; s < R; s++) {
            for (int t0 = 0; t0 < T; t0 += 16) {
                int cnt[16] = {};
                for (int t = t0; t < t0 + 16; t++) {
                    const int a = map_addr(A, (E == 0 ? thread_part<0>(t) : thread_part<1>(t)) | slot_part(E, s));
                    ww = std::max(ww, ++cnt[a % 16]);
                }
            }
            for (int t0 = 0; t0 < T; t0 += 32) {
                int cnt[32] = {};
                for (int t = t0; t < t0 + 32; t++) {
                    const int a = map_addr(A, (E == 0 ? thread_part<1>(t) : thread_part<2>(t)) | slot_part(E + 1, s));
                    wr = std::max(wr, ++cnt[a % 32]);
                }
            }
        }
        printf("r32 exchange %d (%s): worst write %d-way, worst read %d-way\n", E, E == 0 ? "cross-wave" : "wave-local", ww, wr);
        rc |= (ww > 1 || wr > 1);
    }
    // psd row: ds_write_b32 groups of 32 lanes over 32 banks; the store side's ds_read_b128 in the hardware's groups of
    // 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) over sixteen 16-byte bank quads
    {
        int ww = 0;
        for (int s = 0; s < R; s++)
            for (int t0 = 0; t0 < T; t0 += 32) {
                int cnt[32] = {};
                for (int t = t0; t < t0 + 32; t++)
                    ww = std::max(ww, ++cnt[row_word(output_bin(t, s) ^ (N / 2)) % 32]);
            }
        static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                       {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                       {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                       {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
        int wr = 0;
        for (int j = 0; j < 8; j++)
            for (int w = 0; w < NWAVES; w++)
                for (int g = 0; g < 4; g++) {
                    int cnt[16] = {};
                    for (int l = 0; l < 16; l++) {
                        const int c = grp[g][l] + 64 * w + 512 * j;  // 16-byte chunk of the row this lane stores
                        const int word = row_word(4 * c);
                        if (word % 4) {
                            printf("row: chunk %d does not stay 16-byte aligned\n", c);
                            return 1;
                        }
                        wr = std::max(wr, ++cnt[(word / 4) % 16]);
                    }
                }
        printf("r32 psd row: worst ds_write_b32 %d-way, worst ds_read_b128 %d-way\n", ww, wr);
        rc |= (ww > 1 || wr > 1);
        for (int k = 0; k < N; k += 4)
            for (int i = 1; i < 4; i++)
                if (row_word(k + i) != row_word(k) + i) {
                    printf("row: run at %d is not contiguous\n", k);
                    return 1;
                }
    }
    // prefetch loads: a wave instruction reads 64 consecutive samples
    for (int m = 0; m < R; m++)
        for (int t0 = 0; t0 < T; t0 += 64)
            for (int t = t0; t < t0 + 64; t++)
                if (thread_sample(t) + slot_sample(m) != thread_sample(t0) + slot_sample(m) + (t - t0)) {
                    printf("load: wave %d slot %d is not 512 contiguous bytes\n", t0 / 64, m);
                    return 1;
                }
    // pass-2 twiddle rows: a wave instruction reads 64 consecutive entries; pos2 is a bijection
    {
        std::vector<char> seen(1024, 0);
        for (int lo = 0; lo < 1024; lo++) {
            const int p = pos2_of_lo(lo);
            if (p < 0 || p >= 1024 || seen[p]) {
                printf("pos2_of_lo is not a bijection\n");
                return 1;
            }
            seen[p] = 1;
        }
        for (int u = 0; u < 2; u++)
            for (int t0 = 0; t0 < T; t0 += 64)
                for (int t = t0; t < t0 + 64; t++)
                    if (tw2_pos(t, u) != tw2_pos(t0, u) + (t - t0)) {
                        printf("tw2: wave %d does not read consecutive entries\n", t0 / 64);
                        return 1;
                    }
    }
    return rc;
}

static int check(orc_iq_fft_t orc_fft, unsigned seed)
{
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> iq(2 * (size_t)N);
    for (auto &v : iq)
        v = nd(rng);
    for (int n = 0; n < N; n++) {
        iq[2 * n] += 100.f * (float)cos(2 * M_PI * 37.0 * n / N);
        iq[2 * n + 1] += 100.f * (float)sin(2 * M_PI * 37.0 * n / N);
    }
    std::vector<double> wre, wim;
    fft64::radix2_factors(N, wre, wim);
    std::vector<cplx> tw((size_t)kTwTotal);
    build_twiddles(wre.data(), wim.data(), tw.data());

    std::vector<double> xr((size_t)N), xi((size_t)N), lre((size_t)kExchangeBytes / 8), lim((size_t)kExchangeBytes / 8);
    // setSamplesFromIQ (dsp/fft.go:59-69)
    for (int t = 0; t < T; t++)
        for (int m = 0; m < R; m++) {
            const int n = thread_sample(t) + slot_sample(m);
            xr[(size_t)t * R + m] = (double)iq[2 * n];
            xi[(size_t)t * R + m] = (double)iq[2 * n + 1];
        }
    const cplx *twp = tw.data();
    for (int t = 0; t < T; t++)
        run_pass<5, true>(&xr[(size_t)t * R], &xi[(size_t)t * R], [twp](int row, int) { return twp[kTw0 + row]; });
    // E0: barrier-separated, all write then all read
    std::fill(lre.begin(), lre.end(), std::nan(""));
    std::fill(lim.begin(), lim.end(), std::nan(""));
    for (int t = 0; t < T; t++)
        for (int s = 0; s < R; s++) {
            const int a = map_addr_thread<0, 0>(t) + map_addr_slot<0>(0, s);
            lre[a] = xr[(size_t)t * R + s];
            lim[a] = xi[(size_t)t * R + s];
        }
    for (int t = 0; t < T; t++)
        for (int s = 0; s < R; s++) {
            const int a = map_addr_thread<0, 1>(t) + map_addr_slot<0>(1, s);
            xr[(size_t)t * R + s] = lre[a];
            xi[(size_t)t * R + s] = lim[a];
        }
    for (int t = 0; t < T; t++) {
        const int lo = tw1_lo(t);
        run_pass<5, false, 4>(&xr[(size_t)t * R], &xi[(size_t)t * R], [twp, lo](int row, int) { return twp[kTw1 + row * 32 + lo]; });
    }
    // E1: wave-local, no workgroup barrier: emulate the worst schedule, one wave at a time
    std::fill(lre.begin(), lre.end(), std::nan(""));
    std::fill(lim.begin(), lim.end(), std::nan(""));
    for (int t0 = 0; t0 < T; t0 += 64) {
        for (int t = t0; t < t0 + 64; t++)
            for (int s = 0; s < R; s++) {
                const int a = map_addr_thread<1, 1>(t) + map_addr_slot<1>(1, s);
                lre[a] = xr[(size_t)t * R + s];
                lim[a] = xi[(size_t)t * R + s];
            }
        for (int t = t0; t < t0 + 64; t++)
            for (int s = 0; s < R; s++) {
                const int a = map_addr_thread<1, 2>(t) + map_addr_slot<1>(2, s);
                xr[(size_t)t * R + s] = lre[a];
                xi[(size_t)t * R + s] = lim[a];
            }
    }
    for (int t = 0; t < T; t++) {
        const int p0 = tw2_pos(t, 0), p1 = tw2_pos(t, 1);
        // (as the kernel runs it: twiddles requested two chunks ahead of their butterflies)
        auto tw2 = [twp, p0, p1](int row, int u) { return twp[kTw2 + row * 1024 + (u ? p1 : p0)]; };
        FirstChunks2<4, 4> first;
        first_chunks2<4, false, 4>(tw2, first);
        run_pass2_with<4, false, 4>(&xr[(size_t)t * R], &xi[(size_t)t * R], first, tw2);
    }
    std::vector<double> yre((size_t)N), yim((size_t)N);
    std::vector<char> seen((size_t)N, 0);
    for (int t = 0; t < T; t++)
        for (int s = 0; s < R; s++) {
            const int b = output_bin(t, s);
            if (b < 0 || b >= N || seen[b]) {
                printf("output_bin is not a bijection (t=%d s=%d -> %d)\n", t, s, b);
                return 1;
            }
            seen[b] = 1;
            yre[b] = xr[(size_t)t * R + s];
            yim[b] = xi[(size_t)t * R + s];
        }
    std::vector<double> rre((size_t)N), rim((size_t)N);
    orc_fft(N, iq.data(), rre.data(), rim.data());
    long bad = 0;
    for (int i = 0; i < N; i++)
        if (!(yre[i] == rre[i]) || !(yim[i] == rim[i])) {
            if (bad < 5)
                printf("bin %d: got (%a,%a) want (%a,%a)\n", i, yre[i], yim[i], rre[i], rim[i]);
            bad++;
        }
    printf("r32 N=%d T=%d R=%d tw=%d seed %u: %ld mismatches\n", N, T, R, kTwTotal, seed, bad);
    return bad != 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s liborc.so\n", argv[0]);
        return 2;
    }
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    auto orc_fft = (orc_iq_fft_t)dlsym(h, "orc_iq_fft");
    int rc = audit_maps();
    rc |= check(orc_fft, 6);
    rc |= check(orc_fft, 7);
    return rc;
}
