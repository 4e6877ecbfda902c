// CPU emulation of the FFT kernel's phase functions (sdrainer_amd/csrc/fft_f64.h), thread by thread,
// to validate the register/LDS index math and the twiddle layout without a GPU.  TEST ONLY: this is
// not a CPU fallback — nothing in the product links it.  It is compared bit-for-bit against the
// oracle's stage-by-stage radix-2 FFT (oracle/sdr_oracle.c: orc_fft_radix2).
//
// usage: emu_fft <liborc.so>     (exit code 0 = all sizes bit-identical)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../sdrainer_amd/csrc/fft_f64.h"
#include "../../sdrainer_amd/csrc/twiddles.h"

using namespace fft64;

typedef void (*orc_iq_fft_t)(int, const float *, double *, double *);
typedef void (*orc_factors_t)(int, double *, double *);

template <int LOGN, int P>
struct Passes {
    static void run(std::vector<double> &xr, std::vector<double> &xi, const cplx *tw, std::vector<double> &lds_re,
                    std::vector<double> &lds_im)
    {
        using PL = Plan<LOGN>;
        for (int t = 0; t < PL::T; t++)
            butterfly_pass<LOGN, P>(&xr[(size_t)t * PL::R], &xi[(size_t)t * PL::R], t,
                                    [tw](int c, int lo) { return tw[c + lo]; });
        if (P < PL::NPASS - 1) {
            constexpr int E = (P < PL::NPASS - 1 ? P : 0);
            if constexpr (make_reg_plan<LOGN>(E).n > 0) {
                // register exchange (v_permlane16/32_swap and ds_bpermute on the GPU): modelled wave by wave
                for (int t0 = 0; t0 < PL::T; t0 += 64) {
                    exchange_regs_wave<LOGN, E>(reinterpret_cast<double (*)[PL::R]>(&xr[(size_t)t0 * PL::R]));
                    exchange_regs_wave<LOGN, E>(reinterpret_cast<double (*)[PL::R]>(&xi[(size_t)t0 * PL::R]));
                }
            } else {
            // LDS starts each exchange poisoned, so a read of a word nobody wrote yet cannot pass
            std::fill(lds_re.begin(), lds_re.end(), std::nan(""));
            std::fill(lds_im.begin(), lds_im.end(), std::nan(""));
            // A cross-wave exchange is barrier-separated: all threads write, then all threads read.  A
            // wave-local one has no workgroup barrier in the kernel: emulate the worst schedule, every
            // wave running its write and its read before the next wave starts.
            const int group = PL::cross_wave(E) ? PL::T : 64;
            for (int t0 = 0; t0 < PL::T; t0 += group) {
                for (int t = t0; t < t0 + group; t++) {
                    exchange_write<LOGN, E>(&xr[(size_t)t * PL::R], t, lds_re.data());
                    exchange_write<LOGN, E>(&xi[(size_t)t * PL::R], t, lds_im.data());
                    // a wave-local exchange stays inside the wave's own block of words
                    if (!PL::cross_wave(E))
                        for (int s = 0; s < PL::R; s++) {
                            const int a = lds_addr<LOGN, E>(thread_part<LOGN, E>(t)) + lds_addr<LOGN, E>(slot_part<LOGN, E>(s));
                            if (a / make_addr<LOGN>(E).block != t / 64) {
                                printf("LOGN=%d exchange %d: thread %d writes outside its wave's block\n", LOGN, E, t);
                                exit(1);
                            }
                        }
                }
                for (int t = t0; t < t0 + group; t++) {
                    exchange_read<LOGN, E>(&xr[(size_t)t * PL::R], t, lds_re.data());
                    exchange_read<LOGN, E>(&xi[(size_t)t * PL::R], t, lds_im.data());
                }
            }
            }
            Passes<LOGN, (P < PL::NPASS - 1 ? P + 1 : P)>::run_next(xr, xi, tw, lds_re, lds_im);
        }
    }
    static void run_next(std::vector<double> &xr, std::vector<double> &xi, const cplx *tw, std::vector<double> &lds_re,
                         std::vector<double> &lds_im)
    {
        run(xr, xi, tw, lds_re, lds_im);
    }
};


// Bank-conflict audit of exchange E on the MI355X LDS model (MI355X_MICROARCH.md, LDS): a ds_write_b64
// is served in groups of 16 consecutive lanes over 32 four-byte banks, a ds_read_b64 in groups of 32
// lanes over 64 banks.  Returns the worst number of distinct addresses sharing a bank in one group.
template <int LOGN, int E>
static void audit(int *worst_write, int *worst_read)
{
    using PL = Plan<LOGN>;
    *worst_write = *worst_read = 0;
    for (int s = 0; s < PL::R; s++)
        for (int t0 = 0; t0 < PL::T; t0 += 16) {
            int cnt[16] = {};
            for (int t = t0; t < t0 + 16; t++) {
                const int a = lds_addr<LOGN, E>(thread_part<LOGN, E>(t)) + lds_addr<LOGN, E>(slot_part<LOGN, E>(s));
                *worst_write = std::max(*worst_write, ++cnt[a % 16]);
            }
        }
    for (int s = 0; s < PL::R; s++)
        for (int t0 = 0; t0 < PL::T; t0 += 32) {
            int cnt[32] = {};
            for (int t = t0; t < t0 + 32; t++) {
                const int a = lds_addr<LOGN, E>(thread_part<LOGN, E + 1>(t)) + lds_addr<LOGN, E>(slot_part<LOGN, E + 1>(s));
                *worst_read = std::max(*worst_read, ++cnt[a % 32]);
            }
        }
}

template <int LOGN, int E>
struct Audit {
    static int run()
    {
        int w, r, rc = 0;
        audit<LOGN, E>(&w, &r);
        if (make_reg_plan<LOGN>(E).n > 0) {
            printf("LOGN=%d exchange %d: in registers:", LOGN, E);
            for (int i = 0; i < make_reg_plan<LOGN>(E).n; i++) {
                const RegStep st = make_reg_plan<LOGN>(E).st[i];
                if (st.rot)
                    printf(" rotate lane bits (%d,%d)<->(4,5);", st.a, st.a + 1);
                else
                    printf(" swap slot bit %d<->lane bit 4, slot bit %d<->lane bit 5;", st.a, st.b);
            }
            printf(" plan %s\n", reg_plan_valid<LOGN>(E) ? "valid" : "INVALID");
            if (!reg_plan_valid<LOGN>(E))
                rc = 1;
        } else {
            printf("LOGN=%d exchange %d (%s): worst write %d-way, worst read %d-way\n", LOGN, E,
                   Plan<LOGN>::cross_wave(E) ? "cross-wave" : "wave-local", w, r);
            if (LOGN >= 10 && (w > 1 || r > 1))
                rc = 1;  // N = 512 keeps a 2-way write conflict; every other size must be conflict-free
        }
        if constexpr (E + 1 < Plan<LOGN>::NPASS - 1)
            rc |= Audit<LOGN, E + 1>::run();
        return rc;
    }
};

// Global-memory audit: 16 consecutive lanes of a pass-0 load must cover whole 32-byte runs of samples.
template <int LOGN>
static int audit_loads()
{
    using PL = Plan<LOGN>;
    if (PL::LB) {
        // layout B loads the frame straight from memory: the 64 lanes of a wave must read 64 consecutive samples
        // (512 contiguous bytes per wave instruction) for every register slot
        for (int m = 0; m < PL::R; m++)
            for (int t0 = 0; t0 < PL::T; t0 += 64)
                for (int t = t0; t < t0 + 64; t++)
                    if (input_sample<LOGN>(t, m) != input_sample<LOGN>(t0, m) + (t - t0)) {
                        printf("LOGN=%d (layout B): wave %d slot %d does not load 64 consecutive samples\n", LOGN, t0 / 64, m);
                        return 1;
                    }
        printf("LOGN=%d (layout B): every wave loads 512 contiguous bytes per slot\n", LOGN);
        return 0;
    }
    for (int m = 0; m < PL::R; m++)
        for (int t0 = 0; t0 < PL::T; t0 += 4)
            for (int t = t0; t < t0 + 4; t++)
                if (input_sample<LOGN>(t, m) != input_sample<LOGN>(t0, m) + (t - t0)) {
                    printf("LOGN=%d: lanes %d..%d of slot %d do not load consecutive samples\n", LOGN, t0, t0 + 3, m);
                    return 1;
                }
    return 0;
}

// Input staging (fft_f64.h in_granule / in_lds_byte): replay the LDS-DMA image row by row and check that every
// thread finds its 16 samples, that every 1 KB row is read from one contiguous kilobyte of the frame, and that
// no ds_read_b64 group (32 lanes, 64 four-byte banks) has a bank conflict.
template <int LOGN>
static int audit_staging()
{
    using PL = Plan<LOGN>;
    const int N = PL::N, rows = N / 128;
    std::vector<int> image((size_t)N, -1);  // image[byte / 8] = sample number stored there
    for (int r = 0; r < rows; r++)
        for (int p = 0; p < 64; p++) {
            const int g = in_granule<LOGN>(p, r);
            if (g < 0 || g >= 64) {
                printf("LOGN=%d: row %d lane %d fetches granule %d\n", LOGN, r, p, g);
                return 1;
            }
            image[(size_t)(r * 1024 + p * 16) / 8] = r * 128 + 2 * g;
            image[(size_t)(r * 1024 + p * 16) / 8 + 1] = r * 128 + 2 * g + 1;
        }
    int worst = 0;
    for (int m = 0; m < PL::R; m++)
        for (int t0 = 0; t0 < PL::T; t0 += 32) {
            int cnt[32] = {};
            for (int t = t0; t < t0 + 32; t++) {
                const int n = input_sample<LOGN>(t, m);
                const int a = in_lds_byte<LOGN>(n);
                if (a % 8 || image[(size_t)a / 8] != n) {
                    printf("LOGN=%d: thread %d slot %d reads byte %d, which holds sample %d, not %d\n", LOGN, t, m, a,
                           image[(size_t)a / 8], n);
                    return 1;
                }
                // the kernel combines the thread's slot-0 address with a per-slot constant by XOR
                const int combined = in_lds_byte<LOGN>(input_sample<LOGN>(t, 0)) ^ in_lds_byte<LOGN>(input_sample<LOGN>(0, m));
                if (combined != a) {
                    printf("LOGN=%d: thread %d slot %d: combined address %d != %d\n", LOGN, t, m, combined, a);
                    return 1;
                }
                worst = std::max(worst, ++cnt[(a / 8) % 32]);
            }
        }
    printf("LOGN=%d input staging: %d rows, worst ds_read_b64 conflict %d-way\n", LOGN, rows, worst);
    return worst > 1;
}

// Twiddle rows (fft_f64.h make_tw_perm): tw_pos_of_lo is a bijection of [0, S), the position a thread computes from
// its id agrees with it, and the lanes 0-15 of a wave read consecutive entries wherever index bits sit in lanes.
template <int LOGN, int P>
static int audit_tw_rows()
{
    using PL = Plan<LOGN>;
    if constexpr (P < PL::NPASS) {
        constexpr int S = 1 << (P * PL::LOGR);
        constexpr int G = PL::R >> PL::pass_log(P);
        std::vector<char> seen((size_t)S, 0);
        for (int lo = 0; lo < S; lo++) {
            const int p = tw_pos_of_lo<LOGN, P>(lo);
            if (p < 0 || p >= S || seen[p]) {
                printf("LOGN=%d pass %d: tw_pos_of_lo is not a bijection\n", LOGN, P);
                return 1;
            }
            seen[p] = 1;
        }
        for (int t = 0; t < PL::T; t++)
            for (int u = 0; u < G; u++) {
                const int lo = elem_index<LOGN, P>(t, u, 0) & (S - 1);
                if (tw_pos<LOGN, P>(t, u) != tw_pos_of_lo<LOGN, P>(lo)) {
                    printf("LOGN=%d pass %d: thread %d group %d computes position %d, the table has it at %d\n", LOGN, P, t, u,
                           tw_pos<LOGN, P>(t, u), tw_pos_of_lo<LOGN, P>(lo));
                    return 1;
                }
            }
        return audit_tw_rows<LOGN, P + 1>();
    }
    return 0;
}

template <int LOGN>
static int check(orc_iq_fft_t orc_fft, orc_factors_t orc_fac, unsigned seed)
{
    using PL = Plan<LOGN>;
    const int N = PL::N;
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> iq(2 * (size_t)N);
    for (auto &v : iq)
        v = nd(rng);
    // tones so intermediate magnitudes vary widely
    for (int n = 0; n < N; n++) {
        iq[2 * n] += 100.f * (float)cos(2 * M_PI * 37.0 * n / N);
        iq[2 * n + 1] += 100.f * (float)sin(2 * M_PI * 37.0 * n / N);
    }
    std::vector<double> wre, wim;
    radix2_factors(N, wre, wim);
    std::vector<double> ore((size_t)N), oim((size_t)N);
    orc_fac(N, ore.data(), oim.data());
    if (memcmp(wre.data(), ore.data(), sizeof(double) * N) || memcmp(wim.data(), oim.data(), sizeof(double) * N)) {
        printf("LOGN=%d: twiddle tables differ between product and oracle\n", LOGN);
        return 1;
    }
    std::vector<cplx> tw((size_t)PL::TW_TOTAL);
    build_pass_twiddles<LOGN>(wre.data(), wim.data(), tw.data());

    std::vector<double> xr((size_t)N), xi((size_t)N), lre((size_t)exchange_words<LOGN>()), lim((size_t)exchange_words<LOGN>());
    for (int t = 0; t < PL::T; t++)
        load_input<LOGN>(iq.data(), t, &xr[(size_t)t * PL::R], &xi[(size_t)t * PL::R]);
    Passes<LOGN, 0>::run(xr, xi, tw.data(), lre, lim);

    std::vector<double> yre((size_t)N), yim((size_t)N), seen((size_t)N, 0.0);
    for (int t = 0; t < PL::T; t++)
        for (int s = 0; s < PL::R; s++) {
            const int b = output_bin<LOGN>(t, s);
            if (b < 0 || b >= N || seen[b] != 0.0) {
                printf("LOGN=%d: output_bin not a bijection (t=%d s=%d -> %d)\n", LOGN, t, s, b);
                return 1;
            }
            seen[b] = 1.0;
            yre[b] = xr[(size_t)t * PL::R + s];
            yim[b] = xi[(size_t)t * PL::R + s];
        }
    std::vector<double> rre((size_t)N), rim((size_t)N);
    orc_fft(N, iq.data(), rre.data(), rim.data());
    long bad = 0;
    for (int i = 0; i < N; i++) {
        // bit-identical except possibly the sign of an exact zero (skipped W=1 / W=-i multiplies)
        const bool same_re = (yre[i] == rre[i]);
        const bool same_im = (yim[i] == rim[i]);
        if (!same_re || !same_im) {
            if (bad < 5)
                printf("LOGN=%d bin %d: got (%a,%a) want (%a,%a)\n", LOGN, i, yre[i], yim[i], rre[i], rim[i]);
            bad++;
        }
    }
    printf("LOGN=%d N=%d T=%d R=%d passes=%d tw=%d split=%d: %ld mismatches\n", LOGN, N, PL::T, PL::R, PL::NPASS,
           PL::TW_TOTAL, (int)PL::SPLIT, bad);
    int rc = bad != 0;
    rc |= Audit<LOGN, 0>::run();
    rc |= audit_loads<LOGN>();
    if (!PL::LB)
        rc |= audit_staging<LOGN>();
    // twiddle rows: the positions the threads of a pass read must be a permutation of the row
    rc |= audit_tw_rows<LOGN, 1>();
    return rc;
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
    auto orc_fac = (orc_factors_t)dlsym(h, "orc_radix2_factors");
    int rc = 0;
    rc |= check<9>(orc_fft, orc_fac, 1);
    rc |= check<10>(orc_fft, orc_fac, 2);
    rc |= check<11>(orc_fft, orc_fac, 3);
    rc |= check<12>(orc_fft, orc_fac, 4);
    rc |= check<13>(orc_fft, orc_fac, 5);
    rc |= check<14>(orc_fft, orc_fac, 6);
    return rc;
}
