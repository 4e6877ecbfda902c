// CPU emulation of the FFT kernel's phase functions (sdrainer_amd/csrc/fft_f64.h), thread by thread,
// to validate the register/LDS index math and the twiddle layout without a GPU.  TEST ONLY: this is
// not a CPU fallback — nothing in the product links it.  It is compared bit-for-bit against the
// oracle's stage-by-stage radix-2 FFT (oracle/sdr_oracle.c: orc_fft_radix2).
//
// usage: emu_fft <liborc.so>     (exit code 0 = all sizes bit-identical)
#include <dlfcn.h>

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
            butterfly_pass<LOGN, P>(&xr[(size_t)t * PL::R], &xi[(size_t)t * PL::R], t, tw);
        if (P < PL::NPASS - 1) {
            // barrier-separated phases: all threads write, then all threads read
            for (int t = 0; t < PL::T; t++) {
                exchange_write<LOGN, (P < PL::NPASS - 1 ? P : 0)>(&xr[(size_t)t * PL::R], t, lds_re.data());
                exchange_write<LOGN, (P < PL::NPASS - 1 ? P : 0)>(&xi[(size_t)t * PL::R], t, lds_im.data());
            }
            for (int t = 0; t < PL::T; t++) {
                exchange_read<LOGN, (P < PL::NPASS - 1 ? P : 0)>(&xr[(size_t)t * PL::R], t, lds_re.data());
                exchange_read<LOGN, (P < PL::NPASS - 1 ? P : 0)>(&xi[(size_t)t * PL::R], t, lds_im.data());
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

    std::vector<double> xr((size_t)N), xi((size_t)N), lre((size_t)N), lim((size_t)N);
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
