// Certifies the fast dB path of gomath.h against the literal Go algorithm on the CPU (TEST ONLY):
//  * whenever psd_value_in_db_fast accepts, its float32 equals psd_value_in_db's, bit for bit;
//  * |y_fast - y_go| stays orders of magnitude below the acceptance guard kDbGuard;
//  * the acceptance rate is what the design assumes (the slow path must stay rare).
// usage: emu_log [n_random]   (exit code 0 = certified)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../sdrainer_amd/csrc/gomath.h"

int main(int argc, char **argv)
{
    const long n_random = argc > 1 ? atol(argv[1]) : 20000000L;
    // one table blob per block size (the exponent table folds log2 N in)
    static unsigned char blobs[15][gomath::kDbTabBytes] __attribute__((aligned(16)));
    for (int logn = 9; logn <= 14; logn++)
        gomath::build_db_tables(logn, blobs[logn]);
    std::mt19937_64 rng(12345);
    long tested = 0, accepted = 0, mismatches = 0;
    double max_dy = 0;
    auto one = [&](float psd, int logn) {
        const double inv_n2 = std::ldexp(1.0, -2 * logn);
        const float exact = gomath::psd_value_in_db(psd, inv_n2);
        float fast = 0;
        const gomath::DbTables tab = gomath::db_tables(blobs[logn]);
        const bool ok = gomath::psd_value_in_db_fast(psd, tab, &fast);
        tested++;
        if (ok) {
            accepted++;
            if (std::memcmp(&fast, &exact, sizeof fast) != 0) {
                if (mismatches < 10)
                    printf("MISMATCH psd=%a logn=%d fast=%a exact=%a\n", psd, logn, fast, exact);
                mismatches++;
            }
        }
        const double v = 20.0 * (double)psd * inv_n2;
        if (v >= 2.2250738585072014e-308 && v < INFINITY && std::fabs(psd) >= 1.17549435e-38f) {
            const double dy = std::fabs(gomath::db_fast_y(psd, tab) - 10.0 * gomath::log10(v));
            if (dy > max_dy)
                max_dy = dy;
        }
    };
    // every float32 exponent, a sweep of mantissas around interval edges of the 1024-entry table
    for (int e = 1; e < 255; e++)
        for (int k = 0; k < 1024; k += (e % 8 == 0 ? 1 : 16))
            for (int d = -2; d <= 2; d++) {
                uint32_t b = ((uint32_t)e << 23) | (((uint32_t)k << 13) + (uint32_t)d) % (1u << 23);
                float f;
                std::memcpy(&f, &b, sizeof f);
                for (int logn = 9; logn <= 14; logn++)
                    one(f, logn);
            }
    // special values
    const float specials[] = {0.f, -0.f, -1.f, -1e-40f, INFINITY, -INFINITY, NAN, -NAN, 1e-45f, 1.1e-38f, 1.17549435e-38f, 3.4e38f, 1.f, 2.f, 0.5f};
    for (float f : specials)
        for (int logn = 9; logn <= 14; logn++)
            one(f, logn);
    // random positive float32 bit patterns (uniform over exponents) and realistic PSD magnitudes
    std::uniform_int_distribution<uint32_t> anybits(1, 0x7f7fffffu);
    std::uniform_real_distribution<double> lg(-12.0, 10.0);
    for (long i = 0; i < n_random; i++) {
        uint32_t b = anybits(rng);
        float f;
        std::memcpy(&f, &b, sizeof f);
        one(f, 9 + (int)(i % 6));
        one((float)std::pow(10.0, lg(rng)), 9 + (int)((i / 6) % 6));
    }
    printf("tested %ld, accepted %ld (%.5f%%), mismatches %ld, max |y_fast - y_go| = %.3e (guard %.1e)\n", tested,
           accepted, 100.0 * accepted / tested, mismatches, max_dy, gomath::kDbGuard);
    const bool ok = mismatches == 0 && max_dy < gomath::kDbGuard / 50 && accepted > 0.99 * (tested - 100000);
    return ok ? 0 : 1;
}
