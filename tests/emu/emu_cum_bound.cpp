// CPU check of the cumulation bound (sdrainer_amd/csrc/gomath.h cum_bound_*; k_peaks.hip k_cum_bound): for float32 psd
// values of every exponent - dense sweeps, random mantissas, the edges (zero, subnormals, the smallest and largest
// normals, values just below and above powers of two, infinity, NaN) - and every block size,
//   1. per term:   fl32(fl32(10 log10(20 psd / N^2)) + 120)  <=  a128 units(psd) + per_frame   (strictly below, with room)
//   2. per column: the ordered float32 sum of 100 terms on top of a carry  <=  cum_bound(carry, sum of units, 100)
//                  for random columns (noise-like, carrier-like, wide-range, with specials mixed in) and every split of
//                  the 100 frames into a carried part and a new part;
//   3. monotonicity of what FindPeaks does with it: fl32(bound / 100) > thr is true whenever fl32(exact / 100) > thr is.
// Built with -ffp-contract=off (the terms are the literal Go arithmetic of gomath.h).  Prints the smallest margins seen.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../sdrainer_amd/csrc/gomath.h"

static float term(float psd, double inv_n2) { return gomath::psd_value_in_db(psd, inv_n2) + 120.0f; }
static float from_bits(uint32_t b)
{
    float f;
    memcpy(&f, &b, 4);
    return f;
}

int main()
{
    long long bad = 0, terms = 0, columns = 0;
    double min_term_margin = 1e300, min_col_margin = 1e300;
    std::mt19937_64 rng(20261004);
    for (int logn = 9; logn <= 14; logn++) {
        const int n = 1 << logn;
        const double inv_n2 = 1.0 / ((double)n * (double)n);
        double a128, per_frame;
        gomath::cum_bound_constants(n, &a128, &per_frame);
        auto check_term = [&](uint32_t bits) {
            const float psd = from_bits(bits);
            const float s = term(psd, inv_n2);
            const double ub = gomath::cum_bound_special(psd) ? INFINITY : a128 * (double)gomath::cum_bound_units(psd) + per_frame;
            if (((bits >> 16) + 0x8080u) >> 16 != (gomath::cum_bound_special(psd) ? 1u : 0u))  // the kernel's four-at-a-time form of the test
                bad++;
            terms++;
            if (std::isnan(s))
                return;  // (NaN psd: no ordering to check; its units make the column a candidate, checked below)
            if (!((double)s <= ub)) {
                if (bad++ < 10)
                    printf("term: N %d psd bits %08x  s %.9g  bound %.9g\n", n, bits, (double)s, ub);
            } else if (std::isfinite(s)) {
                min_term_margin = std::min(min_term_margin, ub - (double)s);
            }
        };
        // every exponent: 4096 evenly spaced mantissas, the two ends of the binade, random mantissas
        for (uint32_t e = 0; e <= 255; e++) {
            for (uint32_t k = 0; k < 4096; k++)
                check_term((e << 23) | (k << 11));
            check_term((e << 23) | 0x7fffffu);
            check_term((e << 23) | 1u);
            for (int k = 0; k < 4096; k++)
                check_term((e << 23) | (uint32_t)(rng() & 0x7fffffu));
            // mantissas where log2(m) - (m - 1) peaks (m = 1 / ln 2 = 1.4427)
            for (int k = -2048; k < 2048; k++)
                check_term((e << 23) | (uint32_t)(0x38aa3b + k * 64));
        }
        // columns
        std::normal_distribution<double> gauss(0.0, 1.0);
        for (int col = 0; col < 20000; col++) {
            float psd[100];
            const int kind = col % 5;
            const double level = std::pow(10.0, -9.0 + 12.0 * (double)(rng() % 1000) / 1000.0);
            for (int f = 0; f < 100; f++) {
                double v;
                if (kind == 0) {  // noise: exponential distribution about a level
                    const double a = gauss(rng), b2 = gauss(rng);
                    v = level * 0.5 * (a * a + b2 * b2);
                } else if (kind == 1) {  // keyed carrier over noise
                    const double a = gauss(rng), b2 = gauss(rng);
                    v = ((f / 7) & 1) ? level * 1e7 : level * 0.5 * (a * a + b2 * b2);
                } else if (kind == 2) {  // any float32 at all (positive)
                    v = 0;
                    psd[f] = from_bits((uint32_t)(rng() & 0x7fffffffu));  // (infinities and NaNs among them, about 1 in 256)
                    continue;
                } else if (kind == 3) {  // tiny values, zeros and subnormals mixed in
                    v = (f % 9 == 0) ? 0.0 : (f % 9 == 1) ? 1e-42 : level * 1e-30;
                } else {  // huge
                    v = level * 1e28;
                }
                psd[f] = (float)v;
            }
            const int count0 = (int)(rng() % 100);  // frames already in the carry
            float carry = 0.f;
            for (int f = 0; f < count0; f++)
                carry += term(psd[f], inv_n2);
            float exact = carry;
            uint32_t units = 0;
            bool special = false;
            for (int f = count0; f < 100; f++) {
                exact += term(psd[f], inv_n2);
                units += gomath::cum_bound_units(psd[f]);
                special = special || gomath::cum_bound_special(psd[f]);
            }
            const float ub = gomath::cum_bound(count0 ? (double)carry : 0.0, units, 100 - count0, a128, per_frame, special);
            columns++;
            // Round 5: the unit counts are formed by k_psd_scan and put together by k_bound_finish (k_noise_scan.hip,
            // k_peaks.hip) - per value max(bits >> 16, 128) WITHOUT the + 1, the + 1 of every frame added once per run, a
            // slot's frames dealt over one or two workgroups whose counts are added, and "special" as one running maximum
            // of bits >> 16 (which a lane shares over all its columns: more columns than needed become +infinity, never
            // fewer).  The same total, the same flag for this column's own frames:
            {
                const int n_run = 100 - count0, split = count0 + (n_run + 1) / 2;
                uint32_t part[2] = {0, 0}, hw_max = 0;
                for (int f = count0; f < 100; f++) {
                    uint32_t bits;
                    std::memcpy(&bits, &psd[f], 4);
                    const uint32_t hw = bits >> 16;
                    part[f < split ? 0 : 1] += hw < 128u ? 128u : hw;
                    hw_max = hw > hw_max ? hw : hw_max;
                }
                part[0] += (uint32_t)(split - count0);
                part[1] += (uint32_t)(100 - split);
                const bool sp = hw_max >= 0x7f80u;
                if (part[0] + part[1] != units || sp != special) {
                    if (bad++ < 10)
                        printf("scan order: units %u + %u against %u, special %d against %d (N %d kind %d)\n", part[0], part[1], units, (int)sp,
                               (int)special, n, kind);
                }
            }
            if (std::isnan(exact)) {
                // a NaN cumulation is never "above" in the reference; ours must reach the exact evaluation or be NaN itself
                if (!(special || std::isnan(ub) || std::isnan(carry) || std::isinf(carry)))
                    if (bad++ < 10)
                        printf("column: NaN cumulation without a special frame (N %d kind %d)\n", n, kind);
                continue;
            }
            if (!(exact <= ub)) {
                if (bad++ < 10)
                    printf("column: N %d kind %d count0 %d exact %.9g bound %.9g\n", n, kind, count0, (double)exact, (double)ub);
            } else if (std::isfinite(exact)) {
                min_col_margin = std::min(min_col_margin, (double)ub - (double)exact);
            }
            // what FindPeaks does with it, against thresholds around the value
            for (int k = -3; k <= 3; k++) {
                const float thr = exact / 100.0f + (float)k * 1e-4f;
                if ((exact / 100.0f > thr) && !(ub / 100.0f > thr)) {
                    if (bad++ < 10)
                        printf("decision: exact above thr, bound not (N %d)\n", n);
                }
            }
        }
    }
    printf("%lld terms, %lld columns, %lld violations; smallest margin of a term %.3g dB, of a column %.3g\n", terms, columns, bad,
           min_term_margin, min_col_margin);
    return bad ? 1 : 0;
}
