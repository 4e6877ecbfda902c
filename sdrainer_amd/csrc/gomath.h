// gomath.h — the Go stdlib math routines the strainer path calls, written once for host and device.
//
// The reference computes its dB projection with math.Log10 (dsp/fft.go:79-85) and go-dsp builds its
// twiddle tables with math.Sincos.  On amd64 both are pure-Go (FreeBSD e_log.c / Cephes derived,
// go1.23.4 src/math/log.go, log10.go, sincos.go) and the Go compiler does not fuse multiply-add
// there, so bit-parity needs the same operation sequence in IEEE double with contraction OFF:
// this translation unit must be compiled with -ffp-contract=off (csrc/build.py does).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SDR_HD __host__ __device__
#else
#define SDR_HD
#endif

namespace gomath {

// math.Log (src/math/log.go)
SDR_HD inline double log(double x)
{
    const double Ln2Hi = 6.93147180369123816490e-01;
    const double Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01;
    const double L2 = 3.999999999940941908e-01;
    const double L3 = 2.857142874366239149e-01;
    const double L4 = 2.222219843214978396e-01;
    const double L5 = 1.818357216161805012e-01;
    const double L6 = 1.531383769920937332e-01;
    const double L7 = 1.479819860511658591e-01;
    const double HalfSqrt2 = 1.41421356237309504880168872420969808 / 2;

    if (x != x || x == INFINITY)
        return x;
    if (x < 0)
        return NAN;
    if (x == 0)
        return -INFINITY;

    int ki;
    double f1 = ::frexp(x, &ki);
    if (f1 < HalfSqrt2) {
        f1 *= 2;
        ki--;
    }
    const double f = f1 - 1;
    const double k = (double)ki;

    const double s = f / (2 + f);
    const double s2 = s * s;
    const double s4 = s2 * s2;
    const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

// math.Log2 (src/math/log10.go)
SDR_HD inline double log2(double x)
{
    const double InvLn2 = 1 / 0.693147180559945309417232121458176568;
    int e;
    const double frac = ::frexp(x, &e);
    if (frac == 0.5)
        return (double)(e - 1);
    return gomath::log(frac) * InvLn2 + (double)e;
}

// math.Log10 (src/math/log10.go)
SDR_HD inline double log10(double x)
{
    const double Ln2OverLn10 = 0.301029995663981195213738894724493027;
    return gomath::log2(x) * Ln2OverLn10;
}

// dsp.PSDValueIndB[float32] (dsp/fft.go:83-85): T(10*log10(20*float64(x)/N^2)).
// N is a power of two, so the division by N^2 is an exact scaling: inv_n2 = 2^(-2 log2 N).
SDR_HD inline float psd_value_in_db(float psd, double inv_n2)
{
    return (float)(10.0 * gomath::log10(20.0 * (double)psd * inv_n2));
}

// ---------------------------------------------------------------------------------------------
// Certified fast path for psd_value_in_db.
//
// The result is a float32, but the reference computes it through a ~60-instruction float64 log.  The
// argument of that log, v = 20 * float64(psd) / N^2, is EXACT in float64 (24-bit psd times the 3-bit 20
// times a power of two), so the mathematical value the reference approximates is
//     y = 10 log10(m) + E * 10 log10(2) + [10 log10(20) - 2 log2(N) * 10 log10(2)],   psd = m * 2^E, 1 <= m < 2.
// The fast path evaluates that straight from the float32 bits: a 1024-entry table over the top ten mantissa
// bits gives c_i ~ m with |m / c_i - 1| <= 2^-11, a degree-4 log1p finishes 10 log10(m) (truncation 6e-18),
// and a 512-entry table indexed by sign+exponent supplies the E term, with NaN in the entries of zero /
// subnormal / infinite / NaN / negative inputs.  FMA is allowed: only the error bound matters here.  The
// float32 of y is ACCEPTED only if no float32 rounding boundary lies within kDbGuard of y (y - guard and
// y + guard round to the same float32; a NaN y never does).  Both y and the reference's value lie within
// ~1e-13 of the true logarithm (tests/emu/emu_log.cpp measures it), kDbGuard is 1e-10, so an accepted
// result is the float32 the reference would have produced; anything else (about three values in 10^5,
// plus every special input) reports `false` and the caller runs the literal Go algorithm.
// ---------------------------------------------------------------------------------------------
struct LogTabEntry {
    double inv_c;    // 1 / c_i,  c_i = 1 + (i + 0.5) / 1024
    double db_of_c;  // 10 log10(c_i)
};
constexpr int kLogTabSize = 1024;  // LogTabEntry[kLogTabSize], 16 KB
constexpr int kExpTabSize = 512;   // double[kExpTabSize], 4 KB: index = float32 bits >> 23 (sign and exponent)
constexpr int kDbTabBytes = kLogTabSize * 16 + kExpTabSize * 8;
constexpr double kDbGuard = 1e-10;

// Both tables as one blob: [LogTabEntry x 1024][double x 512]
struct DbTables {
    const LogTabEntry *log_tab;
    const double *exp_tab;
};
SDR_HD inline DbTables db_tables(const void *blob)
{
    const LogTabEntry *lt = static_cast<const LogTabEntry *>(blob);
    return DbTables{lt, reinterpret_cast<const double *>(lt + kLogTabSize)};
}

// y ~= 10*log10(20 * psd / N^2) from the float32 bits of psd (NaN for inputs the tables mark as special)
SDR_HD inline double db_fast_y(float psd, DbTables t)
{
    uint32_t bits;
    __builtin_memcpy(&bits, &psd, sizeof bits);
    const uint32_t mant_one = (bits & 0x007fffffu) | 0x3f800000u;
    float mf;
    __builtin_memcpy(&mf, &mant_one, sizeof mf);
    const double m = (double)mf;  // exact
    const LogTabEntry e = t.log_tab[(bits >> 13) & 1023u];
    const double eb = t.exp_tab[bits >> 23];
    const double r = __builtin_fma(m, e.inv_c, -1.0);  // |r| <= 2^-11
    // (10 / ln 10) * log1p(r) = r * K * (1 - r/2 + r^2/3 - r^3/4)  (+ O(r^5))
    const double K = 4.34294481903251827651128918916605082;  // 10 / ln(10)
    double q = __builtin_fma(-K / 4.0, r, K / 3.0);
    q = __builtin_fma(q, r, -K / 2.0);
    q = __builtin_fma(q, r, K);
    return __builtin_fma(q, r, e.db_of_c) + eb;
}

SDR_HD inline bool psd_value_in_db_fast(float psd, DbTables t, float *out)
{
    const double y = db_fast_y(psd, t);
    // Rounding to float32 is monotone: if y - guard and y + guard round to the same float32, so does every
    // value between them, the reference's among them.  (Covers asymmetric intervals at powers of two and
    // values straddling zero without any bit tests; false for NaN.)
    const float lo = (float)(y - kDbGuard), hi = (float)(y + kDbGuard);
    *out = lo;
    return lo == hi;
}

// host: the tables above for block size 2^logn
inline void build_db_tables(int logn, void *blob)
{
    LogTabEntry *lt = static_cast<LogTabEntry *>(blob);
    double *et = reinterpret_cast<double *>(lt + kLogTabSize);
    for (int i = 0; i < kLogTabSize; i++) {
        const long double c = 1.0L + ((long double)i + 0.5L) / (long double)kLogTabSize;
        lt[i].inv_c = (double)(1.0L / c);
        lt[i].db_of_c = (double)(10.0L * ::log10l(c));
    }
    const long double A = 10.0L * ::log10l(2.0L);
    const long double B = 10.0L * ::log10l(20.0L) - 2.0L * (long double)logn * A;
    for (int i = 0; i < kExpTabSize; i++) {
        const int e = i & 255;
        const bool special = (i >> 8) != 0 || e == 0 || e == 255;  // negative; zero / subnormal; inf / NaN
        et[i] = special ? (double)NAN : (double)((long double)(e - 127) * A + B);
    }
}

// ---------------------------------------------------------------------------------------------
// Upper bound of a cumulation (k_peaks.hip k_cum_bound; tests/emu/emu_cum_bound.cpp checks every claim below on the CPU).
// A term of the cumulation is s = fl32(fl32(10 log10(20 psd / N^2)) + 120) (dsp/fft.go:79-81, rx/receiver.go:377).  For
// psd = m 2^E, 1 <= m < 2:  log2(psd) = E + log2(m) <= E + (m - 1) + 0.0861, and the float32 bits of psd, shifted right
// by 16, are (E + 127) 128 + floor(128 (m - 1)).  With units(psd) = max(bits >> 16, 128) + 1 (the max: zero and subnormal
// psd count as 2^-126, which is larger):
//     s <= A units / 128 + per_frame,   A = 10 log10 2,   per_frame = A (-127 + 0.0861) + 10 log10 20 - 20 log10 N + 120 + 2e-5
// (2e-5: the two float32 roundings of a term - it is below 512 in magnitude - and the reference's own logarithm, 1e-13).
// The ordered float32 sum of up to 100 such terms on top of a carry exceeds the real sum by less than 0.2 (|partial
// sums| < 2^16 for any finite psd: a hundred additions, half an ulp of 2^-8 each); the bound is formed in float64 and
// given 0.05 more before it is rounded to float32.  A column that holds an infinite or NaN psd (or a set sign bit: no sum
// of squares has one) gets the bound +infinity: its exact evaluation decides, as the reference's would.
// ---------------------------------------------------------------------------------------------
SDR_HD inline uint32_t cum_bound_units(float psd)
{
    uint32_t bits;
    __builtin_memcpy(&bits, &psd, sizeof bits);
    const uint32_t hw = bits >> 16;
    return (hw < 128u ? 128u : hw) + 1u;
}
SDR_HD inline bool cum_bound_special(float psd)  // infinity, NaN, sign bit: (bits >> 16) >= 0x7f80
{
    uint32_t bits;
    __builtin_memcpy(&bits, &psd, sizeof bits);
    return (bits >> 16) >= 0x7f80u;
}
// host: the constants for block size n, rounded so that the bound can only grow
inline void cum_bound_constants(int n, double *a128, double *per_frame)
{
    const double A_up = 3.0102999566398121, A_down = 3.0102999566398116;  // 10 log10(2) = 3.010299956639811952...
    const double log2_slack = 0.0861;  // > max over m in [1, 2) of log2(m) - (m - 1) = 0.086071...
    const double K = 10.0 * ::log10(20.0) - 20.0 * ::log10((double)n);
    *a128 = A_up / 128.0;
    // (A multiplies a NEGATIVE number here: the smaller value keeps the product an upper bound)
    *per_frame = A_down * (-127.0 + log2_slack) + K + 120.0 + 2e-5 + 1e-9;
}
// the bound of carry + (sum of `frames` terms whose units add up to `units`), as a float32
SDR_HD inline float cum_bound(double carry, uint32_t units, int frames, double a128, double per_frame, bool special)
{
    return special ? __builtin_inff() : (float)(carry + a128 * (double)units + (double)frames * per_frame + 0.25);
}

// (host only)
// math.Sincos (src/math/sincos.go), |x| < 2^29 branch.  Host only: twiddle tables are built once on
// the host and uploaded, exactly as go-dsp caches them.
inline void sincos(double x, double *sn, double *cs)
{
    static const double S[6] = {1.58962301576546568060E-10, -2.50507477628578072866E-8, 2.75573136213857245213E-6,
                                -1.98412698295895385996E-4, 8.33333333332211858878E-3,  -1.66666666666666307295E-1};
    static const double Cc[6] = {-1.13585365213876817300E-11, 2.08757008419747316778E-9, -2.75573141792967388112E-7,
                                 2.48015872888517045348E-5,   -1.38888888888730564116E-3, 4.16666666666665929218E-2};
    const double PI4A = 7.85398125648498535156E-1, PI4B = 3.77489470793079817668E-8, PI4C = 2.69515142907905952645E-15;
    const double Pi = 3.14159265358979323846264338327950288;
    if (x == 0) {
        *sn = x;
        *cs = 1;
        return;
    }
    if (x != x || std::isinf(x)) {
        *sn = NAN;
        *cs = NAN;
        return;
    }
    bool sinSign = false, cosSign = false;
    if (x < 0) {
        x = -x;
        sinSign = true;
    }
    uint64_t j = (uint64_t)(x * (4 / Pi));
    double y = (double)j;
    if (j & 1) {
        j++;
        y++;
    }
    j &= 7;
    const double z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    if (j > 3) {
        j -= 4;
        sinSign = !sinSign;
        cosSign = !cosSign;
    }
    if (j > 1)
        cosSign = !cosSign;
    const double zz = z * z;
    double c = 1.0 - 0.5 * zz + zz * zz * ((((((Cc[0] * zz) + Cc[1]) * zz + Cc[2]) * zz + Cc[3]) * zz + Cc[4]) * zz + Cc[5]);
    double s = z + z * zz * ((((((S[0] * zz) + S[1]) * zz + S[2]) * zz + S[3]) * zz + S[4]) * zz + S[5]);
    if (j == 1 || j == 2) {
        const double t = s;
        s = c;
        c = t;
    }
    *sn = sinSign ? -s : s;
    *cs = cosSign ? -c : c;
}

}  // namespace gomath
