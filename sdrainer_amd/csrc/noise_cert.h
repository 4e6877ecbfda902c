// noise_cert.h — dsp.FindNoiseFloor (dsp/fft.go:215-252) and the two rolling-mean inputs of rx/receiver.go:383-384,
// EXACT WHERE THEY ARE CONSUMED, from order-free sums.
//
// What the reference computes per frame: ten sequential float64 sums over windows of W psd values, the first minimum of
// their means (compared in float64), then a sequential float64 sum of (psd[i] - mean)^2 over psd[edge .. resultTo] - up to
// 12 000 strictly ordered additions - and from those two numbers
//     psdNoiseFloor = float32(minMean)                                   -> nf_in  = PSDValueIndB(psdNoiseFloor) + 120
//     variance -> float32(sqrt(variance))                                -> dev_in = float32(float64(PSDValueIndB(.) + 120) * 0.25)
// Nothing else of the two sums is used anywhere (the float64 variance itself never leaves FindNoiseFloor's caller): what
// is consumed is two float32 roundings and nine comparisons.  A float32 rounding hides 29 of a float64's bits, so the
// ORDER of the additions - which is all that makes the chains serial - matters only when a sum lands within its own
// rounding uncertainty (1e-13 ... 1e-12 relative) of a float32 rounding boundary (6e-8 apart) or of another window's
// mean: about three frames in 10^5.  So:
//   * k_psd_scan forms S1_w = sum x and S2_w = sum x^2 per window in whatever order is fast (x^2 of a float32 is exact in
//     float64; the sums are of non-negative terms: any order is within (terms - 1) u of the real sum, u = 2^-53);
//   * certify() below brackets every quantity the reference would have got from ITS order - each sequential sum of n
//     non-negative terms lies within (n - 1) u (1 + tiny) of the real sum [Higham, Accuracy and Stability, (4.4)], the
//     means, the quotient, the square root and the float32 conversions are monotone - and accepts a frame only if both
//     ends of every bracket round / compare the same way; then the value IS the reference's, bit for bit;
//   * a frame that is not accepted (or holds an infinity / NaN) is flagged, and k_noise_exact_list runs the literal loops
//     for it (exact_frame() below, the code of the oracle).
// tests/emu/emu_noise_cert.cpp runs certify() against exact_frame() on the CPU: random and adversarial rows (values
// planted on float32 rounding boundaries, equal windows, huge carriers, zeros, subnormals, infinities); an accepted frame
// must agree in every consumed bit, and the acceptance rate is reported.
//
// Everything here is SDR_HD; compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>

#include "gomath.h"

namespace noise {

constexpr int kMaxWindows = 10;
constexpr double kU = 1.1102230246251565e-16;  // 2^-53

struct Geom {
    int n;          // block size
    int edge;       // edgeWidth
    int window;     // windowSize = (n - 2 edge) / 10
    int n_windows;  // windows the reference's loop evaluates (9 or 10)
    double inv_n2;  // 1 / n^2 (exact)
};

// relative uncertainty of the SCAN's sums (k_psd_scan): a lane adds at most J values in sequence, six butterfly levels
// add the lanes; certify() adds at most ten windows on top.  An upper bound on (additions) u for any block size here.
constexpr double kScanTerms = 48.0;

struct Result {
    float min_mean;   // float32(minValue)
    float dev_in;     // value put into noiseDeviationMean
    float nf_in;      // value put into noiseFloorMean
    double variance;  // the reference's variance to within its bracket (exact when `exact`)
    int window;       // winning window
    bool ok;          // accepted: min_mean, dev_in, nf_in are the reference's bits
    int why;          // not accepted: 1 special sum, 2 means too close to order, 3 minimum on a float32 boundary, 4 / 5 special value,
                      // 6 variance bracket reaches zero, 7 standard deviation on a float32 boundary
};

SDR_HD inline double next_up(double x) { return ::nextafter(x, INFINITY); }
SDR_HD inline double next_down(double x) { return ::nextafter(x, -INFINITY); }

// rx/receiver.go:383-384 from the two float32 values they consume
SDR_HD inline float nf_in_of(float psd_noise_floor, double inv_n2) { return gomath::psd_value_in_db(psd_noise_floor, inv_n2) + 120.0f; }
SDR_HD inline float dev_in_of(float sqrt_var_f32, double inv_n2)
{
    return (float)((double)(gomath::psd_value_in_db(sqrt_var_f32, inv_n2) + 120.0f) * 0.25);
}

// s1[w], s2[w]: the scan's sums over window w; x_at(i): float64(psd[i]) of this frame (read for ONE index).
template <class XAT>
SDR_HD inline Result certify(const double *s1, const double *s2, const Geom &g, XAT x_at)
{
    Result r{};
    r.ok = false;
    const double W = (double)g.window;
    // --- the means and their first minimum (dsp/fft.go:228-238) ---
    // sequential sum of W terms: S_seq = S (1 + t), |t| <= (W - 1) u / (1 - (W - 1) u); the scan's own sum is within
    // kScanTerms u of S; the quotient by W is monotone, and one ulp outward covers its own rounding at both ends.
    const double dm = ((W - 1.0) + kScanTerms) * kU * 1.0001;
    double min_lo = 0, min_hi = 0;
    int win = -1;
    for (int w = 0; w < g.n_windows; w++) {
        const double S = s1[w];
        if (!(S >= 0.0) || !(S <= 1.7e308) || !(s2[w] <= 1.7e308))
            return (r.why = 1), r;  // NaN, infinity (or a negative sum: there is no such psd): the literal loops decide
        const double lo = next_down(next_down(S * (1.0 - dm)) / W);
        const double hi = next_up(next_up(S * (1.0 + dm)) / W);
        if (win < 0 || hi < min_lo) {  // `first`, or certainly smaller
            min_lo = lo;
            min_hi = hi;
            win = w;
        } else if (!(lo >= min_hi)) {
            return (r.why = 2), r;  // may or may not be smaller than the minimum so far
        }
    }
    const float mm_lo = (float)min_lo, mm_hi = (float)min_hi;
    if (!(mm_lo == mm_hi))
        return (r.why = 3), r;
    r.window = win;
    r.min_mean = mm_lo;
    r.nf_in = nf_in_of(mm_lo, g.inv_n2);
    // --- the variance (dsp/fft.go:244-249): sum over psd[edge .. resultTo] inclusive, resultTo = edge + (win + 1) W ---
    double S1 = 0, S2 = 0;
    for (int w = 0; w <= win; w++) {
        S1 += s1[w];
        S2 += s2[w];
    }
    const double xl = x_at(g.edge + (win + 1) * g.window);
    if (!(xl >= 0.0) || !(xl <= 1.8e19))
        return (r.why = 4), r;
    S1 += xl;
    S2 += xl * xl;
    const double nt = (double)(win + 1) * W + 1.0;
    if (!(S2 <= 1.7e308))
        return (r.why = 5), r;
    // V(m) = sum (x - m)^2 = S2 - 2 m S1 + nt m^2 for the mean m in [min_lo, min_hi]; the evaluation loses at most
    // (kScanTerms + 8) u of the sum of the magnitudes; between the two ends a parabola dips at most nt width^2 below
    // the lower of them
    const double width = min_hi - min_lo;
    const double mag = S2 + 2.0 * min_hi * S1 + nt * min_hi * min_hi;
    const double e_abs = mag * (kScanTerms + 8.0) * kU * 1.0001 + nt * width * width;
    const double v_a = (S2 - 2.0 * min_lo * S1) + nt * min_lo * min_lo;
    const double v_b = (S2 - 2.0 * min_hi * S1) + nt * min_hi * min_hi;
    double v_lo = (v_a < v_b ? v_a : v_b) - e_abs;
    double v_hi = (v_a < v_b ? v_b : v_a) + e_abs;
    if (!(v_lo > 0.0) || !(v_hi <= 1.7e308))
        return (r.why = 6), r;
    // the reference's terms: d = fl(x - m) (1 rounding), fl(d d) (1 more) - within 3 u (1 + tiny) of (x - m)^2 - and
    // their sequential sum, nt - 1 more roundings of a growing non-negative sum
    const double dv = (nt + 3.0) * kU * 1.0001;
    v_lo = next_down(v_lo * (1.0 - dv));
    v_hi = next_up(v_hi * (1.0 + dv));
    const double var_lo = next_down(v_lo / W), var_hi = next_up(v_hi / W);
    const float sd_lo = (float)::sqrt(var_lo), sd_hi = (float)::sqrt(var_hi);  // (IEEE sqrt, float32 rounding: monotone)
    r.variance = 0.5 * (var_lo + var_hi);
    if (!(sd_lo == sd_hi))
        return (r.why = 7), r;
    r.dev_in = dev_in_of(sd_lo, g.inv_n2);
    r.ok = true;
    return r;
}

// The literal algorithm (dsp/fft.go:215-252 + rx/receiver.go:383-384) for one frame, in the three pieces the device's
// fallback kernel runs on different threads: the window sums (independent chains: window_sum below), the first minimum of
// their means (select_window), the variance chain and what is made of it (variance_term, finish_frame).
struct Selection {
    double min_value, result_mean;
    int window;
};
SDR_HD inline Selection select_window(const Geom &g, double x0, const double *window_sums)
{
    Selection s{x0, 0.0, 0};  // :217 minValue := float64(psd[0]), overridden by `first`
    bool first = true;
    for (int w = 0; w < g.n_windows; w++) {
        const double mean = window_sums[w] / (double)g.window;
        if (mean < s.min_value || first) {  // :232
            s.min_value = mean;
            first = false;
            s.result_mean = mean;
            s.window = w;
        }
    }
    return s;
}
// :246 math.Pow(float64(psd[i]) - resultMean, 2)
SDR_HD inline double variance_term(double x, double result_mean)
{
    const double d = x - result_mean;
    return d * d;
}
// resultFrom = edge (App. C1 of the survey), resultTo = edge + (window + 1) W, both inclusive
SDR_HD inline int result_to(const Geom &g, int window) { return g.edge + (window + 1) * g.window; }
SDR_HD inline Result finish_frame(const Geom &g, const Selection &s, double term_sum)
{
    Result r{};
    r.variance = term_sum / (double)g.window;
    r.window = s.window;
    r.min_mean = (float)s.min_value;
    r.nf_in = nf_in_of(r.min_mean, g.inv_n2);
    r.dev_in = dev_in_of((float)::sqrt(r.variance), g.inv_n2);
    r.ok = true;
    return r;
}
template <class XAT>
SDR_HD inline Result exact_frame(const Geom &g, XAT x_at, const double *window_sums)
{
    const Selection s = select_window(g, x_at(0), window_sums);
    double sum = 0;
    const int to = result_to(g, s.window);
    for (int i = g.edge; i <= to; i++)
        sum += variance_term(x_at(i), s.result_mean);
    return finish_frame(g, s, sum);
}

// sequential float64 sum of window w (dsp/fft.go:239-241)
template <class XAT>
SDR_HD inline double window_sum(const Geom &g, XAT x_at, int w)
{
    double sum = 0;
    const int b = g.edge + w * g.window;
    for (int i = 0; i < g.window; i++)
        sum += x_at(b + i);
    return sum;
}

}  // namespace noise
