// twiddles.h — host-side construction of go-dsp's radix-2 factor table.
//
// go-dsp (github.com/mjibson/go-dsp v0.0.0-20180508042940-11479a337f12, fft/radix2.go) caches
// W_n[k] = e^{-2 pi i k / n} per size: the size-4 table is the literal {1, -i, -1, i}; each doubling
// copies the even entries from the half-size table and fills the odd ones with
// math.Sincos(-2*Pi/float64(n)*float64(k)).  The FFT kernels consume these VALUES (re-laid-out per
// register pass, fft_f64.h), so the device never evaluates a sine.
#pragma once
#include <vector>

#include "gomath.h"

namespace fft64 {

inline void radix2_factors(int n, std::vector<double> &re, std::vector<double> &im)
{
    const double MinusTwoPi = -2.0 * 3.14159265358979323846264338327950288;
    re.assign((size_t)n, 0.0);
    im.assign((size_t)n, 0.0);
    std::vector<double> pre = {1, 0, -1, 0}, pim = {0, -1, 0, 1};
    if (n < 4) {
        for (int k = 0; k < n; k++) {
            re[k] = k == 0 ? 1.0 : -1.0;
            im[k] = 0.0;
        }
        return;
    }
    for (int i = 8; i <= n; i <<= 1) {
        std::vector<double> cre((size_t)i), cim((size_t)i);
        for (int k = 0, j = 0; k < i; k += 2, j++) {
            cre[k] = pre[j];
            cim[k] = pim[j];
        }
        for (int k = 1; k < i; k += 2) {
            double s, c;
            gomath::sincos(MinusTwoPi / (double)i * (double)k, &s, &c);
            cre[k] = c;
            cim[k] = s;
        }
        pre.swap(cre);
        pim.swap(cim);
    }
    re = pre;
    im = pim;
}

}  // namespace fft64
