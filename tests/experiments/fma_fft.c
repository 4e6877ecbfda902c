/* Experiment only (tests/experiments/fma_fft_decisions.py): the radix-2 DIT graph of the
 * oracle's FFT (go-dsp fft.FFT as restated in oracle/sdr_oracle.c) with the multiply-adds a contracting compiler fuses:
 * the complex product as one multiply and one fused multiply-add per component, the psd as one of each.  The reference
 * (Go on amd64) never fuses; this is what "allowing FMA" would compute.  Built with -ffp-contract=off -mfma: the fma()
 * calls below are the only fused operations.  Not part of the product, not part of the oracle. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static unsigned reverse_bits(unsigned v, int s)
{
    unsigned r = 0;
    for (int i = 0; i < s; i++) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

/* iq: n interleaved float32 I,Q; wre/wim: go-dsp's factor table for n; psd: n float32, fft-shifted */
__attribute__((visibility("default"))) void fma_iq_to_psd(int n, const float *iq, const double *wre, const double *wim, float *psd)
{
    int s = 0;
    while ((1 << s) < n)
        s++;
    double *buf = malloc(sizeof(double) * (size_t)n * 4);
    double *rre = buf, *rim = buf + n, *tre = buf + 2 * n, *tim = buf + 3 * n;
    for (unsigned i = 0; i < (unsigned)n; i++) {
        unsigned r = reverse_bits(i, s);
        rre[r] = (double)iq[2 * i];
        rim[r] = (double)iq[2 * i + 1];
    }
    for (int stage = 2; stage <= n; stage <<= 1) {
        int blocks = n / stage, s_2 = stage / 2;
        for (int b = 0; b < blocks; b++) {
            int nb = b * stage;
            for (int j = 0; j < s_2; j++) {
                int idx = j + nb, idx2 = idx + s_2;
                double ar = rre[idx2], ai = rim[idx2];
                double br = wre[blocks * j], bi = wim[blocks * j];
                double wr = fma(ar, br, -(ai * bi));
                double wi = fma(ar, bi, ai * br);
                tre[idx] = rre[idx] + wr;
                tim[idx] = rim[idx] + wi;
                tre[idx2] = rre[idx] - wr;
                tim[idx2] = rim[idx] - wi;
            }
        }
        double *sw;
        sw = rre; rre = tre; tre = sw;
        sw = rim; rim = tim; tim = sw;
    }
    for (int i = 0; i < n; i++)
        psd[(i + n / 2) % n] = (float)fma(rre[i], rre[i], rim[i] * rim[i]);
    free(buf);
}
