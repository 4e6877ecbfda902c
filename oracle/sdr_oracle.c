/*
 * sdr_oracle.c — CPU restatement of the sdrainer IQ-strainer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (sdrainer_amd/, libsdrainer_hip.so) never links or imports it.
 *
 * What this is: a plain-C restatement of the reference's Go algorithm for the
 * path named by BASELINE.json (FFT -> dB/PSD projection -> noise floor ->
 * thresholds -> per-listener envelope -> Morse decoder; cumulation -> peak
 * scan), following the reference's arithmetic literally — same types
 * (float32 / float64), same evaluation order, no FMA contraction (build with
 * -ffp-contract=off), including the quirks listed in SURVEY.md App. C.
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - cw/decode.go restatement: PINNED by the reference's nine recorded
 *     streams + expected strings (cw/decode_test.go:184-192) and by the speed
 *     tests (decode_test.go:58-175), see tests/test_oracle_decoder.py.
 *   - dsp.BoolDebouncer: PINNED by dsp/dsp_test.go:13-23.
 *   - binToSpectrumIndex / FrequencyMapping: PINNED by dsp/fft_test.go.
 *   - PeaksTable: PINNED by rx/peaks_test.go scenarios.
 *   - FFT values / PSD / MagnitudeIndB / FindNoiseFloor / FindPeaks /
 *     Receiver.run: the reference holds NO test or golden vector for these,
 *     the Go toolchain is absent, and the FFT itself lives in the un-vendored
 *     dependency github.com/mjibson/go-dsp v0.0.0-20180508042940-11479a337f12
 *     => PARITY UNPINNED for those rows.  The FFT is cross-checked against
 *     numpy's float64 FFT (tests/test_oracle_dsp.py) and the Go stdlib math
 *     routines (math.Log10 / math.Sincos, go1.23.4, pure-Go on amd64) are
 *     restated from their published algorithm as recalled.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* Go stdlib math (go1.23.4 src/math, pure-Go paths used on amd64)           */
/* ------------------------------------------------------------------------ */

/* math.Log — src/math/log.go (FreeBSD e_log.c derived).  Used via Log2/Log10
 * by dsp/fft.go:79-85. */
ORC_API double orc_go_log(double x)
{
    const double Ln2Hi = 6.93147180369123816490e-01; /* 3fe62e42 fee00000 */
    const double Ln2Lo = 1.90821492927058770002e-10; /* 3dea39ef 35793c76 */
    const double L1 = 6.666666666666735130e-01;
    const double L2 = 3.999999999940941908e-01;
    const double L3 = 2.857142874366239149e-01;
    const double L4 = 2.222219843214978396e-01;
    const double L5 = 1.818357216161805012e-01;
    const double L6 = 1.531383769920937332e-01;
    const double L7 = 1.479819860511658591e-01;
    const double Sqrt2 = 1.41421356237309504880168872420969808;

    if (isnan(x) || (isinf(x) && x > 0))
        return x;
    if (x < 0)
        return NAN;
    if (x == 0)
        return -INFINITY;

    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < Sqrt2 / 2) {
        f1 *= 2;
        ki--;
    }
    double f = f1 - 1;
    double k = (double)ki;

    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

/* math.Log2 — src/math/log10.go: Frexp, exact for powers of two. */
ORC_API double orc_go_log2(double x)
{
    const double Ln2 = 0.693147180559945309417232121458176568;
    int e;
    double frac = frexp(x, &e);
    if (frac == 0.5)
        return (double)(e - 1);
    return orc_go_log(frac) * (1 / Ln2) + (double)e;
}

/* math.Log10 — src/math/log10.go: log2(x) * (Ln2/Ln10). */
ORC_API double orc_go_log10(double x)
{
    const double Ln2_over_Ln10 = 0.301029995663981195213738894724493027; /* Ln2/Ln10 */
    return orc_go_log2(x) * Ln2_over_Ln10;
}

/* math.Sincos — src/math/sincos.go (Cephes derived), |x| < 2^29 branch only.
 * Used by go-dsp to build its twiddle tables. */
ORC_API void orc_go_sincos(double x, double *sn, double *cs)
{
    static const double _sin[6] = {
        1.58962301576546568060E-10, -2.50507477628578072866E-8,
        2.75573136213857245213E-6,  -1.98412698295895385996E-4,
        8.33333333332211858878E-3,  -1.66666666666666307295E-1,
    };
    static const double _cos[6] = {
        -1.13585365213876817300E-11, 2.08757008419747316778E-9,
        -2.75573141792967388112E-7,  2.48015872888517045348E-5,
        -1.38888888888730564116E-3,  4.16666666666665929218E-2,
    };
    const double PI4A = 7.85398125648498535156E-1;
    const double PI4B = 3.77489470793079817668E-8;
    const double PI4C = 2.69515142907905952645E-15;
    const double Pi = 3.14159265358979323846264338327950288;

    if (x == 0) {
        *sn = x;
        *cs = 1;
        return;
    }
    if (isnan(x) || isinf(x)) {
        *sn = NAN;
        *cs = NAN;
        return;
    }
    int sinSign = 0, cosSign = 0;
    if (x < 0) {
        x = -x;
        sinSign = 1;
    }
    uint64_t j = (uint64_t)(x * (4 / Pi));
    double y = (double)j;
    if (j & 1) {
        j++;
        y++;
    }
    j &= 7;
    double z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    if (j > 3) {
        j -= 4;
        sinSign = !sinSign;
        cosSign = !cosSign;
    }
    if (j > 1)
        cosSign = !cosSign;

    double zz = z * z;
    double c = 1.0 - 0.5 * zz +
               zz * zz * ((((((_cos[0] * zz) + _cos[1]) * zz + _cos[2]) * zz + _cos[3]) * zz + _cos[4]) * zz + _cos[5]);
    double s = z + z * zz * ((((((_sin[0] * zz) + _sin[1]) * zz + _sin[2]) * zz + _sin[3]) * zz + _sin[4]) * zz + _sin[5]);
    if (j == 1 || j == 2) {
        double t = s;
        s = c;
        c = t;
    }
    if (cosSign)
        c = -c;
    if (sinSign)
        s = -s;
    *sn = s;
    *cs = c;
}

/* ------------------------------------------------------------------------ */
/* go-dsp fft (github.com/mjibson/go-dsp v0.0.0-20180508042940-11479a337f12) */
/* restated from its published radix-2 algorithm          */
/* ------------------------------------------------------------------------ */

/* go-dsp fft/radix2.go getRadix2Factors/EnsureRadix2Factors: W[k]=e^{-2 pi i k/n}.
 * size-4 table is the literal {1,-i,-1,i}; each doubling copies the even
 * entries from the half-size table and computes odd entries with
 * math.Sincos(-2*Pi/float64(i)*float64(n)).  Output: n entries. */
ORC_API void orc_radix2_factors(int n, double *re, double *im)
{
    const double MinusTwoPi = -2.0 * 3.14159265358979323846264338327950288;
    if (n < 4) { /* n==1,2: go-dsp never reads beyond factors[0] */
        for (int k = 0; k < n; k++) {
            re[k] = (k == 0) ? 1.0 : -1.0;
            im[k] = 0.0;
        }
        return;
    }
    double *pre = (double *)malloc(sizeof(double) * (size_t)n);
    double *pim = (double *)malloc(sizeof(double) * (size_t)n);
    int p = 4;
    pre[0] = 1; pim[0] = 0;
    pre[1] = 0; pim[1] = -1;
    pre[2] = -1; pim[2] = 0;
    pre[3] = 0; pim[3] = 1;
    double *cre = (double *)malloc(sizeof(double) * (size_t)n);
    double *cim = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 8; i <= n; i <<= 1) {
        for (int k = 0, j = 0; k < i; k += 2, j++) {
            cre[k] = pre[j];
            cim[k] = pim[j];
        }
        for (int k = 1; k < i; k += 2) {
            double s, c;
            orc_go_sincos(MinusTwoPi / (double)i * (double)k, &s, &c);
            cre[k] = c;
            cim[k] = s;
        }
        memcpy(pre, cre, sizeof(double) * (size_t)i);
        memcpy(pim, cim, sizeof(double) * (size_t)i);
        p = i;
    }
    memcpy(re, pre, sizeof(double) * (size_t)p);
    memcpy(im, pim, sizeof(double) * (size_t)p);
    free(pre); free(pim); free(cre); free(cim);
}

static unsigned reverse_bits(unsigned v, int s)
{
    unsigned r = 0;
    for (int i = 0; i < s; i++) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

/* Per-size cache of twiddle tables (go-dsp caches them in a global map). */
#define ORC_MAX_LOG2 24
static double *g_fac_re[ORC_MAX_LOG2 + 1];
static double *g_fac_im[ORC_MAX_LOG2 + 1];

static int ilog2(int n)
{
    int s = 0;
    while ((1 << s) < n)
        s++;
    return s;
}

static void ensure_factors(int n)
{
    int s = ilog2(n);
    if (g_fac_re[s])
        return;
    double *re = (double *)malloc(sizeof(double) * (size_t)n);
    double *im = (double *)malloc(sizeof(double) * (size_t)n);
    orc_radix2_factors(n, re, im);
    g_fac_re[s] = re;
    g_fac_im[s] = im;
}

/* go-dsp fft/radix2.go radix2FFT: bit-reversal reorder, then decimation-in-time
 * stages stage=2,4,..,n: t[idx]=r[idx]+r[idx2]*W[blocks*j], t[idx2]=r[idx]-...
 * complex128 arithmetic as the Go compiler emits it on amd64 (no FMA):
 * (a+bi)(c+di) = (ac - bd) + (ad + bc)i.  n must be a power of two
 * (the only branch the BASELINE configs reach). */
ORC_API void orc_fft_radix2(int n, const double *xre, const double *xim, double *yre, double *yim)
{
    if (n <= 1) {
        if (n == 1) { yre[0] = xre[0]; yim[0] = xim[0]; }
        return;
    }
    ensure_factors(n);
    int s = ilog2(n);
    const double *wre = g_fac_re[s], *wim = g_fac_im[s];
    double *rre = (double *)malloc(sizeof(double) * (size_t)n);
    double *rim = (double *)malloc(sizeof(double) * (size_t)n);
    double *tre = (double *)malloc(sizeof(double) * (size_t)n);
    double *tim = (double *)malloc(sizeof(double) * (size_t)n);
    for (unsigned i = 0; i < (unsigned)n; i++) {
        unsigned r = reverse_bits(i, s);
        rre[r] = xre[i];
        rim[r] = xim[i];
    }
    for (int stage = 2; stage <= n; stage <<= 1) {
        int blocks = n / stage;
        int s_2 = stage / 2;
        for (int b = 0; b < blocks; b++) {
            int nb = b * stage;
            for (int j = 0; j < s_2; j++) {
                int idx = j + nb;
                int idx2 = idx + s_2;
                double ar = rre[idx2], ai = rim[idx2];
                double br = wre[blocks * j], bi = wim[blocks * j];
                double wr = ar * br - ai * bi;
                double wi = ar * bi + ai * br;
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
    memcpy(yre, rre, sizeof(double) * (size_t)n);
    memcpy(yim, rim, sizeof(double) * (size_t)n);
    free(rre); free(rim); free(tre); free(tim);
}

/* ------------------------------------------------------------------------ */
/* dsp/fft.go                                                                */
/* ------------------------------------------------------------------------ */

/* dsp/fft.go:54-57 binToSpectrumIndex */
ORC_API int orc_bin_to_spectrum_index(int bin, int block_size)
{
    int center = block_size / 2;
    return (bin + center) % block_size;
}

/* dsp/fft.go:71-73 PSD[T=float32]: T(Pow(re,2)+Pow(im,2)); math.Pow(x,2)
 * reduces to the exactly rounded x*x (Frexp / square / Ldexp). */
static float psd_f32(double re, double im)
{
    return (float)(re * re + im * im);
}

/* dsp/fft.go:83-85 PSDValueIndB[T=float32] */
ORC_API float orc_psd_value_in_db(float psd_value, int block_size)
{
    double n2 = (double)block_size * (double)block_size; /* math.Pow(float64(N),2) */
    return (float)(10.0 * orc_go_log10(20.0 * (double)psd_value / n2));
}

/* dsp/fft.go:23-37 FFT.IQToSpectrumAndPSD with the projection closure of
 * rx/receiver.go:376-378 (MagnitudeIndB + dBmShift(120), a float32 add);
 * dsp/fft.go:59-69 setSamplesFromIQ; dsp/fft.go:79-81 MagnitudeIndB (PSD is
 * rounded to float32 BEFORE the log). */
ORC_API void orc_iq_to_spectrum_and_psd(int n, const float *iq, float *spectrum, float *psd)
{
    double *xre = (double *)malloc(sizeof(double) * (size_t)n * 4);
    double *xim = xre + n, *yre = xre + 2 * n, *yim = xre + 3 * n;
    for (int i = 0; i < n; i++) {
        xre[i] = (double)iq[2 * i];
        xim[i] = (double)iq[2 * i + 1];
    }
    orc_fft_radix2(n, xre, xim, yre, yim);
    for (int i = 0; i < n; i++) {
        int k = orc_bin_to_spectrum_index(i, n);
        float p = psd_f32(yre[i], yim[i]);
        float db = orc_psd_value_in_db(p, n); /* same formula as MagnitudeIndB */
        spectrum[k] = db + 120.0f;
        psd[k] = p;
    }
    free(xre);
}

/* Raw FFT output (complex128) of one IQ frame, for the numpy cross-check. */
ORC_API void orc_iq_fft(int n, const float *iq, double *yre, double *yim)
{
    double *xre = (double *)malloc(sizeof(double) * (size_t)n * 2);
    double *xim = xre + n;
    for (int i = 0; i < n; i++) {
        xre[i] = (double)iq[2 * i];
        xim[i] = (double)iq[2 * i + 1];
    }
    orc_fft_radix2(n, xre, xim, yre, yim);
    free(xre);
}

/* dsp/fft.go:215-252 FindNoiseFloor — literal, including the quirks of
 * SURVEY.md App. C1 (resultFrom is always edgeWidth; variance is summed over
 * psd[edge..resultTo] inclusive but divided by windowSize). */
ORC_API void orc_find_noise_floor(const float *psd, int n, int edge_width, float *min_mean, double *variance)
{
    int windowSize = (n - 2 * edge_width) / 10;
    double minValue = (double)psd[0];
    double sum = 0;
    int count = 0;
    int first = 1;
    int from = 0;
    double resultMean = 0;
    int resultFrom = 0;
    int resultTo = 0;
    for (int i = edge_width; i < n - edge_width; i++) {
        if (count == 0)
            from = i;
        if (count == windowSize) {
            count = 0;
            double mean = sum / (double)windowSize;
            if (mean < minValue || first) {
                minValue = mean;
                first = 0;
                resultMean = mean;
                resultFrom = from;
                resultTo = i;
            }
            sum = 0;
        }
        sum += (double)psd[i];
        count++;
    }
    sum = 0;
    for (int i = resultFrom; i <= resultTo; i++) {
        double d = (double)psd[i] - resultMean;
        sum += d * d; /* math.Pow(d, 2) */
    }
    *variance = sum / (double)windowSize;
    *min_mean = (float)minValue;
}

/* dsp/fft.go:95-135 FrequencyMapping[F=int] */
typedef struct {
    int sampleRate, blockSize;
    double binSize;
    int centerBin;
    int64_t centerFrequency, fromFrequency;
} orc_freqmap;

static void freqmap_init(orc_freqmap *m, int sampleRate, int blockSize, int64_t center)
{
    m->sampleRate = sampleRate;
    m->blockSize = blockSize;
    m->binSize = (double)sampleRate / (double)blockSize;
    m->centerBin = blockSize / 2;
    m->centerFrequency = center;
    m->fromFrequency = center - sampleRate / 2;
}

/* Go int(float64) on amd64 is CVTTSD2SQ: NaN / out of range -> INT64_MIN. */
static int64_t go_int_of_f64(double v)
{
    if (!(v > -9223372036854775808.0 && v < 9223372036854775808.0))
        return INT64_MIN;
    return (int64_t)v;
}

static int64_t freqmap_bin_to_frequency(const orc_freqmap *m, int bin, double location)
{
    double delta = m->binSize * location;
    /* wrap-around add as Go's int arithmetic does */
    return (int64_t)((uint64_t)m->fromFrequency + (uint64_t)go_int_of_f64((double)bin * m->binSize + delta));
}

static int freqmap_frequency_to_bin(const orc_freqmap *m, int64_t f)
{
    int64_t bin = go_int_of_f64(((double)f - (double)m->fromFrequency) / m->binSize);
    if (bin > m->blockSize - 1)
        bin = m->blockSize - 1;
    if (bin < 0)
        bin = 0;
    return (int)bin;
}

ORC_API int64_t orc_bin_to_frequency(int sample_rate, int block_size, int64_t center, int bin, double location)
{
    orc_freqmap m;
    freqmap_init(&m, sample_rate, block_size, center);
    return freqmap_bin_to_frequency(&m, bin, location);
}

ORC_API int orc_frequency_to_bin(int sample_rate, int block_size, int64_t center, int64_t f)
{
    orc_freqmap m;
    freqmap_init(&m, sample_rate, block_size, center);
    return freqmap_frequency_to_bin(&m, f);
}

/* dsp/fft.go:292-309 PeakCenterCorrection */
ORC_API double orc_peak_center_correction(int bin, const float *spectrum, int n)
{
    if (bin <= 0 || bin >= n - 1)
        return 0;
    double y1 = fabs((double)spectrum[bin - 1]);
    double y2 = fabs((double)spectrum[bin]);
    double y3 = fabs((double)spectrum[bin + 1]);
    return (y3 - y1) / (2 * (2 * y2 - y1 - y3));
}

/* dsp/fft.go:179-188 Peak[float32,int] with fixed-width fields */
typedef struct {
    int32_t from, to;
    int64_t from_frequency, to_frequency, signal_frequency;
    float signal_value;
    int32_t signal_bin;
} orc_peak;

/* dsp/fft.go:254-285 FindPeaks */
ORC_API int orc_find_peaks(const float *spectrum, int n, int cumulation_size, float threshold,
                           int sample_rate, int64_t center_frequency, orc_peak *peaks, int max_peaks)
{
    orc_freqmap m;
    freqmap_init(&m, sample_rate, n, center_frequency);
    int np = 0;
    int open = 0;
    orc_peak cur;
    memset(&cur, 0, sizeof cur);
    for (int i = 0; i < n; i++) {
        float value = spectrum[i] / (float)cumulation_size;
        if (!open && value > threshold) {
            memset(&cur, 0, sizeof cur);
            cur.from = i;
            cur.signal_value = value;
            cur.signal_bin = i;
            open = 1;
        } else if (open && value <= threshold) {
            cur.to = i - 1;
            cur.from_frequency = freqmap_bin_to_frequency(&m, cur.from, -0.5);
            cur.to_frequency = freqmap_bin_to_frequency(&m, cur.to, 0.5);
            double corr = orc_peak_center_correction(cur.signal_bin, spectrum, n);
            cur.signal_frequency = freqmap_bin_to_frequency(&m, cur.signal_bin, corr);
            if (np < max_peaks)
                peaks[np] = cur;
            np++;
            open = 0;
        } else if (open && cur.signal_value < value) {
            cur.signal_value = value;
            cur.signal_bin = i;
        }
    }
    if (open) {
        cur.to = n - 1;
        cur.from_frequency = freqmap_bin_to_frequency(&m, cur.from, -0.5);
        cur.to_frequency = freqmap_bin_to_frequency(&m, cur.to, 0.5);
        double corr = orc_peak_center_correction(cur.signal_bin, spectrum, n);
        cur.signal_frequency = freqmap_bin_to_frequency(&m, cur.signal_bin, corr);
        if (np < max_peaks)
            peaks[np] = cur;
        np++;
    }
    return np;
}

/* ------------------------------------------------------------------------ */
/* kiwi/client.go:284-308 decodeIQMessage / decodeIQBytes (source wire format) */
/* ------------------------------------------------------------------------ */

/* payload = body of one SND message; returns the number of float32 values written
 * (= (n_bytes - 17) / 2), or -1 if the payload is shorter than its 17-byte header.
 * The reference holds no test vector for this function (parity unpinned); the arithmetic is one
 * correctly rounded float32 division: float32(int16(be16)) / float32(math.MaxInt16). */
ORC_API long orc_decode_iq_message(const uint8_t *payload, size_t n_bytes, float *iq)
{
    if (n_bytes < 17)
        return -1;
    const uint8_t *b = payload + 17;
    size_t n = (n_bytes - 17) / 2;
    for (size_t i = 0; i < n; i++) {
        uint16_t raw = (uint16_t)((b[2 * i] << 8) | b[2 * i + 1]); /* binary.BigEndian.Uint16 */
        iq[i] = (float)(int16_t)raw / 32767.0f;
    }
    return (long)n;
}

/* ------------------------------------------------------------------------ */
/* dsp/dsp.go                                                                */
/* ------------------------------------------------------------------------ */

/* dsp/dsp.go:239-281 RollingMean[float32] */
typedef struct {
    float *values;
    int len;
    float n;
    int next;
    float sumForMean, mean;
} orc_rolling_mean;

static void rm_init(orc_rolling_mean *v, int n)
{
    v->values = (float *)calloc((size_t)n, sizeof(float));
    v->len = n;
    v->n = (float)n;
    v->next = 0;
    v->sumForMean = 0;
    v->mean = 0;
}

static float rm_put(orc_rolling_mean *v, float value)
{
    v->sumForMean -= v->values[v->next];
    v->values[v->next] = value;
    v->sumForMean += v->values[v->next];
    v->mean = v->sumForMean / v->n;
    v->next = (v->next + 1) % v->len;
    return v->mean;
}

ORC_API void *orc_rolling_mean_new(int n)
{
    orc_rolling_mean *v = (orc_rolling_mean *)malloc(sizeof *v);
    rm_init(v, n);
    return v;
}
ORC_API float orc_rolling_mean_put(void *h, float value) { return rm_put((orc_rolling_mean *)h, value); }
ORC_API void orc_rolling_mean_free(void *h)
{
    free(((orc_rolling_mean *)h)->values);
    free(h);
}

/* dsp/dsp.go:139-182 BoolDebouncer */
typedef struct {
    int threshold;
    int effectiveState, lastRawState, stateCount;
} orc_debouncer;

static void deb_init(orc_debouncer *d, int threshold)
{
    memset(d, 0, sizeof *d);
    d->threshold = threshold;
}

static int deb_debounce(orc_debouncer *d, int raw)
{
    if (d->threshold < 2)
        return raw;
    if (raw != d->lastRawState)
        d->stateCount = 1;
    else
        d->stateCount++;
    d->lastRawState = raw;
    if (d->stateCount >= d->threshold) {
        if (raw != d->effectiveState)
            d->effectiveState = raw;
    }
    return d->effectiveState;
}

ORC_API void *orc_debouncer_new(int threshold)
{
    orc_debouncer *d = (orc_debouncer *)malloc(sizeof *d);
    deb_init(d, threshold);
    return d;
}
ORC_API int orc_debouncer_debounce(void *h, int raw) { return deb_debounce((orc_debouncer *)h, raw != 0); }
ORC_API void orc_debouncer_free(void *h) { free(h); }

/* dsp/dsp.go:34-136 Goertzel */
typedef struct {
    double pitch;
    int sampleRate, blocksize;
    double coeff, magnitudeLimitLow, magnitudeLimit, magnitudeThreshold;
} orc_goertzel;

/* Go math.Round: half away from zero */
static double go_round(double x) { return round(x); }

/* dsp/dsp.go:72-75 calculateBlocksize */
ORC_API int orc_goertzel_blocksize(double pitch, int sample_rate, double blocksize_ratio)
{
    double minBlocksize = go_round((double)sample_rate / pitch);
    return (int)go_round((blocksize_ratio * (double)sample_rate) / minBlocksize) * (int)minBlocksize;
}

static void goertzel_init(orc_goertzel *f, double pitch, int sampleRate, double ratio)
{
    const double Pi = 3.14159265358979323846264338327950288;
    f->pitch = pitch;
    f->sampleRate = sampleRate;
    f->blocksize = orc_goertzel_blocksize(pitch, sampleRate, ratio);
    int binIndex = (int)(0.5 + ((double)f->blocksize * pitch / (double)sampleRate));
    double omega = 2 * Pi * (double)binIndex / (double)f->blocksize;
    /* math.Cos: Go's pure-Go Cephes cos; shares the polynomial with Sincos */
    double s, c;
    orc_go_sincos(omega, &s, &c);
    f->coeff = 2 * c;
    f->magnitudeLimitLow = (double)f->blocksize / 2;
    f->magnitudeLimit = 0;
    f->magnitudeThreshold = 0.75;
}

/* dsp/dsp.go:98-106 Magnitude */
static double goertzel_magnitude(const orc_goertzel *f, const float *block, int n)
{
    double q0, q1 = 0, q2 = 0;
    for (int i = 0; i < n; i++) {
        q0 = f->coeff * q1 - q2 + (double)block[i];
        q2 = q1;
        q1 = q0;
    }
    return sqrt((q1 * q1) + (q2 * q2) - q1 * q2 * f->coeff);
}

/* dsp/dsp.go:111-123 NormalizedMagnitude */
static double goertzel_normalized(orc_goertzel *f, const float *block, int n)
{
    double magnitude = goertzel_magnitude(f, block, n);
    if (magnitude > f->magnitudeLimitLow)
        f->magnitudeLimit = (f->magnitudeLimit + ((magnitude - f->magnitudeLimit) / 6));
    if (f->magnitudeLimit < f->magnitudeLimitLow)
        f->magnitudeLimit = f->magnitudeLimitLow;
    return magnitude / f->magnitudeLimit;
}

/* ------------------------------------------------------------------------ */
/* cw/decode.go                                                              */
/* ------------------------------------------------------------------------ */

/* Morse table.  The reference takes it from github.com/ftl/digimodes
 * v0.0.0-20231231131023-cffadad68e9e (cw.Code), which is not in the
 * container.  Restated from the published International Morse code (ITU-R
 * M.1677-1) plus the extensions the reference's tests pin: 'ä' (.-.-, from
 * ly2px_4), '§' = 8 dits (decode_test.go:28).  Pinned subset: the entries of
 * cw/decode_test.go:23-29 and every character of the nine recorded streams. */
typedef struct {
    uint32_t rune;
    const char *code;
} morse_entry;

static const morse_entry MORSE[] = {
    {'a', ".-"},     {'b', "-..."},   {'c', "-.-."},   {'d', "-.."},    {'e', "."},      {'f', "..-."},
    {'g', "--."},    {'h', "...."},   {'i', ".."},     {'j', ".---"},   {'k', "-.-"},    {'l', ".-.."},
    {'m', "--"},     {'n', "-."},     {'o', "---"},    {'p', ".--."},   {'q', "--.-"},   {'r', ".-."},
    {'s', "..."},    {'t', "-"},      {'u', "..-"},    {'v', "...-"},   {'w', ".--"},    {'x', "-..-"},
    {'y', "-.--"},   {'z', "--.."},   {'0', "-----"},  {'1', ".----"},  {'2', "..---"},  {'3', "...--"},
    {'4', "....-"},  {'5', "....."},  {'6', "-...."},  {'7', "--..."},  {'8', "---.."},  {'9', "----."},
    {'.', ".-.-.-"}, {',', "--..--"}, {'?', "..--.."}, {'/', "-..-."},  {'=', "-...-"},  {'+', ".-.-."},
    {'-', "-....-"}, {'@', ".--.-."}, {':', "---..."}, {';', "-.-.-."}, {'\'', ".----."}, {'"', ".-..-."},
    {'(', "-.--."},  {')', "-.--.-"}, {'_', "..--.-"}, {'!', "-.-.--"}, {'&', ".-..."},  {'$', "...-..-"},
    {0xE4, ".-.-"},  {0xF6, "---."},  {0xFC, "..--"},  {0xA7, "........"},
};
#define MORSE_COUNT ((int)(sizeof MORSE / sizeof MORSE[0]))

#define ORC_UNKNOWN_RUNE 0xA6u /* cw/decode.go:33 unknownCharacter */
#define MAX_SYMBOLS 8          /* cw/decode.go:36 */
#define SYM_NONE 0
#define SYM_DIT 1
#define SYM_DA 2

ORC_API int orc_morse_count(void) { return MORSE_COUNT; }
ORC_API uint32_t orc_morse_rune(int i) { return MORSE[i].rune; }
ORC_API const char *orc_morse_code(int i) { return MORSE[i].code; }

static const char *morse_lookup_rune(uint32_t r)
{
    for (int i = 0; i < MORSE_COUNT; i++)
        if (MORSE[i].rune == r)
            return MORSE[i].code;
    return NULL;
}

/* cw/decode.go:360-431 AdaptiveThreshold */
typedef struct {
    double preset, upperBound, low, high, last, threshold;
} orc_adaptive;

static void at_update(orc_adaptive *t) { t->threshold = sqrt(t->low * t->high); }
static void at_reset(orc_adaptive *t)
{
    t->low = t->preset;
    t->high = 3 * t->low;
    t->last = t->low;
    at_update(t);
}
static void at_new(orc_adaptive *t, double preset)
{
    t->preset = preset;
    t->upperBound = 10;
    at_reset(t);
}
static void at_preset(orc_adaptive *t, double preset)
{
    t->preset = preset;
    at_reset(t);
}
static void at_put(orc_adaptive *t, double duration)
{
    const double highFactor = 2;
    const double avgWeight = 0.75;
    const double currentWeight = 1.0 - avgWeight;
    if (duration >= t->low * t->upperBound)
        return;
    if (t->last >= duration * highFactor) {
        t->low = avgWeight * t->low + currentWeight * duration;
        t->high = avgWeight * t->high + currentWeight * t->last;
    } else if (duration >= t->last * highFactor) {
        t->low = avgWeight * t->low + currentWeight * t->last;
        t->high = avgWeight * t->high + currentWeight * duration;
    }
    t->last = duration;
    at_update(t);
}

/* cw/decode.go:108-129 Decoder; output collected as runes */
typedef struct {
    double tickSeconds, ticks;
    int lastState;
    double onStart, offStart, wpm;
    int decoding;
    int abortDecodeAfterDits;
    uint8_t currentChar[MAX_SYMBOLS];
    int currentCharInvalid;
    orc_adaptive onThreshold, offThreshold;
    uint32_t *out;
    int out_len, out_cap;
} orc_decoder;

static void dec_write(orc_decoder *d, uint32_t r)
{
    if (d->out_len == d->out_cap) {
        d->out_cap = d->out_cap ? d->out_cap * 2 : 64;
        d->out = (uint32_t *)realloc(d->out, sizeof(uint32_t) * (size_t)d->out_cap);
    }
    d->out[d->out_len++] = r;
}

/* cw/decode.go:191-195 wpmToDit */
static double dec_wpm_to_dit(const orc_decoder *d, double wpm)
{
    double ditSeconds = 60.0 / (50.0 * wpm);
    return ceil(ditSeconds / d->tickSeconds);
}
/* cw/decode.go:197-200 ditToWPM */
static double dec_dit_to_wpm(const orc_decoder *d, double ditTicks)
{
    double ditSeconds = ditTicks * d->tickSeconds;
    return 60.0 / (50.0 * ditSeconds);
}

static void char_clear(uint8_t *c) { memset(c, SYM_NONE, MAX_SYMBOLS); }
static int char_append(uint8_t *c, uint8_t s)
{
    for (int i = 0; i < MAX_SYMBOLS; i++)
        if (c[i] == SYM_NONE) {
            c[i] = s;
            return 1;
        }
    return 0;
}
static int char_empty(const uint8_t *c) { return c[0] == SYM_NONE; }

/* cw/decode.go:131-147 NewDecoder */
static void dec_init(orc_decoder *d, int sampleRate, int blockSize)
{
    memset(d, 0, sizeof *d);
    d->tickSeconds = (double)blockSize / (double)sampleRate;
    d->wpm = 20;
    d->abortDecodeAfterDits = 10;
    char_clear(d->currentChar);
    double dit = dec_wpm_to_dit(d, d->wpm);
    at_new(&d->onThreshold, dit);
    at_new(&d->offThreshold, dit);
}

/* cw/decode.go:172-178 Clear */
static void dec_clear(orc_decoder *d)
{
    d->decoding = 0;
    char_clear(d->currentChar);
    d->ticks = 0;
    d->onStart = 0;
    d->offStart = 0;
}
/* cw/decode.go:180-185 presetWPM */
static void dec_preset_wpm(orc_decoder *d, int wpm)
{
    d->wpm = (double)wpm;
    double dit = dec_wpm_to_dit(d, d->wpm);
    at_preset(&d->onThreshold, dit);
    at_preset(&d->offThreshold, dit);
}
/* cw/decode.go:166-170 Reset (lastState / currentCharInvalid survive, App. C6) */
static void dec_reset(orc_decoder *d)
{
    dec_preset_wpm(d, 20);
    dec_clear(d);
    at_reset(&d->onThreshold);
}

/* cw/decode.go:315-350 decodeCurrentChar */
static void dec_decode_current_char(orc_decoder *d)
{
    if (char_empty(d->currentChar))
        return;
    if (d->currentCharInvalid) {
        d->currentCharInvalid = 0;
        char_clear(d->currentChar);
        dec_write(d, ORC_UNKNOWN_RUNE);
        return;
    }
    char code[MAX_SYMBOLS + 1];
    int n = 0;
    for (; n < MAX_SYMBOLS && d->currentChar[n] != SYM_NONE; n++)
        code[n] = d->currentChar[n] == SYM_DIT ? '.' : '-';
    code[n] = 0;
    uint32_t r = ORC_UNKNOWN_RUNE;
    for (int i = 0; i < MORSE_COUNT; i++)
        if (strcmp(MORSE[i].code, code) == 0) {
            r = MORSE[i].rune;
            break;
        }
    dec_write(d, r);
    char_clear(d->currentChar);
}

/* cw/decode.go:307-313 appendSymbol */
static void dec_append_symbol(orc_decoder *d, uint8_t s)
{
    if (!char_append(d->currentChar, s)) {
        dec_decode_current_char(d);
        char_append(d->currentChar, s);
    }
}

/* cw/decode.go:252-275 onRisingEdge */
static void dec_on_rising_edge(orc_decoder *d, double offDuration)
{
    if (offDuration < 2.0) /* minDitTime */
        return;
    at_put(&d->offThreshold, offDuration);
    double threshold = d->offThreshold.threshold;
    double upperThreshold = 4.5 * d->offThreshold.low;
    if (offDuration >= upperThreshold) {
        dec_decode_current_char(d);
        dec_write(d, ' ');
    } else if (offDuration >= threshold) {
        dec_decode_current_char(d);
    }
}

/* cw/decode.go:277-298 onFallingEdge */
static void dec_on_falling_edge(orc_decoder *d, double onDuration)
{
    if (onDuration < 2.0)
        return;
    at_put(&d->onThreshold, onDuration);
    double threshold = d->onThreshold.threshold;
    double upperThreshold = 2 * d->onThreshold.high;
    if (onDuration >= upperThreshold) {
        d->currentCharInvalid = 1;
    } else if (onDuration >= threshold) {
        dec_append_symbol(d, SYM_DA);
        d->wpm = (d->wpm + dec_dit_to_wpm(d, d->onThreshold.low)) / 2.0;
    } else {
        dec_append_symbol(d, SYM_DIT);
    }
}

/* cw/decode.go:202-250 Tick */
static void dec_tick(orc_decoder *d, int state)
{
    d->ticks++;
    double now = d->ticks;
    if (state != d->lastState) {
        if (state) {
            d->onStart = now;
            dec_on_rising_edge(d, now - d->offStart);
        } else {
            d->offStart = now;
            dec_on_falling_edge(d, now - d->onStart);
        }
        d->decoding = 1;
    }
    d->lastState = state;
    double currentDuration = state ? now - d->onStart : now - d->offStart;
    double upperBound = d->offThreshold.threshold * (double)d->abortDecodeAfterDits;
    if (d->decoding && currentDuration > upperBound) {
        d->decoding = 0;
        dec_decode_current_char(d);
    }
}

ORC_API void *orc_decoder_new(int sample_rate, int block_size)
{
    orc_decoder *d = (orc_decoder *)malloc(sizeof *d);
    dec_init(d, sample_rate, block_size);
    return d;
}
ORC_API void orc_decoder_free(void *h)
{
    free(((orc_decoder *)h)->out);
    free(h);
}
ORC_API void orc_decoder_reset(void *h) { dec_reset((orc_decoder *)h); }
ORC_API void orc_decoder_clear(void *h) { dec_clear((orc_decoder *)h); }
ORC_API void orc_decoder_preset_wpm(void *h, int wpm) { dec_preset_wpm((orc_decoder *)h, wpm); }
ORC_API void orc_decoder_tick(void *h, int state) { dec_tick((orc_decoder *)h, state != 0); }
ORC_API void orc_decoder_ticks(void *h, const uint8_t *states, int n)
{
    for (int i = 0; i < n; i++)
        dec_tick((orc_decoder *)h, states[i] != 0);
}
ORC_API void orc_decoder_stop(void *h) { dec_decode_current_char((orc_decoder *)h); } /* decode.go:352-354 */
ORC_API double orc_decoder_wpm(void *h) { return ((orc_decoder *)h)->wpm; }
ORC_API int orc_decoder_out_len(void *h) { return ((orc_decoder *)h)->out_len; }
ORC_API const uint32_t *orc_decoder_out(void *h) { return ((orc_decoder *)h)->out; }
ORC_API void orc_decoder_out_reset(void *h) { ((orc_decoder *)h)->out_len = 0; }
/* state snapshot for device parity: low/high/last/threshold of both thresholds */
ORC_API void orc_decoder_state(void *h, double *out12)
{
    orc_decoder *d = (orc_decoder *)h;
    out12[0] = d->ticks; out12[1] = d->onStart; out12[2] = d->offStart; out12[3] = d->wpm;
    out12[4] = d->onThreshold.low; out12[5] = d->onThreshold.high; out12[6] = d->onThreshold.last;
    out12[7] = d->onThreshold.threshold;
    out12[8] = d->offThreshold.low; out12[9] = d->offThreshold.high; out12[10] = d->offThreshold.last;
    out12[11] = d->offThreshold.threshold;
}

/* Morse symbol stream generator following cw/decode_test.go:255-287
 * generateStream (with digimodes' WPMToDit = 60s/(50*wpm) in integer
 * nanoseconds and WriteToSymbolStream's dit / da / symbol / char / word break
 * sequence).  timing = {dit, da, symbolBreak, charBreak, wordBreak} units.
 * Returns the number of ticks written (or needed, if > cap). */
ORC_API int orc_generate_stream(int sample_rate, int block_size, int wpm, const int *timing,
                                const uint32_t *text, int text_len, uint8_t *out, int cap)
{
    double tickSeconds = (double)block_size / (double)sample_rate;
    int64_t dit_ns = (int64_t)60 * 1000000000LL / (int64_t)(50 * wpm);
    int64_t tick_ns = (int64_t)(tickSeconds * 1e9);
    int baseTicks = (int)(dit_ns / tick_ns);
    int n = 0;
#define EMIT(v, cnt)                       \
    do {                                   \
        for (int _i = 0; _i < (cnt); _i++) { \
            if (n < cap) out[n] = (v);     \
            n++;                           \
        }                                  \
    } while (0)
    int pending_char_break = 0;
    for (int t = 0; t < text_len; t++) {
        uint32_t r = text[t];
        if (r == ' ') {
            EMIT(0, baseTicks * timing[4]);
            pending_char_break = 0;
            continue;
        }
        const char *code = morse_lookup_rune(r);
        if (!code)
            continue;
        if (pending_char_break)
            EMIT(0, baseTicks * timing[3]);
        for (int i = 0; code[i]; i++) {
            if (i > 0)
                EMIT(0, baseTicks * timing[2]);
            EMIT(1, baseTicks * (code[i] == '.' ? timing[0] : timing[1]));
        }
        pending_char_break = 1;
    }
    EMIT(0, 3 * baseTicks * timing[4]);
#undef EMIT
    return n;
}

/* ------------------------------------------------------------------------ */
/* cw/audio.go AudioDemodulator (config C1)                                  */
/* ------------------------------------------------------------------------ */

typedef struct {
    orc_goertzel filter;
    orc_debouncer debouncer;
    orc_decoder decoder;
    double maxScale;
    float scale;
    float *block;
    int fill;
} orc_audio;

/* cw/audio.go:37-58 NewAudioDemodulator */
ORC_API void *orc_audio_new(double pitch, int sample_rate)
{
    orc_audio *a = (orc_audio *)calloc(1, sizeof *a);
    goertzel_init(&a->filter, pitch, sample_rate, 0.005);
    deb_init(&a->debouncer, 3);
    a->maxScale = 12;
    a->scale = 1;
    dec_init(&a->decoder, sample_rate, a->filter.blocksize);
    a->block = (float *)malloc(sizeof(float) * (size_t)a->filter.blocksize);
    a->fill = 0;
    return a;
}
ORC_API void orc_audio_free(void *h)
{
    orc_audio *a = (orc_audio *)h;
    free(a->block);
    free(a->decoder.out);
    free(a);
}
ORC_API int orc_audio_blocksize(void *h) { return ((orc_audio *)h)->filter.blocksize; }
ORC_API double orc_audio_coeff(void *h) { return ((orc_audio *)h)->filter.coeff; }
ORC_API void orc_audio_set_scale(void *h, double s) { ((orc_audio *)h)->scale = (float)s; }
ORC_API void orc_audio_set_debounce(void *h, int t) { ((orc_audio *)h)->debouncer.threshold = t; }
ORC_API void orc_audio_set_magnitude_threshold(void *h, double t) { ((orc_audio *)h)->filter.magnitudeThreshold = t; }

/* cw/audio.go:213-221 truncate */
static float truncate1(float v) { return v > 1 ? 1 : (v < -1 ? -1 : v); }

/* cw/audio.go:169-211 run (sample loop) + cw/audio.go:149-158 Write (mono).
 * Optional per-block outputs: normalised magnitude, raw state, debounced. */
ORC_API int orc_audio_write(void *h, const float *samples, int n, double *mags, uint8_t *raw, uint8_t *deb, int cap)
{
    orc_audio *a = (orc_audio *)h;
    int bs = a->filter.blocksize;
    int nb = 0;
    for (int i = 0; i < n; i++) {
        a->block[a->fill++] = samples[i];
        if (a->fill < bs)
            continue;
        float scale = a->scale;
        if (scale == 0) {
            /* dsp/dsp.go:19-28 FilterBlock.Max */
            float mx = 0;
            for (int k = 0; k < bs; k++) {
                float ab = (float)fabs((double)a->block[k]);
                if (ab > mx)
                    mx = ab;
            }
            double inv = 1 / (double)mx;
            scale = (float)(inv < a->maxScale ? inv : a->maxScale); /* math.Min */
        }
        if (scale != 1)
            for (int k = 0; k < bs; k++)
                a->block[k] = truncate1(a->block[k] * scale);
        double magnitude = goertzel_normalized(&a->filter, a->block, bs);
        int state = magnitude > a->filter.magnitudeThreshold;
        a->fill = 0;
        int debounced = deb_debounce(&a->debouncer, state);
        dec_tick(&a->decoder, debounced);
        if (nb < cap) {
            if (mags) mags[nb] = magnitude;
            if (raw) raw[nb] = (uint8_t)state;
            if (deb) deb[nb] = (uint8_t)debounced;
        }
        nb++;
    }
    return nb;
}
ORC_API void orc_audio_close(void *h) { dec_decode_current_char(&((orc_audio *)h)->decoder); } /* audio.go:205-207 */
ORC_API int orc_audio_out_len(void *h) { return ((orc_audio *)h)->decoder.out_len; }
ORC_API const uint32_t *orc_audio_out(void *h) { return ((orc_audio *)h)->decoder.out; }

/* ------------------------------------------------------------------------ */
/* rx/peaks.go PeaksTable (host bookkeeping; pinned by rx/peaks_test.go)     */
/* ------------------------------------------------------------------------ */

enum { PK_NONE = 0, PK_NEW = 1, PK_ACTIVE = 2, PK_INACTIVE = 3 };

typedef struct {
    orc_peak peak;
    int state;
    double since;
    int used;
} pt_entry;

typedef struct {
    int size;
    int *bins; /* index into entries, -1 = nil */
    pt_entry *entries;
    int n_entries, cap_entries;
    double now;         /* manual clock, seconds */
    double peakTimeout; /* rx/peaks.go:11 = 120 s */
    uint64_t rng;       /* deterministic stand-in for math/rand */
} orc_peaks_table;

ORC_API void *orc_peaks_table_new(int size)
{
    orc_peaks_table *t = (orc_peaks_table *)calloc(1, sizeof *t);
    t->size = size;
    t->bins = (int *)malloc(sizeof(int) * (size_t)size);
    for (int i = 0; i < size; i++)
        t->bins[i] = -1;
    t->peakTimeout = 120.0;
    t->rng = 0x9E3779B97F4A7C15ull;
    return t;
}
ORC_API void orc_peaks_table_free(void *h)
{
    orc_peaks_table *t = (orc_peaks_table *)h;
    free(t->bins);
    free(t->entries);
    free(t);
}
ORC_API void orc_peaks_table_set_now(void *h, double now) { ((orc_peaks_table *)h)->now = now; }
ORC_API void orc_peaks_table_seed(void *h, uint64_t seed) { ((orc_peaks_table *)h)->rng = seed ? seed : 1; }

static int pt_new_entry(orc_peaks_table *t, const orc_peak *p, int state)
{
    if (t->n_entries == t->cap_entries) {
        t->cap_entries = t->cap_entries ? t->cap_entries * 2 : 64;
        t->entries = (pt_entry *)realloc(t->entries, sizeof(pt_entry) * (size_t)t->cap_entries);
    }
    pt_entry *e = &t->entries[t->n_entries];
    e->peak = *p;
    e->state = state;
    e->since = t->now;
    e->used = 1;
    return t->n_entries++;
}
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

/* rx/peaks.go:121-125 clear */
static void pt_clear(orc_peaks_table *t, int from, int to)
{
    for (int i = imax(0, from); i <= imin(to, t->size - 1); i++)
        t->bins[i] = -1;
}
/* rx/peaks.go:115-119 put */
static void pt_put_internal(orc_peaks_table *t, int e)
{
    const orc_peak *p = &t->entries[e].peak;
    for (int i = imax(0, p->from); i <= imin(p->to, t->size - 1); i++)
        t->bins[i] = e;
}

/* rx/peaks.go:73-103 Put / :46-71 ForcePut.  Returns entry id or -1 if refused. */
static int pt_put(orc_peaks_table *t, const orc_peak *p, int force, int state)
{
    int clearFrom = -1, clearTo = -1;
    for (int i = imax(0, p->from); i <= imin(p->to, t->size - 1); i++) {
        int e = t->bins[i];
        if (e < 0)
            continue;
        if (!force && (t->entries[e].state == PK_ACTIVE || t->entries[e].state == PK_INACTIVE))
            return -1;
        if (clearFrom == -1)
            clearFrom = t->entries[e].peak.from;
        clearTo = t->entries[e].peak.to;
    }
    if (clearFrom > -1 && clearTo > -1)
        pt_clear(t, clearFrom, clearTo);
    int e = pt_new_entry(t, p, state);
    pt_put_internal(t, e);
    return e;
}
ORC_API int orc_peaks_table_put(void *h, int from, int to)
{
    orc_peak p;
    memset(&p, 0, sizeof p);
    p.from = from;
    p.to = to;
    return pt_put((orc_peaks_table *)h, &p, 0, PK_NEW);
}
ORC_API int orc_peaks_table_force_put(void *h, int from, int to)
{
    orc_peak p;
    memset(&p, 0, sizeof p);
    p.from = from;
    p.to = to;
    return pt_put((orc_peaks_table *)h, &p, 1, PK_NEW);
}
/* test helper: place an entry with a given state without overlap rules */
ORC_API int orc_peaks_table_place(void *h, int from, int to, int state)
{
    orc_peaks_table *t = (orc_peaks_table *)h;
    orc_peak p;
    memset(&p, 0, sizeof p);
    p.from = from;
    p.to = to;
    int e = pt_new_entry(t, &p, state);
    pt_put_internal(t, e);
    return e;
}
/* entry id at bin, -1 = nil */
ORC_API int orc_peaks_table_at(void *h, int bin)
{
    orc_peaks_table *t = (orc_peaks_table *)h;
    if (bin < 0 || bin >= t->size)
        return -1;
    return t->bins[bin];
}
ORC_API int orc_peaks_table_state(void *h, int entry) { return ((orc_peaks_table *)h)->entries[entry].state; }

/* rx/peaks.go:127-147 Cleanup */
ORC_API void orc_peaks_table_cleanup(void *h)
{
    orc_peaks_table *t = (orc_peaks_table *)h;
    int i = 0;
    while (i < t->size) {
        int e = t->bins[i];
        i++;
        if (e < 0)
            continue;
        if (t->entries[e].state == PK_ACTIVE)
            continue;
        if (t->now - t->entries[e].since < t->peakTimeout)
            continue;
        pt_clear(t, t->entries[e].peak.from, t->entries[e].peak.to);
        i = t->entries[e].peak.to + 1;
    }
}
/* rx/peaks.go:161-171 getInternal */
static int pt_get_internal(orc_peaks_table *t, int from, int to)
{
    int e = t->bins[from];
    if (e < 0)
        return -1;
    if (t->entries[e].peak.to != to)
        return -1;
    return e;
}
/* rx/peaks.go:153-159 Activate */
ORC_API void orc_peaks_table_activate(void *h, int from, int to)
{
    orc_peaks_table *t = (orc_peaks_table *)h;
    int e = pt_get_internal(t, from, to);
    if (e < 0)
        return; /* the reference would nil-deref here (App. C4) */
    if (t->entries[e].state != PK_NEW && t->entries[e].state != PK_INACTIVE)
        return;
    t->entries[e].state = PK_ACTIVE;
}
/* rx/peaks.go:174-181 Deactivate */
ORC_API void orc_peaks_table_deactivate(void *h, int from, int to)
{
    orc_peaks_table *t = (orc_peaks_table *)h;
    int e = pt_get_internal(t, from, to);
    if (e < 0)
        return;
    if (t->entries[e].state != PK_ACTIVE)
        return;
    t->entries[e].state = PK_INACTIVE;
}
static uint64_t pt_rand(orc_peaks_table *t)
{
    /* xorshift64*: deterministic stand-in for the reference's unseeded
     * global math/rand (rx/peaks.go:185), see SURVEY.md App. C5 */
    uint64_t x = t->rng;
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    t->rng = x;
    return x * 0x2545F4914F6CDD1Dull;
}
/* rx/peaks.go:183-207 FindNext: size/2 random probes, then a linear scan */
ORC_API int orc_peaks_table_find_next(void *h)
{
    orc_peaks_table *t = (orc_peaks_table *)h;
    for (int k = 0; k < t->size / 2; k++) {
        int i = (int)(pt_rand(t) % (uint64_t)t->size);
        int e = t->bins[i];
        if (e < 0 || t->entries[e].state != PK_NEW)
            continue;
        return e;
    }
    for (int i = 0; i < t->size; i++) {
        int e = t->bins[i];
        if (e < 0 || t->entries[e].state != PK_NEW)
            continue;
        return e;
    }
    return -1;
}
ORC_API void orc_peaks_table_entry(void *h, int e, int *from, int *to)
{
    orc_peaks_table *t = (orc_peaks_table *)h;
    *from = t->entries[e].peak.from;
    *to = t->entries[e].peak.to;
}

/* ------------------------------------------------------------------------ */
/* rx/receiver.go run loop, one band, explicit listener set                  */
/* ------------------------------------------------------------------------ */

#define ORC_CUMULATION_SIZE 100 /* rx/receiver.go:18 */
#define ORC_NOISE_WINDOW 60     /* rx/receiver.go:21 */

typedef struct {
    int bin;
    int attached;
    orc_debouncer deb;
    orc_decoder dec;
} orc_listener;

typedef struct {
    int sampleRate, blockSize, edgeWidth, debounce;
    float peakThreshold;
    int64_t centerFrequency;
    orc_rolling_mean nfMean, devMean;
    float *spectrum, *psd, *cumulation;
    int cumulationCount;
    orc_listener *listeners;
    int n_listeners, cap_listeners;
    int find_peaks_enabled;
    long frames;
} orc_receiver;

/* per-frame record for parity */
typedef struct {
    float min_mean;     /* FindNoiseFloor T(minValue) */
    double variance;    /* FindNoiseFloor variance */
    float dev_in;       /* input to noiseDeviationMean.Put */
    float nf_in;        /* input to noiseFloorMean.Put */
    float noise_dev;    /* noiseDeviation */
    float noise_floor;  /* noiseFloor */
    float peak_thr;     /* peakThreshold */
    float listen_thr;   /* noiseFloor+noiseDeviation */
} orc_frame_rec;

ORC_API void *orc_receiver_new(int sample_rate, int block_size, int edge_width, float peak_threshold,
                               int debounce, int64_t center_frequency)
{
    orc_receiver *r = (orc_receiver *)calloc(1, sizeof *r);
    r->sampleRate = sample_rate;
    r->blockSize = block_size;
    r->edgeWidth = edge_width;
    r->peakThreshold = peak_threshold;
    r->debounce = debounce;
    r->centerFrequency = center_frequency;
    rm_init(&r->nfMean, ORC_NOISE_WINDOW);
    rm_init(&r->devMean, ORC_NOISE_WINDOW);
    r->spectrum = (float *)calloc((size_t)block_size, sizeof(float));
    r->psd = (float *)calloc((size_t)block_size, sizeof(float));
    r->cumulation = (float *)calloc((size_t)block_size, sizeof(float));
    r->find_peaks_enabled = 1;
    return r;
}
ORC_API void orc_receiver_free(void *h)
{
    orc_receiver *r = (orc_receiver *)h;
    for (int i = 0; i < r->n_listeners; i++)
        free(r->listeners[i].dec.out);
    free(r->listeners);
    free(r->nfMean.values);
    free(r->devMean.values);
    free(r->spectrum);
    free(r->psd);
    free(r->cumulation);
    free(r);
}
ORC_API void orc_receiver_set_peak_threshold(void *h, float t) { ((orc_receiver *)h)->peakThreshold = t; }
ORC_API void orc_receiver_set_edge_width(void *h, int e) { ((orc_receiver *)h)->edgeWidth = e; }
ORC_API void orc_receiver_set_find_peaks(void *h, int on) { ((orc_receiver *)h)->find_peaks_enabled = on; }

/* Attach a fresh listener to `bin` (rx/listener.go:84-94 Attach on a newly
 * bound listener: new SpectralDemodulator => new debouncer (threshold =
 * receiver's signal debounce) and new Decoder, then Reset). Returns its id. */
ORC_API int orc_receiver_attach(void *h, int bin)
{
    orc_receiver *r = (orc_receiver *)h;
    if (r->n_listeners == r->cap_listeners) {
        r->cap_listeners = r->cap_listeners ? 2 * r->cap_listeners : 16;
        r->listeners = (orc_listener *)realloc(r->listeners, sizeof(orc_listener) * (size_t)r->cap_listeners);
    }
    orc_listener *l = &r->listeners[r->n_listeners];
    l->bin = bin;
    l->attached = 1;
    deb_init(&l->deb, r->debounce);
    dec_init(&l->dec, r->sampleRate, r->blockSize);
    dec_reset(&l->dec);
    return r->n_listeners++;
}
ORC_API void orc_receiver_detach(void *h, int id) { ((orc_receiver *)h)->listeners[id].attached = 0; }
ORC_API int orc_receiver_text_len(void *h, int id) { return ((orc_receiver *)h)->listeners[id].dec.out_len; }
ORC_API const uint32_t *orc_receiver_text(void *h, int id) { return ((orc_receiver *)h)->listeners[id].dec.out; }
ORC_API void orc_receiver_decoder_state(void *h, int id, double *out12)
{
    orc_decoder_state(&((orc_receiver *)h)->listeners[id].dec, out12);
}
ORC_API const float *orc_receiver_cumulation(void *h) { return ((orc_receiver *)h)->cumulation; }
ORC_API int orc_receiver_cumulation_count(void *h) { return ((orc_receiver *)h)->cumulationCount; }

/* rx/receiver.go:353-463 frame case of Receiver.run, for n_frames frames.
 * Outputs (any may be NULL):
 *   frame_recs[n_frames]
 *   spectrum_out / psd_out [n_frames][N]
 *   values[n_frames][L]  — spectrum[l.SignalBin()] handed to Listen
 *   raw[n_frames][L], deb[n_frames][L] — state before / after the debouncer
 *     (L = listener count at call time; detached listeners record 0)
 *   peaks[max_chunks][max_peaks], peak_counts[max_chunks], peak_frames[max_chunks]:
 *     FindPeaks result of every cumulation that completes inside this call
 * Returns number of completed cumulations. */
ORC_API int orc_receiver_process(void *h, const float *iq, int n_frames, orc_frame_rec *frame_recs,
                                 float *spectrum_out, float *psd_out, float *values, uint8_t *raw, uint8_t *deb,
                                 orc_peak *peaks, int *peak_counts, int *peak_frames, int max_chunks, int max_peaks,
                                 float *cumulation_out)
{
    orc_receiver *r = (orc_receiver *)h;
    int N = r->blockSize;
    int L = r->n_listeners;
    int chunks = 0;
    for (int f = 0; f < n_frames; f++) {
        const float *frame = iq + (size_t)f * 2 * (size_t)N;
        orc_iq_to_spectrum_and_psd(N, frame, r->spectrum, r->psd); /* receiver.go:376-379 */
        float psdNoiseFloor;
        double noiseVariance;
        orc_find_noise_floor(r->psd, N, r->edgeWidth, &psdNoiseFloor, &noiseVariance); /* :381 */
        /* :383 */
        float dev_in = (float)((double)(orc_psd_value_in_db((float)sqrt(noiseVariance), N) + 120.0f) * 0.25);
        float noiseDeviation = rm_put(&r->devMean, dev_in);
        /* :384 */
        float nf_in = orc_psd_value_in_db(psdNoiseFloor, N) + 120.0f;
        float noiseFloor = rm_put(&r->nfMean, nf_in);
        float peakThreshold = r->peakThreshold + noiseFloor; /* :385 */
        float listenThr = noiseFloor + noiseDeviation;        /* :394 */
        if (frame_recs) {
            orc_frame_rec *fr = &frame_recs[f];
            fr->min_mean = psdNoiseFloor;
            fr->variance = noiseVariance;
            fr->dev_in = dev_in;
            fr->nf_in = nf_in;
            fr->noise_dev = noiseDeviation;
            fr->noise_floor = noiseFloor;
            fr->peak_thr = peakThreshold;
            fr->listen_thr = listenThr;
        }
        if (spectrum_out)
            memcpy(spectrum_out + (size_t)f * N, r->spectrum, sizeof(float) * (size_t)N);
        if (psd_out)
            memcpy(psd_out + (size_t)f * N, r->psd, sizeof(float) * (size_t)N);
        /* :388-402 listener fan-out -> listener.go:142 -> spectral.go:48-54 */
        for (int l = 0; l < L; l++) {
            orc_listener *ls = &r->listeners[l];
            float v = 0;
            int st = 0, db = 0;
            if (ls->attached) {
                v = r->spectrum[ls->bin];
                st = v > listenThr;
                db = deb_debounce(&ls->deb, st);
                dec_tick(&ls->dec, db);
            }
            if (values) values[(size_t)f * L + l] = v;
            if (raw) raw[(size_t)f * L + l] = (uint8_t)st;
            if (deb) deb[(size_t)f * L + l] = (uint8_t)db;
        }
        /* :404-407 */
        for (int i = 0; i < N; i++)
            r->cumulation[i] += r->spectrum[i];
        r->cumulationCount++;
        /* :409-461 */
        if (r->cumulationCount == ORC_CUMULATION_SIZE) {
            if (chunks < max_chunks) {
                int np = 0;
                if (r->find_peaks_enabled && peaks)
                    np = orc_find_peaks(r->cumulation, N, ORC_CUMULATION_SIZE, peakThreshold, r->sampleRate,
                                        r->centerFrequency, peaks + (size_t)chunks * max_peaks, max_peaks);
                if (peak_counts) peak_counts[chunks] = np;
                if (peak_frames) peak_frames[chunks] = f;
                if (cumulation_out)
                    memcpy(cumulation_out + (size_t)chunks * N, r->cumulation, sizeof(float) * (size_t)N);
            }
            chunks++;
            memset(r->cumulation, 0, sizeof(float) * (size_t)N);
            r->cumulationCount = 0;
        }
        r->frames++;
    }
    return chunks;
}

/* CPU-baseline entry: the timed hot loop without any parity outputs
 * (FFT+projection, noise floor, thresholds, listeners, cumulation, peak scan). */
ORC_API int orc_receiver_run_baseline(void *h, const float *iq, int n_frames)
{
    /* per call, not static: bench.py runs one receiver per host thread through this entry */
    orc_peak *scratch = (orc_peak *)malloc(sizeof(orc_peak) * 4096);
    int counts[64], frames_[64];
    int total = 0;
    int done = 0;
    while (done < n_frames) {
        int n = n_frames - done;
        if (n > 64 * ORC_CUMULATION_SIZE - 1)
            n = 64 * ORC_CUMULATION_SIZE - 1;
        total += orc_receiver_process(h, iq + (size_t)done * 2 * ((orc_receiver *)h)->blockSize, n, NULL, NULL, NULL,
                                      NULL, NULL, NULL, scratch, counts, frames_, 64, 64, NULL);
        done += n;
    }
    free(scratch);
    return total;
}
