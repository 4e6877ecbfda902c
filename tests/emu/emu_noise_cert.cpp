// CPU check of noise_cert.h: FindNoiseFloor "exact where it is consumed".  For every generated psd row the literal
// algorithm (exact_frame: the oracle's loops) and the certification from order-free sums - formed the way k_psd_scan
// forms them: a lane's values in sequence, six butterfly levels across 64 lanes - are compared: a frame certify() ACCEPTS
// must have the reference's min_mean, nf_in, dev_in bits and winning window; a frame it rejects goes to the literal loops
// on the device, so rejections only cost time - their rate is reported and bounded.  TEST ONLY.
//
// Rows: exponential noise (what a PSD bin of Gaussian noise is), noise with carriers 75 dB up in some or all windows,
// constant rows, rows alternating between two neighbouring float32 values (their mean sits EXACTLY on a float32 rounding
// boundary: must be rejected or right), equal windows in different orders (ties), zeros, subnormals, one huge value,
// infinities and NaNs (must be rejected).
//
// usage: emu_noise_cert [rows per kind]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../sdrainer_amd/csrc/noise_cert.h"

using namespace noise;

static Geom geom(int n, int edge)
{
    Geom g;
    g.n = n;
    g.edge = edge;
    g.window = (n - 2 * edge) / 10;
    g.n_windows = (g.window > 0 && (n - 2 * edge) > 10 * g.window) ? 10 : 9;
    g.inv_n2 = 1.0 / ((double)n * (double)n);
    return g;
}

// the scan's sums of window w: lane l adds x[b + l + 64 j] for j = 0, 1, ... in sequence, then xor-butterfly over the lanes
static void scan_sums(const Geom &g, const float *x, int w, double *s1, double *s2)
{
    double a[64], b[64];
    const int b0 = g.edge + w * g.window;
    for (int l = 0; l < 64; l++) {
        a[l] = b[l] = 0.0;
        for (int j = 0; l + 64 * j < g.window; j++) {
            const double xd = (double)x[b0 + l + 64 * j];
            a[l] += xd;
            b[l] += xd * xd;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        double na[64], nb[64];
        for (int l = 0; l < 64; l++) {
            na[l] = a[l] + a[l ^ o];
            nb[l] = b[l] + b[l ^ o];
        }
        memcpy(a, na, sizeof a);
        memcpy(b, nb, sizeof b);
    }
    *s1 = a[0];
    *s2 = b[0];
}

struct Tally {
    long rows = 0, accepted = 0, wrong = 0;
};

static bool same_bits(float a, float b) { return memcmp(&a, &b, 4) == 0; }

static void run_row(const Geom &g, const std::vector<float> &x, Tally &t, const char *kind, bool must_reject)
{
    auto x_at = [&](int i) { return (double)x[i]; };
    double sums[kMaxWindows] = {}, s1[kMaxWindows] = {}, s2[kMaxWindows] = {};
    for (int w = 0; w < g.n_windows; w++) {
        sums[w] = window_sum(g, x_at, w);
        scan_sums(g, x.data(), w, &s1[w], &s2[w]);
    }
    const Result ex = exact_frame(g, x_at, sums);
    const Result ce = certify(s1, s2, g, x_at);
    t.rows++;
    if (!ce.ok)
        return;
    t.accepted++;
    if (must_reject) {
        printf("%s: a row that must take the literal loops was accepted\n", kind);
        t.wrong++;
        return;
    }
    if (ce.window != ex.window || !same_bits(ce.min_mean, ex.min_mean) || !same_bits(ce.nf_in, ex.nf_in) || !same_bits(ce.dev_in, ex.dev_in)) {
        if (t.wrong < 5)
            printf("%s n=%d: accepted but window %d/%d min_mean %a/%a nf_in %a/%a dev_in %a/%a\n", kind, g.n, ce.window, ex.window, ce.min_mean,
                   ex.min_mean, ce.nf_in, ex.nf_in, ce.dev_in, ex.dev_in);
        t.wrong++;
    }
    // the bracket's midpoint is within the bracket's width of the reference's variance
    if (!(fabs(ce.variance - ex.variance) <= 2e-11 * ex.variance)) {
        if (t.wrong < 5)
            printf("%s n=%d: variance %.17g against %.17g\n", kind, g.n, ce.variance, ex.variance);
        t.wrong++;
    }
}

int main(int argc, char **argv)
{
    const int per = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937_64 rng(12345);
    std::exponential_distribution<double> expo(1.0);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    long total_wrong = 0;
    const int sizes[][2] = {{512, 70}, {1024, 140}, {4096, 560}, {8192, 1120}, {16384, 2240}, {16384, 100}, {2048, 0}, {4096, 1900}};
    for (const auto &sz : sizes) {
        const Geom g = geom(sz[0], sz[1]);
        if (g.window < 1)
            continue;
        const int n = g.n;
        const int rows = per * (n <= 1024 ? 20 : n <= 4096 ? 4 : 1);
        Tally noise_t, carr_t, edge_t, tie_t, bad_t;
        std::vector<float> x((size_t)n);
        for (int r = 0; r < rows; r++) {
            const double scale = ::pow(10.0, uni(rng) * 12.0 - 9.0);
            for (auto &v : x)
                v = (float)(scale * expo(rng));
            run_row(g, x, noise_t, "noise", false);
            // carriers 75 dB (3e7 in power) up, in a random subset of the windows (config 3 keeps one window free)
            const unsigned mask = (unsigned)rng();
            for (int w = 0; w < 10; w++)
                if ((mask >> w) & 1)
                    for (int k = 0; k < 3; k++)
                        x[(size_t)(g.edge + w * g.window + (int)(uni(rng) * g.window)) % n] = (float)(scale * 3e7 * (0.5 + uni(rng)));
            run_row(g, x, carr_t, "carriers", false);
        }
        for (int r = 0; r < per; r++) {
            // constant rows; rows alternating between a float32 and its successor (the mean is a rounding boundary)
            const float c = (float)::pow(10.0, uni(rng) * 20.0 - 10.0);
            for (auto &v : x)
                v = c;
            run_row(g, x, edge_t, "constant", false);
            const float c2 = ::nextafterf(c, INFINITY);
            for (int i = 0; i < n; i++)
                x[(size_t)i] = ((i - g.edge) & 1) ? c2 : c;
            run_row(g, x, edge_t, "boundary", false);
            // zeros and subnormals with a few ordinary values
            for (auto &v : x)
                v = uni(rng) < 0.5 ? 0.0f : 1e-42f;
            x[(size_t)(g.edge + 3)] = 1e-3f;
            run_row(g, x, edge_t, "tiny", false);
            // one huge value in an otherwise ordinary row
            for (auto &v : x)
                v = (float)expo(rng);
            x[(size_t)(g.edge + (int)(uni(rng) * 9 * g.window))] = 3e37f;
            run_row(g, x, edge_t, "huge", false);
            // ties: every window holds the same values, shuffled (the sums differ in their last bits at most)
            std::vector<float> base((size_t)g.window);
            for (auto &v : base)
                v = (float)expo(rng);
            for (auto &v : x)
                v = (float)expo(rng);
            for (int w = 0; w < g.n_windows; w++) {
                std::shuffle(base.begin(), base.end(), rng);
                memcpy(&x[(size_t)(g.edge + w * g.window)], base.data(), sizeof(float) * (size_t)g.window);
            }
            run_row(g, x, tie_t, "ties", false);
            // infinities and NaNs inside the windows: never accepted
            for (auto &v : x)
                v = (float)expo(rng);
            x[(size_t)(g.edge + (int)(uni(rng) * g.window * g.n_windows))] = (r & 1) ? INFINITY : NAN;
            run_row(g, x, bad_t, "special", true);
        }
        printf("n=%5d edge=%4d W=%4d windows=%2d: noise %ld/%ld accepted, carriers %ld/%ld, edge cases %ld/%ld, ties %ld/%ld, specials %ld/%ld; wrong %ld\n",
               n, g.edge, g.window, g.n_windows, noise_t.accepted, noise_t.rows, carr_t.accepted, carr_t.rows, edge_t.accepted, edge_t.rows,
               tie_t.accepted, tie_t.rows, bad_t.accepted, bad_t.rows, noise_t.wrong + carr_t.wrong + edge_t.wrong + tie_t.wrong + bad_t.wrong);
        total_wrong += noise_t.wrong + carr_t.wrong + edge_t.wrong + tie_t.wrong + bad_t.wrong;
        // the point of the exercise: nearly every ordinary row is accepted
        if (noise_t.accepted * 1000 < noise_t.rows * 995 || carr_t.accepted * 1000 < carr_t.rows * 995) {
            printf("n=%d: too many ordinary rows rejected\n", n);
            total_wrong++;
        }
    }
    printf("%ld violations\n", total_wrong);
    return total_wrong != 0;
}
