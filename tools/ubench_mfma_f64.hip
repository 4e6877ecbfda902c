// Development aid: what does the float64 matrix pipe compute, bit for bit, and how fast is a dependent chain on it?
//
// Why: the variance chain of FindNoiseFloor (dsp/fft.go:244-249) is up to 16384 strictly ordered float64 additions
// per frame, and a dependent v_add_f64 costs ~10 clocks.  D = C + A x B on the matrix pipe with B = all ones adds
// FOUR terms to each accumulator per instruction.  It can stand in for four ordered additions only if the hardware
// evaluates fma(a3,1, fma(a2,1, fma(a1,1, fma(a0,1,c)))) - each step rounded to float64, in a known order.  This
// probe finds the operand layout empirically (one-hot inputs), then checks random wide-range data against every
// candidate evaluation order, and times dependent chains.  Nothing here is used by the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/bin/ubench_mfma_f64 tools/ubench_mfma_f64.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_4x4(const double *a, const double *b, const double *c, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

__global__ void k_16x16(const double *a, const double *b, const double *c, double *d)
{
    const int l = threadIdx.x;
    d4 acc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; i++)
        d[4 * l + i] = acc[i];
}

// dependent chains: n instructions, each feeding the next one's accumulator
__global__ void k_chain_4x4(const double *a, double *out, int n, unsigned long long *cycles)
{
    const int l = threadIdx.x;
    double acc = 0.0;
    const double one = 1.0;
    double x = a[l];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(x, one, acc, 0, 0, 0);
            asm volatile("" : "+v"(x));
        }
    }
    asm volatile("" : "+v"(acc));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = acc;
    if (l == 0)
        cycles[blockIdx.x] = t1 - t0;
}

__global__ void k_chain_16x16(const double *a, double *out, int n, unsigned long long *cycles)
{
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    const double one = 1.0;
    double x = a[l];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, one, acc, 0, 0, 0);
            asm volatile("" : "+v"(x));
        }
    }
    asm volatile("" : "+v"(acc));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = acc[0] + acc[1] + acc[2] + acc[3];
    if (l == 0)
        cycles[blockIdx.x] = t1 - t0;
}

__global__ void k_chain_valu(const double *a, double *out, int n, unsigned long long *cycles)
{
    const int l = threadIdx.x;
    double acc = 0.0;
    double x = a[l];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc = acc + x;
            asm volatile("" : "+v"(x), "+v"(acc));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = acc;
    if (l == 0)
        cycles[blockIdx.x] = t1 - t0;
}

static double chain(const double *t, const int *ord, int n, double c)
{
    double s = c;
    for (int i = 0; i < n; i++)
        s = fma(t[ord[i]], 1.0, s);
    return s;
}

template <class Launch>
static void probe(const char *name, int n_out, Launch run)
{
    // layout: which A lanes reach which output element (B = ones, C = 0, A one-hot)
    std::vector<std::vector<int>> src(n_out);
    std::vector<double> a(64), b(64, 1.0), c(n_out, 0.0), d(n_out);
    for (int l = 0; l < 64; l++) {
        std::fill(a.begin(), a.end(), 0.0);
        a[l] = 1.0;
        run(a, b, c, d);
        for (int o = 0; o < n_out; o++)
            if (d[o] != 0.0)
                src[o].push_back(l);
    }
    bool four = true;
    for (auto &v : src)
        four = four && v.size() == 4;
    printf("%s: every output element sums %s A lanes (B = ones); output 0 <- lanes", name, four ? "exactly four" : "NOT four");
    for (int l : src[0])
        printf(" %d", l);
    printf("; output 1 <- lanes");
    for (int l : src[1])
        printf(" %d", l);
    printf("\n");
    if (n_out == 64) {
        printf("%s: first source lane of every output lane:", name);
        for (int o = 0; o < n_out; o++)
            printf(" %d", src[o].empty() ? -1 : src[o][0]);
        printf("\n");
    }
    if (!four)
        return;
    // evaluation order: random data with a wide exponent range, so that every association rounds differently
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> mant(1.0, 2.0);
    std::uniform_int_distribution<int> ex(-30, 30);
    int perm[24][4], np = 0;
    int p[4] = {0, 1, 2, 3};
    do {
        memcpy(perm[np++], p, sizeof p);
    } while (std::next_permutation(p, p + 4));
    std::vector<long> hits(24, 0), hits_c_last(24, 0);
    long hits_tree = 0, hits_exact = 0, total = 0;
    for (int trial = 0; trial < 200; trial++) {
        for (auto &v : a)
            v = std::ldexp(mant(rng), ex(rng)) * ((rng() & 1) ? 1 : -1);
        for (auto &v : c)
            v = std::ldexp(mant(rng), ex(rng)) * ((rng() & 1) ? 1 : -1);
        run(a, b, c, d);
        for (int o = 0; o < n_out; o++) {
            double t[4];
            for (int k = 0; k < 4; k++)
                t[k] = a[src[o][k]];
            total++;
            for (int q = 0; q < 24; q++) {
                if (chain(t, perm[q], 4, c[o]) == d[o])
                    hits[q]++;
                // the accumulator added last instead of first
                double s = t[perm[q][0]];
                for (int k = 1; k < 4; k++)
                    s = s + t[perm[q][k]];
                if (s + c[o] == d[o])
                    hits_c_last[q]++;
            }
            if (((t[0] + t[1]) + (t[2] + t[3])) + c[o] == d[o])
                hits_tree++;
            // one rounding at the end (exact sum): long double is enough for 5 terms over 60 binades? no - use a
            // compensated sum as an approximation of "correctly rounded"
            long double e = (long double)c[o];
            for (int k = 0; k < 4; k++)
                e += (long double)t[k];
            if ((double)e == d[o])
                hits_exact++;
        }
    }
    int best = 0;
    for (int q = 1; q < 24; q++)
        if (hits[q] > hits[best])
            best = q;
    printf("%s: %ld outputs checked. Sequential fma chain from the accumulator, best lane order (by position in the sorted source-lane list) [%d %d %d %d]: %ld match (%.4f %%)\n",
           name, total, perm[best][0], perm[best][1], perm[best][2], perm[best][3], hits[best], 100.0 * hits[best] / total);
    int bl = 0;
    for (int q = 1; q < 24; q++)
        if (hits_c_last[q] > hits_c_last[bl])
            bl = q;
    printf("%s: alternatives - accumulator added last: best %.4f %% ; pairwise tree: %.4f %% ; one rounding (80-bit sum): %.4f %%\n", name,
           100.0 * hits_c_last[bl] / total, 100.0 * hits_tree / total, 100.0 * hits_exact / total);
}

int main()
{
    double *da, *db, *dc, *dd;
    hipMalloc(&da, 64 * 8);
    hipMalloc(&db, 64 * 8);
    hipMalloc(&dc, 256 * 8);
    hipMalloc(&dd, 256 * 8);
    auto run4 = [&](std::vector<double> &a, std::vector<double> &b, std::vector<double> &c, std::vector<double> &d) {
        hipMemcpy(da, a.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 64 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_4x4, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 64 * 8, hipMemcpyDeviceToHost);
    };
    auto run16 = [&](std::vector<double> &a, std::vector<double> &b, std::vector<double> &c, std::vector<double> &d) {
        hipMemcpy(da, a.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 256 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_16x16, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 256 * 8, hipMemcpyDeviceToHost);
    };
    probe("v_mfma_f64_4x4x4_4b", 64, run4);
    probe("v_mfma_f64_16x16x4", 256, run16);

    // dependent-chain latency, one wave per CU and four waves per CU (one per SIMD)
    const int n = 32000;
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, 1024 * 64 * 8);
    hipMalloc(&cyc, 1024 * 8);
    std::vector<double> ones(64, 1.0);
    hipMemcpy(da, ones.data(), 64 * 8, hipMemcpyHostToDevice);
    std::vector<unsigned long long> h(1024);
    for (int blocks : {1, 256, 1024}) {
        for (int which = 0; which < 3; which++) {
            if (which == 0)
                hipLaunchKernelGGL(k_chain_valu, dim3(blocks), dim3(64), 0, 0, da, out, n, cyc);
            else if (which == 1)
                hipLaunchKernelGGL(k_chain_4x4, dim3(blocks), dim3(64), 0, 0, da, out, n, cyc);
            else
                hipLaunchKernelGGL(k_chain_16x16, dim3(blocks), dim3(64), 0, 0, da, out, n, cyc);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + blocks);
            const char *nm[3] = {"v_add_f64 (1 term per instruction)", "v_mfma_f64_4x4x4_4b (4 terms, 16 chains)", "v_mfma_f64_16x16x4 (4 terms, 256 chains)"};
            printf("dependent chain, %4d one-wave workgroups: %-42s %.2f clocks per instruction (median over waves)\n", blocks, nm[which],
                   (double)h[blocks / 2] / n);
        }
    }
    return 0;
}
