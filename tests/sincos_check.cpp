// TEST INFRASTRUCTURE: measures the error of csrc/kr_sincos.hpp (compiled for the host) against long-double libm.
// prints: n  max_ulp_sin  max_ulp_cos  max_ulp_vs_glibc_sin  max_ulp_vs_glibc_cos  frac_sin_equal_glibc  frac_cos_equal_glibc
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../raytrace_cpu_amd/csrc/kr_sincos.hpp"

static double ulp_of(double x) { double a = std::fabs(x); return std::nextafter(a, INFINITY) - a; }

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    const double lo = argc > 2 ? atof(argv[2]) : -1.0, hi = argc > 3 ? atof(argv[3]) : 4.2;
    const bool fast = argc > 4 && !strcmp(argv[4], "fast");      // the fast-path routine instead of the strict-path one
    double ms = 0, mc = 0, gs = 0, gc = 0;
    long es = 0, ec = 0;
    unsigned long long st = 88172645463325252ull;
    for (long i = 0; i < n; i++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double x = lo + (hi - lo) * ((st >> 11) * (1.0 / 9007199254740992.0));
        if (i < 64) x = (i % 8) * 0.78539816339744830962 + (i / 8 - 4) * 1e-9 * ((i & 1) ? 1 : -1);   // around multiples of pi/4
        double s, c;
        if (fast) kr_sincos_fast_f64(x, s, c);
        else kr_sincos_f64(x, s, c);
        const long double ls = sinl((long double) x), lc = cosl((long double) x);
        const double us = (double) (fabsl((long double) s - ls) / ulp_of((double) ls)), uc = (double) (fabsl((long double) c - lc) / ulp_of((double) lc));
        if (us > ms) ms = us;
        if (uc > mc) mc = uc;
        const double g1 = std::sin(x), g2 = std::cos(x);
        const double d1 = std::fabs(s - g1) / ulp_of(g1), d2 = std::fabs(c - g2) / ulp_of(g2);
        if (d1 > gs) gs = d1;
        if (d2 > gc) gc = d2;
        es += (s == g1); ec += (c == g2);
    }
    printf("%ld %.4f %.4f %.1f %.1f %.5f %.5f\n", n, ms, mc, gs, gc, (double) es / n, (double) ec / n);
    return 0;
}
