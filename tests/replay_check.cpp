// TEST INFRASTRUCTURE: kr_replay_additions (csrc/kr_replay.hpp, compiled for the host) against the literal loop it replaces.
// prints: cases  mismatches  literal_additions_total  loop_additions_total
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../raytrace_cpu_amd/csrc/kr_replay.hpp"

static unsigned long long st = 88172645463325252ull;
static unsigned long long rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static double uni() { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }

static double loop(double x, double dx, long long n) { for (long long i = 0; i < n; i++) x = x + dx; return x; }

int main(int argc, char** argv)
{
    const long cases = argc > 1 ? atol(argv[1]) : 200000;
    long bad = 0;
    long long lit = 0, tot = 0;
    for (long c = 0; c < cases; c++) {
        const int regime = (int) (rnd() % 8);
        double x = ldexp(1.0 + uni(), (int) (rnd() % 600) - 300);
        if (rnd() & 1) x = -x;
        double dx;
        long long n = 1 + (long long) (rnd() % 100000);
        int ex;
        frexp(x, &ex);
        const double u = ldexp(1.0, ex - 53);                 // ulp of x's binade
        switch (regime) {
            case 0: dx = x * ldexp(uni() + 0.5, -(int) (rnd() % 70)); break;                 // same sign, any size down to absorbed
            case 1: dx = -x * ldexp(uni() + 0.5, -(int) (rnd() % 70)); break;                // opposite sign: |x| shrinks, may cross zero
            case 2: dx = ((double) (rnd() % 64) + 0.5) * u * ((rnd() & 1) ? 1 : -1); break;  // exact ties
            case 3: dx = ((double) (rnd() % 1024) + uni()) * u * ((rnd() & 1) ? 1 : -1); break;   // a few ulps per step: many binade crossings
            case 4: dx = ldexp(uni(), -1060) * ((rnd() & 1) ? 1 : -1); x = ldexp(uni(), -1040) * ((rnd() & 1) ? 1 : -1); n = 1 + rnd() % 5000; break;   // subnormals
            case 5: x = 0.0; dx = ldexp(uni() - 0.5, (int) (rnd() % 40) - 20); n = 1 + rnd() % 5000; break;             // from zero
            case 6: x = ldexp(1.0, (int) (rnd() % 200) - 100) * (1 - ldexp(1.0, -53) * (rnd() % 4)); dx = u * (uni() * 8 - 4); break;   // just below a power of two
            default: dx = uni() < 0.05 ? 0.0 : x * ldexp(uni(), -40); n = 1 + rnd() % 300000; break;                       // the replay's own regime: tiny increments, long runs
        }
        long long l = 0;
        const double got = kr_replay_additions(x, dx, n, &l);
        const double want = loop(x, dx, n);
        lit += l; tot += n;
        if (memcmp(&got, &want, 8) != 0 && !(got != got && want != want)) {
            if (bad < 5) printf("MISMATCH regime %d x %.17g dx %.17g n %lld: got %.17g want %.17g\n", regime, x, dx, n, got, want);
            ++bad;
        }
    }
    printf("%ld %ld %lld %lld\n", cases, bad, lit, tot);
    return bad ? 1 : 0;
}
