// kr_replay.hpp -- n identical floating-point additions in closed form.
//
//     for (i = 0; i < n; i++) x = x + dx;        (each sum rounded to nearest-even, as the loop would)
//
// is what the RK45 fixed-point replay needs for t and phi of a captured ray (kr_device.hpp::step_rk45: up to ~1e5
// additions per ray, during which the other 63 lanes of the wave wait).  Inside one binade every double is a multiple of
// the binade's ulp u, so as long as the exact sum x + dx stays inside the binade the rounded sum is x + q u with the SAME
// integer q = rint(dx / u) every time -- unless dx / u sits exactly on a half (a tie, resolved by the parity of x / u and
// therefore alternating; left to the literal loop).  The additions are thus replayed binade by binade: one integer
// multiply-add per binade, literal additions only for the steps that cross a binade boundary, zero, or a subnormal.
// q = 0 means dx is absorbed: x has stopped moving and every later addition is the same no-op.
// Bit-identical to the loop for every finite input (tests/test_replay_additions.py: 3e6 random cases in every regime
// against the literal loop, compiled for the host; on the device the same function is checked through kr_debug_arith_f64).
// Compiles as device code (hipcc) and as plain host C++ (g++).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define KR_RP_FN __host__ __device__ inline
#else
#define KR_RP_FN static inline
#endif

KR_RP_FN uint64_t kr_rp_bits(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
KR_RP_FN double kr_rp_from_bits(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }

// x after n additions of dx.  `literal_out` (optional) receives the number of additions that were carried out one by one.
KR_RP_FN double kr_replay_additions(double x, double dx, long long n, long long* literal_out = nullptr)
{
    long long literal = 0;
    if (dx == 0 && n > 0) { x = x + dx; n = 0; literal = 1; }                    // (x + 0 = x; one addition settles -0 + +0)
    while (n > 0) {
        const uint64_t bx = kr_rp_bits(x);
        const int ex = (int) ((bx >> 52) & 0x7FF);
        bool closed = false;
        if (ex >= 54 && ex <= 2045 && dx == dx && dx != 0) {                      // x normal, well away from the subnormals, finite; dx a number
            const double u = kr_rp_from_bits((uint64_t) (ex - 52) << 52);         // ulp of x's binade (a normal power of two)
            const double m = dx / u;                                              // exact (division by a power of two), may be huge
            if (fabs(m) < 4611686018427387904.0) {                                // |m| < 2^62
                const double fl = floor(m);
                const double fr = m - fl;                                         // exact: m has at most 53 significant bits
                if (fr != 0.5) {
                    const long long q = (long long) (fr > 0.5 ? fl + 1 : fl);    // rint(dx / u), no tie
                    // X = |x| / u in [2^52, 2^53); moving by Q = +-q per step in |x|; the exact sum must stay in [2^52 u, 2^53 u]
                    const long long X = (long long) ((bx & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull);
                    const bool neg = (bx >> 63) != 0;
                    const long long Q = neg ? -q : q;                             // step of |x| in units of u
                    const bool towards_zero = neg ? (m > 0) : (m < 0);
                    if (q == 0) {
                        // dx is below half an ulp: absorbed for good -- unless x is the power of two itself and dx points below it,
                        // where the grid is twice as fine (that step is taken literally)
                        if (!(X == 4503599627370496ll && towards_zero)) { n = 0; break; }
                    } else {
                    // the exact (unrounded) sum of a step is (X + M) u with M = +-m: closed form needs lo <= X_j + M <= hi at every step,
                    // which holds if it holds for the rounded position with a margin of one unit
                    const long long lo = 4503599627370496ll, hi = 9007199254740992ll;   // 2^52, 2^53
                    long long steps;
                    if (Q > 0) steps = (hi - 1 - X) / Q - 1;                      // X + (steps) Q <= hi - 1 - Q  -> every exact sum < hi
                    else steps = (X - lo - 1) / (-Q) - 1;                         // X + (steps) Q >= lo + 1 + |Q| -> every exact sum > lo
                    if (steps > n) steps = n;
                    if (steps >= 1) {
                        const long long Xn = X + steps * Q;                       // |steps Q| < 2^53: no overflow
                        x = kr_rp_from_bits((bx & 0xFFF0000000000000ull) | ((uint64_t) Xn & 0x000FFFFFFFFFFFFFull));
                        n -= steps;
                        closed = true;
                    }
                    }
                }
            }
        }
        if (!closed) {
            const double y = x + dx;                                              // a boundary step, a tie, zero, a subnormal, inf / NaN: literally
            ++literal;
            --n;
            if (y == x && !(x == 0)) { n = 0; }                                   // a fixed point of the addition (absorbed, or inf)
            x = y;
            if (y != y) { n = 0; }                                                // NaN stays NaN
        }
    }
    if (literal_out) *literal_out = literal;
    return x;
}
