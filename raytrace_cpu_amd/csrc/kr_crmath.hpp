// kr_crmath.hpp -- atan2, asin, acos and tan, correctly rounded in practice, for the device ImagePlane constructor (imageplane.cpp:50-107 calls
// acos twice, atan2, asin and tan per ray: with the device library's routines (<= 1-2 ulp) only 75-79 % of the device-built rays carried the
// reference's phi and 93-96 % its theta and Q; the host C library the reference calls is, for these functions, correctly rounded on all but ~1e-4 of
// arguments).  Everything is evaluated in double-double -- unevaluated sums hi + lo of two doubles, error-free transformations (two_sum, two_prod
// through the fused multiply-add) -- to ~2^-68 and rounded ONCE, so a result differs from the correctly rounded one only when the exact value lies
// within 2^-15 ulp of a rounding boundary.  -ffp-contract=off is assumed (every fused operation is spelled out).
// Compiles as device code (hipcc) and as host C++ (g++), the latter so that tests/crmath_check.cpp can compare with libquadmath on the CPU.
//
//   atan(t), |t| <= 1:  k = rint(32 t), c = k / 32;  u = (t - c) / (1 + t c)  (|u| <= 1 / 64);  atan t = atan c (table, 33 pairs) + atan u,
//                       atan u = u - u^3/3 + u^5/5 - ... - u^15/15  (next term u^16 / 17 <= 5e-31 relative), leading terms in double-double;
//   atan2(y, x):        octant reduction to t = min / max in double-double, pi/2 and pi as pairs;
//   asin x = atan2(x, sqrt(1 - x^2)),  acos x = atan2(sqrt(1 - x^2), x)  with 1 - x^2 and its root in double-double;
//   tan x = sin x / cos x  from kr_sincos.hpp's reduced-argument kernel before ITS final rounding.
#pragma once
#include <math.h>

#include "kr_sincos.hpp"

namespace krcr {

struct dd { double hi, lo; };

KR_SC_FN dd two_sum(double a, double b) { const double s = a + b, bb = s - a; return dd{s, (a - (s - bb)) + (b - bb)}; }
KR_SC_FN dd fast_two_sum(double a, double b) { const double s = a + b; return dd{s, b - (s - a)}; }       // |a| >= |b|
KR_SC_FN dd two_prod(double a, double b) { const double p = a * b; return dd{p, __builtin_fma(a, b, -p)}; }
KR_SC_FN dd dd_add(dd a, dd b) { dd s = two_sum(a.hi, b.hi); const dd t = two_sum(a.lo, b.lo); s.lo += t.hi; s = fast_two_sum(s.hi, s.lo); s.lo += t.lo; return fast_two_sum(s.hi, s.lo); }
KR_SC_FN dd dd_add_d(dd a, double b) { dd s = two_sum(a.hi, b); s.lo += a.lo; return fast_two_sum(s.hi, s.lo); }
KR_SC_FN dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
KR_SC_FN dd dd_mul(dd a, dd b) { dd p = two_prod(a.hi, b.hi); p.lo += a.hi * b.lo + a.lo * b.hi; return fast_two_sum(p.hi, p.lo); }
KR_SC_FN dd dd_mul_d(dd a, double b) { dd p = two_prod(a.hi, b); p.lo += a.lo * b; return fast_two_sum(p.hi, p.lo); }
// a / b to ~2^-104: two quotient digits and a third for the rounding
KR_SC_FN dd dd_div(dd a, dd b)
{
    const double q1 = a.hi / b.hi;
    dd r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    dd q = fast_two_sum(q1, q2);
    return dd_add_d(q, q3);
}
KR_SC_FN dd dd_sqrt(dd a)        // a.hi > 0
{
    const double s = __builtin_sqrt(a.hi);
    const dd sq = two_prod(s, s);
    const double e = ((a.hi - sq.hi) - sq.lo) + a.lo;     // a - s^2: the leading parts cancel exactly
    return fast_two_sum(s, e / (2.0 * s));
}

#if defined(__HIPCC__)
#define KR_CR_TABLE static __device__ const
#else
#define KR_CR_TABLE static const
#endif
KR_CR_TABLE double kAtanTable[33][2] = {
    {0x0p+0, 0x0p+0},
    {0x1.ffd55bba97625p-6, -0x1.5ec431444912cp-60},
    {0x1.ff55bb72cfdeap-5, -0x1.c934d86d23f1dp-60},
    {0x1.7ee182602f10fp-4, -0x1.cfb654c0c3d98p-58},
    {0x1.fd5ba9aac2f6ep-4, -0x1.cd37686760c17p-59},
    {0x1.3d6eee8c6626cp-3, 0x1.61a3b0ce9281bp-57},
    {0x1.7b97b4bce5b02p-3, 0x1.347b0b4f881cap-58},
    {0x1.b90d7529260a2p-3, 0x1.17b10d2e0e5aap-61},
    {0x1.f5b75f92c80ddp-3, 0x1.8ab6e3cf7afbdp-57},
    {0x1.18bf5a30bf178p-2, 0x1.30ca4748b1bf8p-57},
    {0x1.362773707ebccp-2, -0x1.963a544b672d8p-57},
    {0x1.530ad9951cd4ap-2, -0x1.2566480884082p-57},
    {0x1.6f61941e4def1p-2, -0x1.c63aae6f6e918p-56},
    {0x1.8b24d394a1b25p-2, 0x1.b6d0ba3748fa8p-56},
    {0x1.a64eec3cc23fdp-2, -0x1.24dec1b50b7ffp-56},
    {0x1.c0db4c94ec9fp-2, -0x1.cc1ce70934c34p-56},
    {0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56},
    {0x1.f40dd0b541418p-2, -0x1.a3992dc382a23p-57},
    {0x1.0657e94db30dp-1, -0x1.d5b495f6349e6p-56},
    {0x1.1255d9bfbd2a9p-1, -0x1.2bdaee1c0ee35p-58},
    {0x1.1e00babdefeb4p-1, -0x1.928df287a668fp-58},
    {0x1.2958e59308e31p-1, -0x1.09e73b0c6c087p-56},
    {0x1.345f01cce37bbp-1, 0x1.1021137c71102p-55},
    {0x1.3f13fb89e96f4p-1, 0x1.ecf8b492644fp-56},
    {0x1.4978fa3269ee1p-1, 0x1.2419a87f2a458p-56},
    {0x1.538f57b89061fp-1, -0x1.1bb74abda520cp-55},
    {0x1.5d58987169b18p-1, 0x1.0028e4bc5e7cap-57},
    {0x1.66d663923e087p-1, -0x1.6ea6febe8bbbap-56},
    {0x1.700a7c5784634p-1, -0x1.8c34d25aadef6p-56},
    {0x1.78f6bbd5d315ep-1, 0x1.406a08980374p-55},
    {0x1.819d0b7158a4dp-1, -0x1.bf76229d3b917p-56},
    {0x1.89ff5ff57f1f8p-1, -0x1.55b9a5e177a1bp-55},
    {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55},
};
KR_SC_FN dd pi_pair() { return dd{0x1.921fb54442d18p+1, 0x1.1a62633145c07p-53}; }
KR_SC_FN dd half_pi_pair() { return dd{0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54}; }

// atan t for 0 <= t <= 1 (t a double-double)
KR_SC_FN dd atan_unit(dd t)
{
    const double kf = __builtin_rint(32.0 * t.hi);
    const int k = (int) kf;
    const double c = kf * 0.03125;                                   // exact
    // u = (t - c) / (1 + t c)
    const dd num = dd_add_d(t, -c);
    const dd den = dd_add_d(dd_mul_d(t, c), 1.0);
    const dd u = dd_div(num, den);
    // atan u = u - u^3 (1/3 - u^2 (1/5 - u^2 (1/7 - ...)));  |u| <= 1/64: the bracket after 1/3 needs 2^-56 relative, i.e. 1/5 .. in plain double
    const dd u2 = dd_mul(u, u);
    const double z = u2.hi;
    double p = 1.0 / 15.0;
    p = __builtin_fma(-z, p, 1.0 / 13.0);
    p = __builtin_fma(-z, p, 1.0 / 11.0);
    p = __builtin_fma(-z, p, 1.0 / 9.0);
    p = __builtin_fma(-z, p, 1.0 / 7.0);
    p = __builtin_fma(-z, p, 1.0 / 5.0);
    // 1/3 - u^2 p in double-double (1/3 as a pair), times u^3
    const dd third = {0x1.5555555555555p-2, 0x1.5555555555555p-56};
    const dd br = dd_add(third, dd_neg(dd_mul_d(u2, p)));
    const dd u3 = dd_mul(u2, u);
    const dd series = dd_add(u, dd_neg(dd_mul(u3, br)));
    return dd_add(dd{kAtanTable[k][0], kAtanTable[k][1]}, series);
}

// atan2(y, x) for finite, non-zero double-double arguments; returns the unrounded pair
KR_SC_FN dd atan2_dd(dd y, dd x)
{
    const bool yneg = y.hi < 0, xneg = x.hi < 0;
    const dd ay = yneg ? dd_neg(y) : y, ax = xneg ? dd_neg(x) : x;
    const bool steep = ay.hi > ax.hi || (ay.hi == ax.hi && ay.lo > ax.lo);
    const dd t = steep ? dd_div(ax, ay) : dd_div(ay, ax);              // in [0, 1]
    dd a = atan_unit(t);
    if (steep) a = dd_add(half_pi_pair(), dd_neg(a));                            // atan(ay / ax) = pi/2 - atan(ax / ay)
    if (xneg) a = dd_add(pi_pair(), dd_neg(a));
    return yneg ? dd_neg(a) : a;
}

KR_SC_FN double kr_atan2_cr(double y, double x)
{
    // zeros, infinities and NaNs follow the C library's case table; the constructor never produces them (x = 0 gives atan2(+-0, D sin i - y cos i > 0))
    if (!(__builtin_fabs(y) < __builtin_inf()) || !(__builtin_fabs(x) < __builtin_inf()) || y == 0.0 || x == 0.0) return ::atan2(y, x);
    const dd a = atan2_dd(dd{y, 0.0}, dd{x, 0.0});
    return a.hi + a.lo;
}

// sqrt(1 - x^2) as a pair, |x| < 1
KR_SC_FN dd sqrt_one_minus_sq(double x)
{
    const dd x2 = two_prod(x, x);
    const dd om = dd_add_d(dd_neg(x2), 1.0);
    return dd_sqrt(om);
}

KR_SC_FN double kr_asin_cr(double x)
{
    if (!(__builtin_fabs(x) < 1.0) || x == 0.0) return ::asin(x);            // +-1, out of range, NaN, +-0: the library's case table
    const dd a = atan2_dd(dd{x, 0.0}, sqrt_one_minus_sq(x));
    return a.hi + a.lo;
}

KR_SC_FN double kr_acos_cr(double x)
{
    if (!(__builtin_fabs(x) < 1.0) || x == 0.0) return ::acos(x);
    const dd a = atan2_dd(sqrt_one_minus_sq(x), dd{x, 0.0});
    return a.hi + a.lo;
}

// tan x = sin x / cos x from the reduced-argument kernel of kr_sincos.hpp (pairs accurate to ~2^-66), |x| < 1024; elsewhere the library
KR_SC_FN double kr_tan_cr(double x)
{
    if (!(__builtin_fabs(x) < 1024.0) || x == 0.0) return ::tan(x);
    const double t = __builtin_rint(x * 6.36619772367581382433e-01);        // 2/pi
    const int n = (int) t;
    const double r0 = __builtin_fma(-t, 1.57079632679489655800e+00, x);     // (the reduction of kr_sincos_general_f64)
    const double r = __builtin_fma(-t, 6.12323399573676603587e-17, r0);
    const double y = __builtin_fma(-t, 6.12323399573676603587e-17, r0 - r);
    double sh, sl, ch, cl;
    kr_sincos_cr_core_pairs(r, y - t * -1.4973849048591698e-33, sh, sl, ch, cl);
    const dd sn = fast_two_sum(sh, sl), cs = fast_two_sum(ch, cl);
    const dd q = (n & 1) ? dd_neg(dd_div(cs, sn)) : dd_div(sn, cs);          // odd quadrant: tan(r + pi/2) = -cos r / sin r
    return q.hi + q.lo;
}

}  // namespace krcr
