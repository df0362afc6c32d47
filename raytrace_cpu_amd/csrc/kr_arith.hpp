// kr_arith.hpp -- scalar fp64 / fp32 building blocks of the device code: the strict path's correctly rounded quotient and square root (lean
// chains), the libm front (sin / cos: kr_sincos.hpp), the DOPRI5 controller's fifth root, min / max with the reference's operand semantics.
// Pure __device__ functions of registers.  Included by kr_device.hpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kr_sincos.hpp"

namespace kr {

#define KR_DEV __device__ __forceinline__

// ---- scalar helpers -------------------------------------------------------------------------
KR_DEV double kr_abs(double x) { return __builtin_fabs(x); }
KR_DEV float kr_abs(float x) { return __builtin_fabsf(x); }

// Correctly rounded fp64 quotient and square root WITHOUT the range-scaling links of the compiler's sequences.
// The compiler lowers a/b to: v_div_scale x2 -> v_rcp_f64 -> two Newton steps -> q = a*y -> residual -> v_div_fmas ->
// v_div_fixup, an 11-deep dependent chain at ~32 cycles of fp64 latency per link, 20+ times per RK4 step; sqrt is a
// 14-deep chain.  The scale / fmas / fixup links (ldexp / class tests for sqrt) are the identity unless an operand is
// zero, infinite, NaN, denormal or within ~2^100 of the ends of the exponent range; the remaining links ARE the
// compiler's computation, so for every operand pair a healthy ray produces the result is bit-identical to IEEE
// (tests/test_gpu_primitives.py: 2e6 random pairs + edge cases against the compiler's a/b and numpy, on the GPU).
// What the lean chains do NOT reproduce: a zero denominator gives NaN (IEEE: +-inf or NaN), an infinite one NaN (IEEE:
// +-0), a -0 numerator +0.  In the tracer a denominator is exactly zero only on the polar axis (sin theta = 0) or on
// Delta = 0, where the reference's own evaluation is already inf/NaN-poisoned, or for phidot/thetadot = 0 in the step
// heuristic, where the quotient only feeds `step > q` comparisons that are false for +inf and NaN alike
// (tests/test_gpu_parity.py::test_degenerate_denominators_match_oracle).  Guarding instead of accepting that was
// measured and rejected: a range test per quotient 231 ms, an out-of-line IEEE re-run per evaluation 227 ms, the
// compiler's division 188 ms, unguarded lean chains 165 ms (PointSource 1e7 rays, RK4): every guard splits the
// scheduling region the independent chains overlap in.
// The reciprocal both forms below start from: v_rcp_f64 (~2^-23) and ONE cubic step, y0 (1 + e + e^2) = (1 / b)(1 - e^3) with e = 1 - b y0:
// 2^-69 before its rounding, i.e. as close to RN(1 / b) as the two Newton steps of the compiler's sequence get (one fused operation
// more), which is all the final correction q + (a - b q) y needs to land on the correctly rounded quotient (Markstein).
KR_DEV double lean_recip(double b)
{
    const double y0 = __builtin_amdgcn_rcp(b);
    const double e = __builtin_fma(-b, y0, 1.0);
    return __builtin_fma(y0, __builtin_fma(e, e, e), y0);
}
// 1 / b from an estimate y0 = (1 / b)(1 + O(1e-15)) put together from reciprocals already at hand (1 / (x y) from 1 / x and 1 / y, 1 / x from
// 1 / (x y) and y): ONE Newton step, (1 / b)(1 - e^2) with |e| < 2^-49 -- closer to RN(1 / b) than lean_recip's own 2^-69 -- for three issue slots
// instead of a quarter-rate v_rcp_f64 and three more.  The quotients formed with it are the correctly rounded ones, as with lean_recip.
KR_DEV double lean_recip_from(double b, double y0)
{
    return __builtin_fma(y0, __builtin_fma(-b, y0, 1.0), y0);
}
// a / b given y = lean_recip(b): several quotients over one denominator share the reciprocal (same bits as lean_div(a, b) each)
KR_DEV double lean_div_y(double a, double b, double y)
{
    const double q = a * y;
    return __builtin_fma(__builtin_fma(-b, q, a), y, q);
}
KR_DEV double lean_div(double a, double b) { return lean_div_y(a, b, lean_recip(b)); }

// a / b for a divisor b that is uniform over the launch, given inv_b = RN(1/b) computed on the host in IEEE arithmetic.
// q0 = RN(a inv_b) is within an ulp of a/b, r = a - b q0 is exact in the FMA, and RN(q0 + r inv_b) is then the correctly
// rounded quotient (Markstein's theorem; it needs inv_b correctly rounded, which excludes nothing for finite normal b
// whose significand is not all ones -- checked on the host, kr_trace.hip::make_consts).  Same bits as lean_div / IEEE,
// 3 instructions instead of 8.  ok = false (degenerate divisor) falls back to lean_div.
KR_DEV double div_by_uniform(double a, double b, double inv_b, bool ok)
{
    if (!ok) return lean_div(a, b);           // wave-uniform
    const double q0 = a * inv_b;
    const double r = __builtin_fma(-b, q0, a);
    return __builtin_fma(r, inv_b, q0);
}

KR_DEV double lean_sqrt(double x)      // x >= 0; +0 -> +0
{
    // rsq(0) = inf would turn the chain into NaN; capped at 1e300 (a no-op for every x > 0, whose rsq is < 1e154) the chain
    // returns +0 for +0 by itself: one v_min instead of a compare and two selects on the result
    const double y = __builtin_fmin(__builtin_amdgcn_rsq(x), 1e300);
    const double g0 = x * y;
    const double h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0);
    const double h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    return __builtin_fma(d1, h1, g2);
}

// arithmetic of the strict path: the lean chains in double precision, the compiler's (IEEE) sequences in single
KR_DEV double dv(double a, double b) { return lean_div(a, b); }
KR_DEV float dv(float a, float b) { return a / b; }
KR_DEV double sq(double x) { return lean_sqrt(x); }
KR_DEV float sq(float x) { return __builtin_sqrtf(x); }
// 1 / b for several quotients over b (double: the refined reciprocal; float: unused) and the quotient that goes with it
KR_DEV double dv_recip(double b) { return lean_recip(b); }
KR_DEV float dv_recip(float) { return 0.0f; }
KR_DEV double dv_y(double a, double b, double y) { return lean_div_y(a, b, y); }
KR_DEV float dv_y(float a, float b, float) { return a / b; }
KR_DEV double div_const(double a, double b, double inv_b, bool ok) { return div_by_uniform(a, b, inv_b, ok); }
KR_DEV float div_const(float a, float b, float, bool) { return a / b; }

KR_DEV bool kr_finite(double x) { return __builtin_fabs(x) < __builtin_inf(); }
KR_DEV bool kr_finite(float x) { return __builtin_fabsf(x) < __builtin_inff(); }

KR_DEV double kr_sqrt(double x) { return __builtin_sqrt(x); }
KR_DEV float kr_sqrt(float x) { return __builtin_sqrtf(x); }
template <bool LONE = false> KR_DEV void kr_sincos(double x, double& s, double& c) { kr_sincos_f64<LONE>(x, s, c); }      // (LONE: kr_sincos.hpp)
// float: evaluated in double and rounded once -- correctly rounded in all but ~1e-8 of the arguments, which is what glibc's sinf / cosf / powf / tanf
// (the float instantiation's libm, <= 0.56 ulp) are in all but a few per cent: the float kernels then differ from the reference's float build only
// where glibc's own float routines are not correctly rounded (the device library's float routines: <= 1-2 ulp).
template <bool LONE = false> KR_DEV void kr_sincos(float x, float& s, float& c)
{
    double sd, cd;
    kr_sincos_fast_f64((double) x, sd, cd);
    s = (float) sd;
    c = (float) cd;
}
// double: the strict path's own correctly rounded pair (kr_sincos.hpp) instead of the device libm (<= 1 ulp): the O(N) passes, the ray sources
// and the FlatPlane stop test then differ from glibc only where glibc is not correctly rounded; the unused half is dead code
KR_DEV double kr_sin(double x) { double s, c; kr_sincos_f64(x, s, c); return s; }
KR_DEV float kr_sin(float x) { float s, c; kr_sincos(x, s, c); return s; }
KR_DEV double kr_cos(double x) { double s, c; kr_sincos_f64(x, s, c); return c; }
KR_DEV float kr_cos(float x) { float s, c; kr_sincos(x, s, c); return c; }
KR_DEV double kr_pow(double x, double y) { return ::pow(x, y); }
KR_DEV float kr_pow(float x, float y) { return (float) ::pow((double) x, (double) y); }

// x^(1/5) for the DOPRI5 step controller (raytracer.cpp:1517: pow(1/max(err, 1e-10), 0.2), then 0.9 x that clamped to
// [0.1, 5]).  The clamp makes the root matter only for x in [1.7e-5, 5.3e3]; x is first brought into [1e-6, 1e6], which
// cannot change the clamped factor, so that a single-precision seed is always in range.  Seed from the hardware log2 / exp2
// (~1e-7), two Newton steps y <- y (4 + x / y^5) / 5 (error 2 e^2 each) a residual correction and the factor that
// turns the exact root into x^0.2 with the double constant 0.2, correctly rounded (tests/test_gpu_primitives.py), ~45 instructions against ~200 for the library pow.  NaN in, NaN out.
KR_DEV double fifth_root_for_controller(double x)
{
    if (!(x == x)) return x;
    x = __builtin_fmin(__builtin_fmax(x, 1e-6), 1e6);
    const float lg = __builtin_amdgcn_logf((float) x);                     // log2 x
    double y = (double) __builtin_amdgcn_exp2f(0.2f * lg);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y, y4 = y2 * y2, y5 = y4 * y;
        y = y * __builtin_fma(x, lean_div(1.0, y5), 4.0) * 0.2;
    }
    // Last step in double-double: y^5 as an exact product chain, the residual x - y^5 exactly (the leading parts cancel), then ONE rounding of
    // y + (correction + exponent term).  The reference raises to the DOUBLE 0.2 = 1/5 + 1.11e-17, not to 1/5: x^0.2 = x^(1/5) (1 + 1.11e-17 ln x).
    // Correctly rounded on 3e6 random arguments (against __float128 powq); glibc's pow agrees with that on 99.9 %.
    const double y2h = y * y, y2l = __builtin_fma(y, y, -y2h);
    const double y4h = y2h * y2h, y4l = __builtin_fma(y2h, y2h, -y4h) + 2.0 * (y2h * y2l);
    const double y5h = y4h * y, y5l = __builtin_fma(y4h, y, -y5h) + y4l * y;
    const double r = (x - y5h) - y5l;
    const double corr = r * lean_div(0.2, y4h);
    return y + __builtin_fma(y, 7.695479593116622e-18 * (double) lg, corr);       // 1.1102230246251565e-17 * ln 2 * log2 x
}
KR_DEV float fifth_root_for_controller(float x) { return (float) ::pow((double) x, (double) 0.2f); }
KR_DEV double kr_log(double x) { return ::log(x); }
// std::max / std::min operand semantics (they decide what a NaN operand does; raytracer.cpp:1512-1533)
template <typename T> KR_DEV T std_max(T a, T b) { return (a < b) ? b : a; }
template <typename T> KR_DEV T std_min(T a, T b) { return (b < a) ? b : a; }

// y + 2 x, rounded once: the product is exact, so this IS the two-operation sum the reference writes (bit for bit), in one instruction
KR_DEV double kr_fma2(double x, double y) { return __builtin_fma(2.0, x, y); }
KR_DEV float kr_fma2(float x, float y) { return __builtin_fmaf(2.0f, x, y); }

template <typename T> struct Lim;
template <> struct Lim<double> { static KR_DEV double max() { return 1.7976931348623157e308; } };
template <> struct Lim<float> { static KR_DEV float max() { return 3.402823466e38f; } };

constexpr double kPi = 3.14159265358979323846;
constexpr double kPi2 = 1.57079632679489661923;

}  // namespace kr
