// kr_device.hpp -- device-side Kerr null-geodesic arithmetic for gfx950 (MI355X).
//
// One ray per work-item; the whole ray state lives in VGPRs.  Everything here is a pure
// __device__ function of registers: no memory traffic, no LDS, no cross-lane ops.  The per-step
// semantics follow the reference propagators (file:line cited per function, paths relative to the
// reference tree); arithmetic is written with the reference's association so that, built with
// -ffp-contract=off, a step differs from the CPU result only through the device libm (sin, cos,
// pow: <= 1-2 ulp vs glibc), never through re-ordering.
//
// T = double is the product precision; T = float mirrors the reference's second instantiation
// (raytracer.cpp:1896-1897).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kr_trace.h"
#include "kr_sincos.hpp"
#include "kr_replay.hpp"

namespace kr {

#define KR_DEV __device__ __forceinline__

// ---- scalar helpers -------------------------------------------------------------------------
KR_DEV double kr_abs(double x) { return __builtin_fabs(x); }
KR_DEV float kr_abs(float x) { return __builtin_fabsf(x); }
#ifndef KR_LEAN_IEEE
#define KR_LEAN_IEEE 1
#endif

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
// scheduling region the independent chains overlap in.  -DKR_LEAN_IEEE=0 builds the compiler's sequences throughout.
// The reciprocal both forms below start from: v_rcp_f64 (~2^-23) and ONE cubic step, y0 (1 + e + e^2) = (1 / b)(1 - e^3) with e = 1 - b y0:
// 2^-69 before its rounding, i.e. as close to RN(1 / b) as the two Newton steps of the compiler's sequence get (one fused operation
// more), which is all the final correction q + (a - b q) y needs to land on the correctly rounded quotient (Markstein).  KR_LEAN_DIV_CUBIC=0:
// the two Newton steps.
#ifndef KR_LEAN_DIV_CUBIC
#define KR_LEAN_DIV_CUBIC 1
#endif
KR_DEV double lean_recip(double b)
{
    const double y0 = __builtin_amdgcn_rcp(b);
    const double e = __builtin_fma(-b, y0, 1.0);
#if KR_LEAN_DIV_CUBIC
    return __builtin_fma(y0, __builtin_fma(e, e, e), y0);
#else
    const double y1 = __builtin_fma(y0, e, y0);
    return __builtin_fma(y1, __builtin_fma(-b, y1, 1.0), y1);
#endif
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

// arithmetic policy of the strict path: LEAN (double, KR_LEAN_IEEE) or the compiler's sequences
template <bool LEAN> KR_DEV double dv(double a, double b) { if constexpr (LEAN) return lean_div(a, b); else return a / b; }
template <bool LEAN> KR_DEV float dv(float a, float b) { return a / b; }
template <bool LEAN> KR_DEV double sq(double x) { if constexpr (LEAN) return lean_sqrt(x); else return __builtin_sqrt(x); }
template <bool LEAN> KR_DEV float sq(float x) { return __builtin_sqrtf(x); }
// 1 / b for several quotients over b (lean: the refined reciprocal; otherwise unused) and the quotient that goes with it
template <bool LEAN> KR_DEV double dv_recip(double b) { if constexpr (LEAN) return lean_recip(b); else return 0.0; }
template <bool LEAN> KR_DEV float dv_recip(float) { return 0.0f; }
template <bool LEAN> KR_DEV double dv_y(double a, double b, double y) { if constexpr (LEAN) return lean_div_y(a, b, y); else return a / b; }
template <bool LEAN> KR_DEV float dv_y(float a, float b, float) { return a / b; }
KR_DEV double div_const(double a, double b, double inv_b, bool ok)
{
#if KR_LEAN_IEEE
    return div_by_uniform(a, b, inv_b, ok);
#else
    return a / b;
#endif
}
KR_DEV float div_const(float a, float b, float, bool) { return a / b; }
template <typename T> struct LeanDefault { static constexpr bool value = false; };
template <> struct LeanDefault<double> { static constexpr bool value = (KR_LEAN_IEEE != 0); };

KR_DEV bool kr_finite(double x) { return __builtin_fabs(x) < __builtin_inf(); }
KR_DEV bool kr_finite(float x) { return __builtin_fabsf(x) < __builtin_inff(); }

KR_DEV double kr_sqrt(double x) { return __builtin_sqrt(x); }
KR_DEV float kr_sqrt(float x) { return __builtin_sqrtf(x); }
// CR: the correctly rounded strict routine (default) or the shorter < 1-ulp one (kr_sincos.hpp::kr_sincos_t).  KR_RK45_CR_SINCOS: which of the
// two the RK45 bodies use.  1 (default): with the correctly rounded controller root below, 96-99 % of a PointSource's strict RK45 rays carry the
// reference's bits in every output (short kernels: 80-90 %; before either: 67-71 %) and the rest agree to 1e-10 instead of 7e-8; costs 7 % of a
// 1e7-ray RK45 launch (380 -> 406 ms: the longest ray's 1e5 steps each evaluate seven sin/cos pairs).
#ifndef KR_RK45_CR_SINCOS
#define KR_RK45_CR_SINCOS 1
#endif
constexpr bool kRk45CrSincos = (KR_RK45_CR_SINCOS != 0);
template <bool CR = true> KR_DEV void kr_sincos(double x, double& s, double& c) { kr_sincos_t<CR && (KR_CR_SINCOS != 0)>(x, s, c); }
// float: evaluated in double and rounded once -- correctly rounded in all but ~1e-8 of the arguments, which is what glibc's sinf / cosf / powf / tanf
// (the float instantiation's libm, <= 0.56 ulp) are in all but a few per cent: the float kernels then differ from the reference's float build only
// where glibc's own float routines are not correctly rounded (the device library's float routines: <= 1-2 ulp).  KR_F32_VIA_F64=0: the device library.
#ifndef KR_F32_VIA_F64
#define KR_F32_VIA_F64 1
#endif
template <bool CR = true> KR_DEV void kr_sincos(float x, float& s, float& c)
{
#if KR_F32_VIA_F64
    double sd, cd;
    kr_sincos_fast_f64((double) x, sd, cd);
    s = (float) sd;
    c = (float) cd;
#else
    ::sincosf(x, &s, &c);
#endif
}
// double: the strict path's own correctly rounded pair (kr_sincos.hpp) instead of the device libm (<= 1 ulp): the O(N) passes, the ray sources
// and the FlatPlane stop test then differ from glibc only where glibc is not correctly rounded; the unused half is dead code
KR_DEV double kr_sin(double x) { double s, c; kr_sincos_t<(KR_CR_SINCOS != 0)>(x, s, c); return s; }
KR_DEV float kr_sin(float x) { float s, c; kr_sincos<true>(x, s, c); return s; }
KR_DEV double kr_cos(double x) { double s, c; kr_sincos_t<(KR_CR_SINCOS != 0)>(x, s, c); return c; }
KR_DEV float kr_cos(float x) { float s, c; kr_sincos<true>(x, s, c); return c; }
KR_DEV double kr_tan(double x) { return ::tan(x); }
KR_DEV float kr_tan(float x) { return KR_F32_VIA_F64 ? (float) ::tan((double) x) : ::tanf(x); }
KR_DEV double kr_pow(double x, double y) { return ::pow(x, y); }
KR_DEV float kr_pow(float x, float y) { return KR_F32_VIA_F64 ? (float) ::pow((double) x, (double) y) : ::powf(x, y); }

// x^(1/5) for the DOPRI5 step controller (raytracer.cpp:1517: pow(1/max(err, 1e-10), 0.2), then 0.9 x that clamped to
// [0.1, 5]).  The clamp makes the root matter only for x in [1.7e-5, 5.3e3]; x is first brought into [1e-6, 1e6], which
// cannot change the clamped factor, so that a single-precision seed is always in range.  Seed from the hardware log2 / exp2
// (~1e-7), two Newton steps y <- y (4 + x / y^5) / 5 (error 2 e^2 each) a residual correction and the factor that
// turns the exact root into x^0.2 with the double constant 0.2, correctly rounded (tests/test_gpu_primitives.py), ~45 instructions against ~200 for the library pow.  NaN in, NaN out.
#ifndef KR_FIFTH_ROOT
#define KR_FIFTH_ROOT 1
#endif
KR_DEV double fifth_root_for_controller(double x)
{
#if KR_FIFTH_ROOT
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
#else
    return ::pow(x, 0.2);
#endif
}
KR_DEV float fifth_root_for_controller(float x) { return KR_F32_VIA_F64 ? (float) ::pow((double) x, (double) 0.2f) : ::powf(x, 0.2f); }
KR_DEV double kr_log(double x) { return ::log(x); }
KR_DEV double kr_acos(double x) { return ::acos(x); }
KR_DEV double kr_asin(double x) { return ::asin(x); }
KR_DEV double kr_atan2(double y, double x) { return ::atan2(y, x); }
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

// ---- per-launch constants (kernarg) -----------------------------------------------------------
template <typename T> struct TraceConsts {
    T a, horizon, rlim, thetalim;
    T precision, theta_precision, max_tstep, maxtstep_rlim, max_phistep, tol;
    T sp0, sp1, sp2;        // stop_params
    T inv_precision, inv_theta_precision;   // RN(1/precision), RN(1/theta_precision): fast path, and div_by_uniform on the strict one
    bool inv_ok;                            // both divisors qualify for div_by_uniform
    bool rk45_extrapolate;                  // RK45: creeping captured rays are extrapolated to the step limit (default)
    int32_t steplim;
    int32_t stop_kind;
    // The optional clauses of the step heuristic ("if (max_tstep > 0 && r < maxtstep_rlim) ...", "if (max_phistep > 0) ...", "if (rlim > 0 && ...)",
    // "if (thetalim > 0 && ...)", raytracer.cpp:855-871) with the launch-uniform half folded into the constant the per-ray half compares against:
    // a disabled clause gets the value that makes its own comparison false for every ray (r < -inf; step > |inf / phidot|; x > +inf).  Same
    // decisions on every ray, and no uniform lane masks for the compiler to carry (and spill) through the step loop.
    T tstep_rlim_eff;       // max_tstep > 0 ? maxtstep_rlim : -inf
    T phistep_eff;          // max_phistep > 0 ? max_phistep : +inf
    // ("rlim > 0 && r + rdot step > rlim" needs no constant of its own: with rlim <= 0 the loop condition r < rlim admits no step at all.
    // Likewise "thetalim > 0 && theta + thetadot step > thetalim" compares against theta_hi below: +inf unless thetalim > 0.)
    // loop condition of the theta-limit overloads, (tl > 0 && theta < tl) || (tl < 0 && theta > |tl|) || tl == 0  (:799), as theta_lo < theta < theta_hi:
    // tl > 0: (-inf, tl);  tl < 0: (|tl|, +inf);  tl == 0: (-inf, +inf);  NaN: empty (theta_hi = -inf)
    T theta_lo, theta_hi;
};

// ---- per-lane ray state ---------------------------------------------------------------------
template <typename T> struct Lane {
    T t, r, theta, phi;
    T pt, pr, ptheta, pphi;
    T k, h, Q;
    int32_t rdot_sign, thetadot_sign, rdot_flips, eq_cross;
    int32_t steps;          // steps taken in THIS call (the reference's local `steps`)
    int32_t status;
    int32_t steps0;         // rays[i].steps on entry
    bool r_was_positive, theta_was_positive;   // per-call locals, raytracer.cpp:767-768
    // RK45 only
    T step;                 // running step size
    T theta_eq_prev;
    T theta_prev;
    bool in_retry;          // a trial step was rejected: next iteration retries with the same k1
    int32_t creep_m;        // theta advanced by exactly this many ulps in the last creeping outer step (0: none) ...
    int32_t creep_run;      // ... and in this many consecutive outer steps before it
    bool creep_mode;        // the rest of the ray is replayed step by step from k1 alone (step_rk45)
    T creep_dt, creep_dphi; // its t and phi increments per step
    // fast arithmetic, fixed-step integrators: sin / cos of theta carried from the previous step's base point by angle addition (step_fixed)
    bool carry_ok;
    T carry_sin, carry_cos;
    // RK45, strict arithmetic: what the accepted trial's last stage (k7) already knows about the point the next step's k1 is taken at
    bool fsal_valid;
    T f_sin2theta, f_rhosq, f_delta, f_pt, f_thetadotsq, f_abs_ptheta;
};

// momentum_from_consts, src/include/kerr.h:300-335
template <typename T, bool LEAN, bool CR_SINCOS = true>
KR_DEV void momentum_impl(T& pt, T& pr, T& ptheta, T& pphi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T r, T theta, T a, Lane<T>* keep = nullptr)
{
    T sin_theta, cos_theta;
    kr_sincos<CR_SINCOS>(theta, sin_theta, cos_theta);
    const T sin2theta = sin_theta * sin_theta;
    const T rhosq = r * r + (a * cos_theta) * (a * cos_theta);
    const T delta = r * r - 2 * r + a * a;
    const T rhosq_delta = rhosq * delta;
    pt = (rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h;
    pt = dv<LEAN>(pt, rhosq_delta);

    pphi = 2 * a * r * sin2theta * k + (rhosq - 2 * r) * h;
    pphi = dv<LEAN>(pphi, sin2theta * rhosq_delta);

    const T hcs = dv<LEAN>(h * cos_theta, sin_theta);
    T thetadotsq = Q + (k * a * cos_theta + hcs) * (k * a * cos_theta - hcs);
    thetadotsq = dv<LEAN>(thetadotsq, rhosq * rhosq);
    const T abs_ptheta = sq<LEAN>(kr_abs(thetadotsq));
    ptheta = abs_ptheta * thetadot_sign;
    if (keep) {      // (see k1_from_last_stage)
        keep->f_sin2theta = sin2theta; keep->f_rhosq = rhosq; keep->f_delta = delta; keep->f_pt = pt; keep->f_thetadotsq = thetadotsq; keep->f_abs_ptheta = abs_ptheta;
    }

    T rdotsq = k * pt - h * pphi - rhosq * ptheta * ptheta;
    rdotsq = dv<LEAN>(rdotsq * delta, rhosq);
    pr = sq<LEAN>(kr_abs(rdotsq)) * rdot_sign;
}

template <typename T>
KR_DEV void momentum(T& pt, T& pr, T& ptheta, T& pphi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T r, T theta, T a)
{
    momentum_impl<T, LeanDefault<T>::value>(pt, pr, ptheta, pphi, k, h, Q, rdot_sign, thetadot_sign, r, theta, a);
}

// k1 at the current position with turning-point logic; identical in all five reference propagators
// (raytracer.cpp:177-222, :805-849, :1086-1130, :1370-1398, :1680-1708).  RK45_ASSOC selects the RK45
// bodies' association of the phidot denominator ((sin2theta*rhosq)*delta, :1375 vs :818).
// Returns true when the reference would `continue` (theta turning point: sign flipped, nothing moves).
template <typename T, bool RK45_ASSOC, bool LEAN>
KR_DEV bool k1_impl(Lane<T>& s, T a, T& rhosq_o, T& sin2theta_o, T* y_rhosq_o = nullptr)
{
    const T r = s.r, theta = s.theta, k = s.k, h = s.h;
    T sin_theta, cos_theta;
    kr_sincos<(!RK45_ASSOC || kRk45CrSincos)>(theta, sin_theta, cos_theta);          // (RK45_ASSOC <=> called from the RK45 bodies)
    const T sin2theta = sin_theta * sin_theta;
    const T rhosq = r * r + (a * cos_theta) * (a * cos_theta);
    const T delta = r * r - 2 * r + a * a;
    if (RK45_ASSOC) {
        s.pt = dv<LEAN>((rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h, rhosq * delta);
        s.pphi = dv<LEAN>(2 * a * r * sin2theta * k + (rhosq - 2 * r) * h, sin2theta * rhosq * delta);
    } else {
        const T rhosq_delta = rhosq * delta;
        s.pt = (rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h;
        s.pt = dv<LEAN>(s.pt, rhosq_delta);
        s.pphi = 2 * a * r * sin2theta * k + (rhosq - 2 * r) * h;
        s.pphi = dv<LEAN>(s.pphi, sin2theta * rhosq_delta);
    }

    const T hcs = dv<LEAN>(h * cos_theta, sin_theta);
    T thetadotsq = s.Q + (k * a * cos_theta + hcs) * (k * a * cos_theta - hcs);
    thetadotsq = dv<LEAN>(thetadotsq, rhosq * rhosq);

    if (thetadotsq < 0 && s.theta_was_positive) {
        s.thetadot_sign = -s.thetadot_sign;
        s.theta_was_positive = false;
        return true;
    }
    if (thetadotsq >= 0) s.theta_was_positive = true;

    s.ptheta = sq<LEAN>(kr_abs(thetadotsq)) * s.thetadot_sign;

    T rdotsq = k * s.pt - h * s.pphi - rhosq * s.ptheta * s.ptheta;
    const T y_rhosq = dv_recip<LEAN>(rhosq);          // (the caller's two flag quotients over rho^2 reuse it: step_fixed)
    rdotsq = dv_y<LEAN>(rdotsq * delta, rhosq, y_rhosq);
    if (rdotsq <= 0 && s.r_was_positive) {
        s.rdot_sign = -s.rdot_sign;
        s.r_was_positive = false;
        s.rdot_flips++;
    } else if (rdotsq > 0) {
        s.r_was_positive = true;
    }
    s.pr = sq<LEAN>(kr_abs(rdotsq)) * s.rdot_sign;

    rhosq_o = rhosq;
    sin2theta_o = sin2theta;
    if (y_rhosq_o) *y_rhosq_o = y_rhosq;
    return false;
}

template <typename T, bool RK45_ASSOC>
KR_DEV bool k1_with_flips(Lane<T>& s, T a, T& rhosq_o, T& sin2theta_o, T* y_rhosq_o = nullptr)
{
    return k1_impl<T, RK45_ASSOC, LeanDefault<T>::value>(s, a, rhosq_o, sin2theta_o, y_rhosq_o);
}

// RK45: the k1 of a step that follows an ACCEPTED trial is taken at the point that trial's last stage (k7) was evaluated at, and most
// of it is the same arithmetic on the same operands: sin^2, rho^2, Delta, tdot, thetadot^2 and |thetadot| come out bit for bit as k7
// had them (momentum_impl above and k1_impl evaluate identical expressions; the reference recomputes them, :1370-1398).  What differs
// is phidot -- the RK45 bodies associate its denominator as (sin^2 rho^2) Delta, kerr.h as sin^2 (rho^2 Delta) -- and rdot^2, which
// depends on it; those, the turning-point logic and the signs are done here as k1_impl does them.  ~70-115 of a trial step's ~1400
// instructions, on every accepted step; exact.  (Strict arithmetic, double precision.)
template <typename T, bool LEAN>
KR_DEV bool k1_from_last_stage(Lane<T>& s, T a, T& rhosq_o, T& sin2theta_o)
{
    const T r = s.r, k = s.k, h = s.h;
    const T sin2theta = s.f_sin2theta, rhosq = s.f_rhosq, delta = s.f_delta;
    s.pt = s.f_pt;
    s.pphi = dv<LEAN>(2 * a * r * sin2theta * k + (rhosq - 2 * r) * h, sin2theta * rhosq * delta);
    const T thetadotsq = s.f_thetadotsq;
    if (thetadotsq < 0 && s.theta_was_positive) {
        s.thetadot_sign = -s.thetadot_sign;
        s.theta_was_positive = false;
        return true;
    }
    if (thetadotsq >= 0) s.theta_was_positive = true;
    s.ptheta = s.f_abs_ptheta * s.thetadot_sign;
    T rdotsq = k * s.pt - h * s.pphi - rhosq * s.ptheta * s.ptheta;
    rdotsq = dv<LEAN>(rdotsq * delta, rhosq);
    if (rdotsq <= 0 && s.r_was_positive) {
        s.rdot_sign = -s.rdot_sign;
        s.r_was_positive = false;
        s.rdot_flips++;
    } else if (rdotsq > 0) {
        s.r_was_positive = true;
    }
    s.pr = sq<LEAN>(kr_abs(rdotsq)) * s.rdot_sign;
    rhosq_o = rhosq;
    sin2theta_o = sin2theta;
    return false;
}

// ==== fast-arithmetic path (kr_params.flags & KR_FLAG_FAST_MATH, double only) =====================================
// On gfx950 an IEEE fp64 division costs ~67 SIMD-cycles per wave-instruction and an IEEE sqrt ~92, against ~5.4
// for an FMA (scripts/microbench/fp64_peak.hip); the reference's formulation has 5 divisions + 2 square roots per
// derivative evaluation and 6-9 more divisions in the step heuristic, i.e. about half of an RK4 step.  This path
// evaluates the SAME formulas with one reciprocal per derivative evaluation (1/(rho^2 Delta sin^2 theta), from which
// 1/(rho^2 Delta), 1/rho^2 and 1/sin^2 follow by multiplication), reciprocal-multiply in the heuristic, Newton-refined
// v_rcp_f64 / v_rsq_f64 (<= ~1 ulp) and FMA contraction.  Results differ from the strict path by a few ulp per
// operation -- the same order as the libm difference that already separates the strict path from the CPU -- and are
// held to the same parity tolerances (tests/test_gpu_parity.py runs both).
#ifndef KR_RCP_CUBIC
#define KR_RCP_CUBIC 1
#endif
KR_DEV double fast_rcp(double x)
{
    // v_rcp_f64 is good to ~2^-23; with e = 1 - x y one CUBIC step y (1 + e + e^2) = (1/x)(1 - e^3) lands at 2^-69 before its own rounding
    // (<= 1 ulp, tests/test_gpu_primitives.py) in three fused operations -- two Newton steps, the textbook route to the same accuracy, take four
#if KR_RCP_CUBIC
    const double y = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, y, 1.0);
    const double p = __builtin_fma(e, e, e);
    return __builtin_fma(y, p, y);
#else
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
#endif
}

// One Newton step (relative error ~2^-46): for the step-size heuristic only, whose quotients end up under min() / as a step length
// (a step that is 1e-14 longer moves the sample point along the same trajectory; the landing steps hit r_max / theta_max to 1e-16).
#ifndef KR_HEURISTIC_RCP_SHORT
#define KR_HEURISTIC_RCP_SHORT 1
#endif
KR_DEV double fast_rcp_heur(double x)
{
#if KR_HEURISTIC_RCP_SHORT
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
#else
    return fast_rcp(x);
#endif
}
KR_DEV float fast_rcp_heur(float x) { return fast_rcp(x); }

// max(|x|, 1e-300) as ONE v_max_f64 with the |.| source modifier.  (Left to the compiler, fmax(fabs(x), c) on a value that has been through
// an integer operation or a select costs a v_and, a v_mov and a canonicalising v_max x, x first: four instructions, twice per k1.)
KR_DEV double abs_floor(double x)
{
    double r;
    asm("v_max_f64 %0, |%1|, %2" : "=v"(r) : "v"(x), "s"(1e-300));
    return r;
}

// sqrt(max(|x|, 1e-300)): rsq seed + one coupled Newton step + a residual correction.  inv (optional): 1 / that root to ~2^-46 -- the seed
// times (1 + e), with the e the root computes anyway: the step heuristic's 1 / |rdot| and 1 / |thetadot| for ONE more fused operation each
// instead of a v_rcp_f64 (a quarter-rate instruction: 16 issue cycles against 4) and its Newton step.
KR_DEV double fast_sqrt(double x, double* inv = nullptr)
{
    // The floor replaces the x == 0 / x == inf special cases of a plain rsq-based root (5 instructions per call, 8 calls per
    // RK4 step) by one v_max: a vanishing theta-dot or r-dot becomes 1e-150 instead of 0, which no later operation can tell
    // apart (it is added to O(1) angles / radii, and its reciprocal only feeds step-size minima).  +inf gives NaN.
    x = abs_floor(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double h = 0.5 * y;
    const double e = __builtin_fma(-h, g, 0.5);
    if (inv) *inv = __builtin_fma(y, e, y);
    g = __builtin_fma(g, e, g);
    // (the residual d is 2^-45 of x after the coupled step; the seed's h = 1 / (2 sqrt x) to 2^-23 scales it well enough -- the refined h
    // of the textbook sequence would buy 2^-90 instead of 2^-68 before the final rounding, one instruction per root, eight roots per RK4 step)
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

struct FastAux { double sin2theta, inv_rhosq, sn, cs, inv_abs_pr, inv_abs_ptheta; };

// The four derivatives through the separated potentials (Carter): with P = (r^2 + a^2) k - a h,
//   rho^2 tdot   = -a (a k sin^2 - h) + (r^2 + a^2) P / Delta        rho^4 thetadot^2 = Q + cos^2 (k^2 a^2 - h^2 / sin^2)  =: N
//   rho^2 phidot = -(a k - h / sin^2) + a P / Delta                  rho^4 rdot^2     = P^2 - Delta (Q + (h - a k)^2) - Delta (|N| - N)  =: R
// -- algebraically what kerr.h:300-335 evaluates (its rdot^2 = (k tdot - h phidot - rho^2 thetadot^2) Delta / rho^2 is the null condition
// solved for rdot), and the radial equation does not wait for tdot and phidot.  (kerr.h:327-333 builds rdot^2 from |thetadot^2|: beyond a
// polar turning point, where a Runge-Kutta stage may land, that differs from the analytic radial potential by 2 |thetadot^2| Delta -- the
// last term of R: an O(step^3) kink the reference's solution contains, so it is kept.)  One reciprocal, 1 / (rho^2 Delta sin^2), from which
// 1 / (rho^2 Delta), 1 / rho^2 and 1 / sin^2 follow by multiplication; the roots are taken of N and R and scaled by 1 / rho^2 afterwards.
// Every fused multiply-add of the fast path is written out: with "#pragma clang fp contract(fast)" the compiler chose them per kernel
// instance, and the same ray came out an ulp apart from the single-trace and the multi-trace kernels.
struct FastPotentials { double N, R, inv_rho, s2, rhosq; };

KR_DEV FastPotentials potentials_fast(double& pt, double& pphi, double k, double h, double Q, double r, double s, double c, double a)
{
    const double s2 = s * s;
    const double c2 = c * c;
    const double r2 = r * r;
    const double a2 = a * a;
    const double r2a2 = r2 + a2;
    const double rhosq = __builtin_fma(a2, c2, r2);
    const double delta = __builtin_fma(-2.0, r, r2a2);
    const double rd = rhosq * delta;
    const double inv = fast_rcp(rd * s2);          // 1 / (rho^2 Delta sin^2)
    const double inv_rd = inv * s2;                // 1 / (rho^2 Delta)
    const double inv_rho = inv_rd * delta;         // 1 / rho^2
    const double inv_s2 = inv * rd;                // 1 / sin^2
    const double ak = a * k, ah = a * h;           // (invariant along a ray, like h^2, k^2 a^2 and Q + (h - a k)^2 below: computed once per step)
    const double P = __builtin_fma(r2a2, k, -ah);
    pt = __builtin_fma(-(__builtin_fma(a * ak, s2, -ah)), inv_rho, (r2a2 * P) * inv_rd);
    pphi = __builtin_fma(__builtin_fma(h, inv_s2, -ak), inv_rho, (a * P) * inv_rd);
    FastPotentials o;
    o.N = __builtin_fma(c2, __builtin_fma(-(h * h), inv_s2, ak * ak), Q);
    const double hmak = h - ak;
    o.R = __builtin_fma(-delta, __builtin_fabs(o.N) - o.N, __builtin_fma(-delta, __builtin_fma(hmak, hmak, Q), P * P));
    o.inv_rho = inv_rho;
    o.s2 = s2;
    o.rhosq = rhosq;
    return o;
}

// momentum_from_consts (kerr.h:300-335)
KR_DEV void momentum_fast_sc(double& pt, double& pr, double& ptheta, double& pphi, double k, double h, double Q, int rdot_sign,
                             int thetadot_sign, double r, double s, double c, double a)
{
    const FastPotentials o = potentials_fast(pt, pphi, k, h, Q, r, s, c, a);
    ptheta = fast_sqrt(o.N) * (o.inv_rho * thetadot_sign);
    pr = fast_sqrt(o.R) * (o.inv_rho * rdot_sign);
}

KR_DEV void momentum_fast(double& pt, double& pr, double& ptheta, double& pphi, double k, double h, double Q, int rdot_sign,
                          int thetadot_sign, double r, double theta, double a)
{
    double s, c;
    kr_sincos_fast_f64(theta, s, c);
    momentum_fast_sc(pt, pr, ptheta, pphi, k, h, Q, rdot_sign, thetadot_sign, r, s, c, a);
}

// sin/cos of theta0 + d from those of theta0 (the stages of one Runge-Kutta step sit within a few per cent of a radian of its
// base point: |d| <= theta0/50 by the step heuristic unless the MIN_STEP floor is active).  Angle addition with 11th / 10th
// order Taylor kernels: truncation < 3e-20 for |d| <= 1/8; the sums s0 + (...) keep the rounding at ~1 ulp of the larger
// operand.  Outside that range, or when the stage lies across the pole (|d| > theta0 / 2: the sine is a cancellation), the
// full routine is used.
#ifndef KR_STAGE_SINCOS_NEAR
#define KR_STAGE_SINCOS_NEAR 1
#endif
// Horner steps of sincos_near: the coefficient as a scalar-register operand (KR_NEAR_SGPR, default) or a vector register the compiler re-creates per step
#ifndef KR_NEAR_SGPR
#define KR_NEAR_SGPR 1
#endif
#if KR_NEAR_SGPR
#define KR_NEAR_FMA(a, b, c) kr_fma3s((a), (b), (c))
#else
#define KR_NEAR_FMA(a, b, c) kr_fma3((a), (b), KR_K(c))
#endif
#ifndef KR_NEAR_LIMIT
#define KR_NEAR_LIMIT 0.07
#endif
// largest |d| for which sincos_near() uses the angle addition from theta0 (computed once per step, shared by its stages)
KR_DEV double sincos_near_limit(double theta0)
{
    return __builtin_fmin(KR_NEAR_LIMIT, 0.5 * __builtin_fmin(__builtin_fabs(theta0), __builtin_fabs(kPi - theta0)));
}

KR_DEV void sincos_near(double s0, double c0, double d, double& s, double& c)   // valid for |d| <= sincos_near_limit(theta0)
{
    // |d| <= 0.07 (the step heuristic keeps a whole RK4 step within theta / 50 <= 0.063): sin d through d^9 (next term d^11 / 11! <= 5e-21),
    // cos d - 1 through d^8 (next d^10 / 10! <= 8e-19 of a sum of magnitude ~1): one Horner step less on each side than the 1/8 version
    const double d2 = d * d;
    double ps = KR_NEAR_FMA(d2, 1.0 / 362880.0, -1.0 / 5040.0);
    ps = KR_NEAR_FMA(ps, d2, 1.0 / 120.0);
    ps = KR_NEAR_FMA(ps, d2, -1.0 / 6.0);
    const double sd = __builtin_fma(d * d2, ps, d);                 // sin d
    double pc = KR_NEAR_FMA(d2, 1.0 / 40320.0, -1.0 / 720.0);
    pc = KR_NEAR_FMA(pc, d2, 1.0 / 24.0);
    pc = __builtin_fma(pc, d2, -0.5);
    const double cm = d2 * pc;                                       // cos d - 1
    s = __builtin_fma(c0, sd, __builtin_fma(s0, cm, s0));
    c = __builtin_fma(-s0, sd, __builtin_fma(c0, cm, c0));
}

// k1 with the turning-point logic (see k1_with_flips) on the fast path
#ifndef KR_CARRY_SINCOS
#define KR_CARRY_SINCOS 0
#endif
template <bool CARRY = false>
KR_DEV bool k1_with_flips_fast(Lane<double>& s, double a, FastAux& aux)
{
    double sn, c;
    if constexpr (CARRY && KR_CARRY_SINCOS) {
        // sin / cos of the base point: what the previous step derived by angle addition from ITS base point (step_fixed), when every lane of the
        // wave has such a pair; else the full routine, and each lane still takes its own carried pair where it has one (which pair a ray
        // uses is a function of that ray's history alone -- never of its neighbours in the wave)
        // (the base point of a fixed-step integrator lies in [0, pi] -- reflect_poles -- so the routine's core is called directly: its
        // out-of-range fallback is an out-of-line call that takes the addresses of its results, and with those in scope the compiler kept
        // sn / c in scratch memory on this path too)
        if (__builtin_amdgcn_ballot_w64(!s.carry_ok) == 0) {
            sn = s.carry_sin; c = s.carry_cos;
        } else {
            double sf, cf;
            kr_sincos_fast_core_f64(s.theta, sf, cf);
            sn = s.carry_ok ? s.carry_sin : sf;
            c = s.carry_ok ? s.carry_cos : cf;
        }
    } else {
        kr_sincos_fast_f64(s.theta, sn, c);
    }
    const FastPotentials o = potentials_fast(s.pt, s.pphi, s.k, s.h, s.Q, s.r, sn, c, a);
    // thetadot^2 = N / rho^4 and rdot^2 = R / rho^4 have the signs of N and R
    if (o.N < 0 && s.theta_was_positive) {
        s.thetadot_sign = -s.thetadot_sign;
        s.theta_was_positive = false;
        return true;
    }
    if (o.N >= 0) s.theta_was_positive = true;
    double inv_root;
    s.ptheta = fast_sqrt(o.N, &inv_root) * (o.inv_rho * s.thetadot_sign);
    aux.inv_abs_ptheta = inv_root * o.rhosq;                   // 1 / |thetadot| = rho^2 / sqrt |N|
    if (o.R <= 0 && s.r_was_positive) {
        s.rdot_sign = -s.rdot_sign;
        s.r_was_positive = false;
        s.rdot_flips++;
    } else if (o.R > 0) {
        s.r_was_positive = true;
    }
    s.pr = fast_sqrt(o.R, &inv_root) * (o.inv_rho * s.rdot_sign);
    aux.inv_abs_pr = inv_root * o.rhosq;
    aux.sin2theta = o.s2; aux.inv_rhosq = o.inv_rho; aux.sn = sn; aux.cs = c;
    return false;
}

// one derivative evaluation on either path
template <typename T, bool FAST, bool CR_SINCOS = true>
KR_DEV void eval(T& pt, T& pr, T& ptheta, T& pphi, const Lane<T>& s, T r, T theta, T a)
{
    if constexpr (FAST) momentum_fast(pt, pr, ptheta, pphi, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r, theta, a);
    else momentum_impl<T, LeanDefault<T>::value, CR_SINCOS>(pt, pr, ptheta, pphi, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r, theta, a);
}

// loop condition of the theta-limit overloads (raytracer.cpp:172, :799, :1362-1364) or of the
// RayDestination overloads (:1080, :1674)
template <typename T, bool USE_DEST>
KR_DEV bool loop_cond(const Lane<T>& s, const TraceConsts<T>& c)
{
    bool ok = s.r < c.rlim && s.steps < c.steplim;
    if (!USE_DEST) {
        ok = ok && s.theta < c.theta_hi && s.theta > c.theta_lo;        // (TraceConsts::theta_lo / theta_hi)
    }
    return ok;
}

// RayDestination::reached(r, theta, phi, prev_theta): ray_destination.h:90-94 (FlatDisc, through the
// default crossing-aware overload :52-54), :130-142 (DiscWithISCO), :184-190 (FlatPlane)
template <typename T>
KR_DEV bool dest_reached(const TraceConsts<T>& c, T r, T theta, T phi, T prev_theta)
{
    if (c.stop_kind == KR_STOP_FLATDISC) {
        const T tl = c.sp0;
        if (tl > 0) return theta >= tl;
        if (tl < 0) return theta <= -tl;
        return false;
    }
    if (c.stop_kind == KR_STOP_DISC_ISCO) {
        const T r_isco = c.sp0, r_out = c.sp1, tl = c.sp2;
        if (r < r_isco) return false;
        if (r_out > 0 && r > r_out) return false;
        if (tl > 0) return (prev_theta < tl && theta >= tl) || (prev_theta > tl && theta <= tl);
        if (tl < 0) {
            const T m = -tl;
            return (prev_theta > m && theta <= m) || (prev_theta < m && theta >= m);
        }
        return false;
    }
    // KR_STOP_FLATPLANE
    const T incl = c.sp0, phi0 = c.sp1, z_s = c.sp2;
    const T proj = r * (kr_sin(theta) * kr_sin(incl) * kr_cos(phi - phi0) + kr_cos(theta) * kr_cos(incl));
    return proj <= -z_s;
}

// RayDestination::step_limit(): ray_destination.h:55-57 (base), :95-101, :143-151
template <typename T>
KR_DEV T dest_step_limit(const TraceConsts<T>& c, T r, T theta, T ptheta)
{
    T tl;
    if (c.stop_kind == KR_STOP_FLATDISC) {
        tl = c.sp0;
    } else if (c.stop_kind == KR_STOP_DISC_ISCO) {
        if (r < c.sp0) return Lim<T>::max();
        if (c.sp1 > 0 && r > c.sp1) return Lim<T>::max();
        tl = c.sp2;
    } else {
        return Lim<T>::max();
    }
    if (tl > 0 && ptheta > 0 && theta < tl) return (tl - theta) / ptheta;
    if (tl < 0 && ptheta < 0 && theta > -tl) return (-tl - theta) / ptheta;
    return Lim<T>::max();
}

// polar reflection, raytracer.cpp:282-283 / :914-915 / :1498-1499
template <typename T>
KR_DEV void reflect_poles(T& theta, T& phi, int32_t& thetadot_sign)
{
    // a pole crossing is rare: one wave-uniform test, and the per-lane selects only in a wave that has one
    if (__builtin_amdgcn_ballot_w64(theta < T(0) || theta > T(kPi)) == 0) return;
    if (theta < T(0)) { theta = -theta; thetadot_sign = -thetadot_sign; phi += T(kPi); }
    if (theta > T(kPi)) { theta = T(2) * T(kPi) - theta; thetadot_sign = -thetadot_sign; phi += T(kPi); }
}

template <typename T>
KR_DEV bool crossed_equator(T before, T after)
{
    return ((double) before < kPi2 && (double) after >= kPi2) || ((double) before > kPi2 && (double) after <= kPi2);
}

// One iteration of the Euler (raytracer.cpp:172-313) or RK4 (:799-943, :1080-1229) loop body.
// Returns true when the ray has finished (break, or the loop condition no longer holds).
template <typename T, bool RK4, bool USE_DEST, bool FAST>
KR_DEV bool step_fixed(Lane<T>& s, const TraceConsts<T>& c)
{
    const T a = c.a;
    ++s.steps;

    T step;
    T pt1, pr1, ptheta1, pphi1;
    FastAux aux;
    if constexpr (FAST) {
        // (sin / cos carried from step to step: Euler only.  For RK4 it removes 7 of 406 vector instructions per step but costs registers
        // the stage code needs -- 63.2 ms against 62.7 at 1e7 rays, profiles/r03_ab_experiments.txt)
        if (k1_with_flips_fast<!RK4>(s, a, aux)) return !(s.steps < c.steplim);
        pt1 = s.pt; pr1 = s.pr; ptheta1 = s.ptheta; pphi1 = s.pphi;
        // The same heuristic (:855-871) with ONE quarter-rate instruction instead of four.  1 / |rdot| and 1 / |thetadot| come out of the
        // square roots that produced them (fast_sqrt); the time and azimuth caps, min(dt / |tdot|, dphi / |phidot|), share one reciprocal:
        // min(dt |phidot|, dphi |tdot|) / (|tdot| |phidot|).  A cap that is switched off is +inf in the numerator (TraceConsts), a NaN or 0 / 0
        // quotient leaves the step as it is (v_min ignores a NaN operand) -- as the reference's "step > x" does.
        const T inv_pr = aux.inv_abs_pr, inv_pth = aux.inv_abs_ptheta;          // magnitudes: every use below takes |.| anyway
        const T q_th = kr_abs(s.theta) * inv_pth;
        const T dr = s.r - c.horizon;
        step = (kr_abs(dr) * inv_pr) * c.inv_precision;
        if (step > q_th * c.inv_precision) step = q_th * c.inv_theta_precision;
        {
            const T apt = kr_abs(pt1), aphi = kr_abs(pphi1);
            const T dt_eff = (s.r < c.tstep_rlim_eff) ? c.max_tstep : T(1e300);
            const T num = __builtin_fmin(dt_eff * aphi, c.phistep_eff * apt);
            step = __builtin_fmin(step, num * fast_rcp_heur(apt * aphi));
        }
        // "if (step < MIN_STEP) step = MIN_STEP" as v_max: differs from the comparison only when `step` is NaN, i.e. when r or rdot is -- and
        // then r is NaN after this step whatever its length
        step = __builtin_fmax(step, T(KR_MIN_STEP));
        // the two landing clips apply on a ray's LAST step only: one fused test each, the clip itself behind a wave-uniform branch (the empty
        // asm keeps the compiler from turning the branch back into unconditional arithmetic and selects).  The clipped step is the reference's
        // correctly rounded quotient: it decides whether theta + thetadot step lands ON the limit or an ulp short of it (= one more step).
        // With the heuristic's approximate reciprocal (2^-44) 23 of 1e6 Euler rays of a lamp post at a = 0 took that extra step
        // (profiles/r03_hybrid_sweep_euler.jsonl, tests/tool_gpu_euler_diff.py).
        {
            // (the fused sum is within an ulp of the reference's rounded product + rounded sum: ">=" lets every ray through that the reference's
            // own test, made inside, could clip)
            const bool near_r = __builtin_fma(pr1, step, s.r) >= c.rlim;
            if (__builtin_amdgcn_ballot_w64(near_r) != 0) {
                asm volatile("" ::: "memory");
                if (near_r && s.r + pr1 * step > c.rlim) step = kr_abs(lean_div(c.rlim - s.r, pr1));
            }
        }
        if (!USE_DEST) {
            const bool near_th = __builtin_fma(ptheta1, step, s.theta) >= c.theta_hi;
            if (__builtin_amdgcn_ballot_w64(near_th) != 0) {
                asm volatile("" ::: "memory");
                if (near_th && s.theta + ptheta1 * step > c.theta_hi) step = kr_abs(lean_div(c.theta_hi - s.theta, ptheta1));
            }
        }
        if (pt1 <= 0) s.status |= KR_STATUS_ERGO;
        // (1 - 2r/rho^2) tdot + (2 a r sin^2/rho^2) phidot IS the conserved energy k (= -p_t): analytically it cannot turn negative, and
        // numerically only where its two terms (~ k / Delta) are 1e15 times k.  Away from the horizon, for k > 0, the test is skipped
        // (wave-uniform; a NaN k or r takes the evaluation, whose comparison is then false as in the reference).
        if (__builtin_amdgcn_ballot_w64(!(dr > T(1e-6)) || !(s.k > T(0))) != 0) {
            const T two_r_rho = 2 * s.r * aux.inv_rhosq;
            if ((1 - two_r_rho) * pt1 + (two_r_rho * a * aux.sin2theta) * pphi1 < 0) s.status |= KR_STATUS_NEG_ENERGY;
        }
    } else {
    T rhosq, sin2theta, y_rhosq;
    if (k1_with_flips<T, false>(s, a, rhosq, sin2theta, &y_rhosq)) return !(s.steps < c.steplim);   // r, theta unchanged
    pt1 = s.pt; pr1 = s.pr; ptheta1 = s.ptheta; pphi1 = s.pphi;

    // step-size heuristic (:224-243 / :855-871 / :1136-1151)
    step = div_const(kr_abs(dv<LeanDefault<T>::value>(s.r - c.horizon, pr1)), c.precision, c.inv_precision, c.inv_ok);
    {
        const T q_th = kr_abs(dv<LeanDefault<T>::value>(s.theta, ptheta1));
        if (step > div_const(q_th, c.precision, c.inv_precision, c.inv_ok)) step = div_const(q_th, c.theta_precision, c.inv_theta_precision, c.inv_ok);
    }
    if (s.r < c.tstep_rlim_eff) {                       // max_tstep > 0 && r < maxtstep_rlim  (TraceConsts)
        const T st = kr_abs(dv<LeanDefault<T>::value>(c.max_tstep, pt1));
        if (step > st) step = st;
    }
    {                                                   // max_phistep > 0: otherwise the quotient is inf / NaN and the comparison false
        const T sp = kr_abs(dv<LeanDefault<T>::value>(c.phistep_eff, pphi1));
        if (step > sp) step = sp;
    }
    if ((double) step < KR_MIN_STEP) step = T(KR_MIN_STEP);
    if (s.r + pr1 * step > c.rlim) step = kr_abs(dv<LeanDefault<T>::value>(c.rlim - s.r, pr1));
    if (!USE_DEST) {
        if (s.theta + ptheta1 * step > c.theta_hi) step = kr_abs(dv<LeanDefault<T>::value>(c.theta_hi - s.theta, ptheta1));
    }

    // flags (:264-273 / :874-887); neither ends the ray
    if (pt1 <= 0) s.status |= KR_STATUS_ERGO;
    // (1 - 2r/rho^2) tdot + (2 a r sin^2/rho^2) phidot is the conserved energy k (= -p_t) evaluated from tdot and phidot: its two terms are
    // <= ~(r^2 + a^2)^2 k / (rho^2 Delta) in size, so with r - r_horizon > 1e-6 (Delta > 1e-9 for every a < 0.99999) their rounding errors, 1e-16
    // of the terms, stay below 1e-7 k: for k > 0 the sum cannot come out negative, in the reference or here.  The flag is therefore only
    // evaluated -- with the reference's operations -- by waves in which some ray is that close to the horizon or has k <= 0 / NaN.
    if (sizeof(T) == 4 || __builtin_amdgcn_ballot_w64(!(s.r - c.horizon > T(1e-6)) || !(s.k > T(0))) != 0) {
        if ((1 - dv_y<LeanDefault<T>::value>(2 * s.r, rhosq, y_rhosq)) * pt1 + dv_y<LeanDefault<T>::value>(2 * a * s.r * sin2theta, rhosq, y_rhosq) * pphi1 < 0) s.status |= KR_STATUS_NEG_ENERGY;
    }
    }

    const T theta_prev = s.theta;
    if (!RK4) {
        s.t += pt1 * step;
        s.r += pr1 * step;
        s.theta += ptheta1 * step;
        s.phi += pphi1 * step;
    } else {
        // k2..k4 use k1's signs and move only (r, theta)  (:889-905)
        // stage evaluation; the fast path gets sin/cos of the stage angle from those of the base point
        // Stages 2-4.  Fast path: sin/cos of the stage angles come from the base point's by angle addition, which is valid
        // while every stage stays within near_limit of it -- practically always.  The stages are therefore computed
        // optimistically, branch-free, and in the rare other case all three are redone with the full routine (one branch
        // per step instead of one per stage; nothing of `s` has been touched yet).
        double near_limit = 0;
        if constexpr (FAST) near_limit = sincos_near_limit(s.theta);
        T acc_t, acc_phi, acc_r, acc_theta, pt4, pr4, ptheta4, pphi4;
        auto stages = [&](auto near) -> bool {
            constexpr bool kNear = decltype(near)::value;
            bool within = true;
            auto stage = [&](T& pt, T& pr, T& ptheta, T& pphi, T r_stage, T dtheta) {
                if constexpr (FAST) {
                    double sn, cs;
                    if constexpr (kNear) {
                        within = within && (__builtin_fabs(dtheta) <= near_limit);
                        sincos_near(aux.sn, aux.cs, dtheta, sn, cs);
                    } else {
                        kr_sincos_fast_f64(s.theta + dtheta, sn, cs);
                    }
                    momentum_fast_sc(pt, pr, ptheta, pphi, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r_stage, sn, cs, a);
                } else {
                    eval<T, false>(pt, pr, ptheta, pphi, s, r_stage, s.theta + dtheta, a);
                }
            };
            // x1 + 2 x2 as ONE fused multiply-add: 2 x2 is exact, so fma(2, x2, x1) rounds the same sum once -- the reference's bits (:908-912)
            // on either path, one instruction instead of two; the stage radius r + h pr is fused on the fast path only (the product rounds)
            const T half = step / 2;
            auto at = [&](T h, T v) -> T { if constexpr (FAST) return __builtin_fma(h, v, s.r); else return s.r + h * v; };
            T pt2, pr2, ptheta2, pphi2;
            stage(pt2, pr2, ptheta2, pphi2, at(half, pr1), half * ptheta1);
            acc_t = kr_fma2(pt2, pt1);
            acc_phi = kr_fma2(pphi2, pphi1);
            T pt3, pr3, ptheta3, pphi3;
            stage(pt3, pr3, ptheta3, pphi3, at(half, pr2), half * ptheta2);
            acc_t = kr_fma2(pt3, acc_t);
            acc_phi = kr_fma2(pphi3, acc_phi);
            acc_r = kr_fma2(pr3, kr_fma2(pr2, pr1));
            acc_theta = kr_fma2(ptheta3, kr_fma2(ptheta2, ptheta1));
            stage(pt4, pr4, ptheta4, pphi4, at(step, pr3), step * ptheta3);
            return within;
        };
        if constexpr (FAST && KR_STAGE_SINCOS_NEAR) {
            if (!stages(std::true_type{})) stages(std::false_type{});
        } else {
            stages(std::false_type{});
        }
        // x += (step/6)(k1 + 2k2 + 2k3 + k4), summed left to right as in :908-912
        const T w = FAST ? step * T(1.0 / 6.0) : div_const(step, T(6), T(1.0 / 6.0), true);
        if constexpr (FAST) {
            s.t = __builtin_fma(w, acc_t + pt4, s.t);
            s.r = __builtin_fma(w, acc_r + pr4, s.r);
            s.theta = __builtin_fma(w, acc_theta + ptheta4, s.theta);
            s.phi = __builtin_fma(w, acc_phi + pphi4, s.phi);
        } else {
            s.t += w * (acc_t + pt4);
            s.r += w * (acc_r + pr4);
            s.theta += w * (acc_theta + ptheta4);
            s.phi += w * (acc_phi + pphi4);
        }
    }
    if (crossed_equator(theta_prev, s.theta)) ++s.eq_cross;
    if constexpr (FAST && !RK4 && KR_CARRY_SINCOS) {
        // The next step's base point is theta_prev + dth with dth = theta - theta_prev EXACT (the two are within a factor of two of each other):
        // its sin / cos follow from this step's by the angle addition the RK4 stages use -- 14 operations instead of the ~40 of a reduction,
        // two polynomials and a quadrant fix-up.  Every addition leaves <= 1 ulp in the pair, as the integration leaves half an ulp in theta
        // itself; every 1024th step of a ray, and whenever the increment is out of the addition's range or the ray went over a pole, the
        // pair is taken afresh.
        const T dth = s.theta - theta_prev;
        double sn, cs;
        sincos_near(aux.sn, aux.cs, dth, sn, cs);
        s.carry_sin = sn; s.carry_cos = cs;
        s.carry_ok = (__builtin_fabs(dth) <= sincos_near_limit(theta_prev)) && ((s.steps & 1023) != 0) && !(s.theta < T(0) || s.theta > T(kPi));
    }
    reflect_poles(s.theta, s.phi, s.thetadot_sign);

    if (s.r <= c.horizon) { s.status |= KR_STATUS_HORIZON; return true; }
    if (USE_DEST) {
        if (dest_reached(c, s.r, s.theta, s.phi, theta_prev)) { s.status |= KR_STATUS_DEST; return true; }
    }
    return !loop_cond<T, USE_DEST>(s, c);
}

#ifndef KR_CREEP_RUN
#define KR_CREEP_RUN 8      // consecutive creeping outer steps (same ulp count) before the rest of a captured ray is extrapolated
#endif

// ---- RK45 / DOPRI5 (raytracer.cpp:1260-1598, :1600-1894) ----------------------------------------
template <typename T> struct Dopri {
    // Butcher tableau, :1316-1330, formed exactly as T(n)/d
    static constexpr T a21 = T(1) / 5;
    static constexpr T a31 = T(3) / 40, a32 = T(9) / 40;
    static constexpr T a41 = T(44) / 45, a42 = T(-56) / 15, a43 = T(32) / 9;
    static constexpr T a51 = T(19372) / 6561, a52 = T(-25360) / 2187, a53 = T(64448) / 6561, a54 = T(-212) / 729;
    static constexpr T a61 = T(9017) / 3168, a62 = T(-355) / 33, a63 = T(46732) / 5247, a64 = T(49) / 176, a65 = T(-5103) / 18656;
    static constexpr T b1 = T(35) / 384, b3 = T(500) / 1113, b4 = T(125) / 192, b5 = T(-2187) / 6784, b6 = T(11) / 84;
    static constexpr T e1 = T(71) / 57600, e3 = T(-71) / 16695, e4 = T(71) / 1920, e5 = T(-17253) / 339200, e6 = T(22) / 525, e7 = T(-1) / 40;
};

// Seeds the running step when a ray enters propagate_rk45 (:1341-1359): k1 WITHOUT flip logic, heuristic
// WITHOUT boundary clips, theta test guarded by |thetadot| > 0 and compared against theta_precision.
template <typename T>
KR_DEV void rk45_seed(Lane<T>& s, const TraceConsts<T>& c)
{
    const T a = c.a, r = s.r, theta = s.theta, k = s.k, h = s.h;
    T sin_theta, cos_theta;
    kr_sincos<kRk45CrSincos>(theta, sin_theta, cos_theta);
    const T sin2theta = sin_theta * sin_theta;
    const T rhosq = r * r + (a * cos_theta) * (a * cos_theta);
    const T delta = r * r - 2 * r + a * a;
    s.pt = ((rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h) / (rhosq * delta);
    s.pphi = (2 * a * r * sin2theta * k + (rhosq - 2 * r) * h) / (sin2theta * rhosq * delta);
    const T hcs = h * cos_theta / sin_theta;
    const T thetadotsq = (s.Q + (k * a * cos_theta + hcs) * (k * a * cos_theta - hcs)) / (rhosq * rhosq);
    s.ptheta = kr_sqrt(kr_abs(thetadotsq)) * s.thetadot_sign;
    const T rdotsq = (k * s.pt - h * s.pphi - rhosq * s.ptheta * s.ptheta) * delta / rhosq;
    s.pr = kr_sqrt(kr_abs(rdotsq)) * s.rdot_sign;

    T step = kr_abs((r - c.horizon) / s.pr) / c.precision;
    if (kr_abs(s.ptheta) > 0 && step > kr_abs(theta / s.ptheta) / c.theta_precision) step = kr_abs(theta / s.ptheta) / c.theta_precision;
    if (r < c.tstep_rlim_eff && step > kr_abs(c.max_tstep / s.pt)) step = kr_abs(c.max_tstep / s.pt);
    if (step > kr_abs(c.phistep_eff / s.pphi)) step = kr_abs(c.phistep_eff / s.pphi);
    if ((double) step < KR_MIN_STEP) step = T(KR_MIN_STEP);
    s.step = step;
    s.theta_eq_prev = theta;
    s.in_retry = false;
}

// One outer step of a lane in creep mode (see the end of step_rk45): what the full step would do at its start -- ++steps, k1
// with the turning-point tests, the ERGO / NEG_ENERGY flags, all on the real code path -- then the increments that are known:
// r stays, theta moves by creep_m ulps (integer arithmetic on its bits), t and phi by their recorded increments (these two,
// and the momenta left in the record, are accurate to ~1e-11 rather than to the bit).  If k1 does anything but confirm the
// state (a sign flip, a turning-point flag), everything is put back and the lane returns to full steps.  When both status
// bits can no longer change, the remaining steps are applied at once.  Returns 1: ray finished, 0: continue, -1: left creep mode.
template <typename T, bool USE_DEST, bool FAST>
KR_DEV int creep_step(Lane<T>& s, const TraceConsts<T>& c, uint32_t& attempts, uint32_t& creep_steps)
{
    const T a = c.a;
    const Lane<T> keep = s;
    ++s.steps;
    bool confirmed;
    if constexpr (FAST) {
        FastAux aux;
        confirmed = !k1_with_flips_fast(s, a, aux);
        if (confirmed) {
            if (s.pt <= 0) s.status |= KR_STATUS_ERGO;
            const T two_r_rho = 2 * s.r * aux.inv_rhosq;
            if ((1 - two_r_rho) * s.pt + (two_r_rho * a * aux.sin2theta) * s.pphi < 0) s.status |= KR_STATUS_NEG_ENERGY;
        }
    } else {
        T rhosq, sin2theta;
        confirmed = !k1_with_flips<T, true>(s, a, rhosq, sin2theta);
        if (confirmed) {
            if (s.pt <= 0) s.status |= KR_STATUS_ERGO;
            if ((1 - 2 * s.r / rhosq) * s.pt + (2 * a * s.r * sin2theta / rhosq) * s.pphi < 0) s.status |= KR_STATUS_NEG_ENERGY;
        }
    }
    if (!confirmed || s.rdot_sign != keep.rdot_sign || s.thetadot_sign != keep.thetadot_sign || s.rdot_flips != keep.rdot_flips ||
        s.r_was_positive != keep.r_was_positive || s.theta_was_positive != keep.theta_was_positive) {
        s = keep;
        s.creep_mode = false;
        s.creep_run = 0;
        s.creep_m = 0;
        return -1;
    }
    long long todo = 1;
    if ((s.status & KR_STATUS_NEG_ENERGY) && ((s.status & KR_STATUS_ERGO) || s.pt > T(1))) {
        todo = 1 + ((long long) c.steplim - s.steps);        // neither flag can change any more: this step and all the remaining ones
        s.steps = c.steplim;
    }
    s.fsal_valid = false;                                    // theta moves without a last stage having been evaluated there
    const long long bits = (long long) __builtin_bit_cast(unsigned long long, (double) s.theta) + (long long) s.creep_m * todo;
    s.theta = (T) __builtin_bit_cast(double, (unsigned long long) bits);
    s.t = s.t + (T) todo * s.creep_dt;
    s.phi = s.phi + (T) todo * s.creep_dphi;
    attempts += (uint32_t) todo;
    creep_steps += (uint32_t) todo;
    return (s.steps < c.steplim) ? 0 : 1;
}

// One wave iteration of RK45 = at most one TRIAL step per lane.  The reference nests a retry loop inside
// the outer step (:1438-1541); here a rejected lane keeps its k1 (s.pt..s.pphi hold k1 until a trial is
// accepted) and retries on the next iteration, so a rejection never stalls the other 63 lanes.
// attempts/rejects are per-lane counters.  Returns true when the ray has finished.
template <typename T, bool USE_DEST, bool FAST>
KR_DEV bool step_rk45(Lane<T>& s, const TraceConsts<T>& c, uint32_t& attempts, uint32_t& rejects, uint32_t& stationary_steps, uint32_t& creep_steps,
                      int replay_batch)
{
    using D = Dopri<T>;
    const T a = c.a;

    if constexpr (sizeof(T) == 8) {
        if (s.creep_mode) {
            // replay_batch > 1 when every ray of the wave is in creep mode (the tail of a launch): several outer steps per wave iteration
            for (int u = 0; u < replay_batch; ++u) {
                const int rc = creep_step<T, USE_DEST, FAST>(s, c, attempts, creep_steps);
                if (rc > 0) return true;
                if (rc < 0) break;                 // back to full steps, starting with this one
                if (u + 1 == replay_batch) return false;
            }
        }
    }

    // snapshot of every variable that feeds back into the next outer step (for the fixed-point test below)
    const bool fresh = !s.in_retry;
    const T step_in = s.step;
    const int32_t rs_in = s.rdot_sign, ts_in = s.thetadot_sign;
    const bool rwp_in = s.r_was_positive, twp_in = s.theta_was_positive;

    if (!s.in_retry) {
        ++s.steps;
        T step_max;
        if constexpr (FAST) {
            FastAux aux;
            if (k1_with_flips_fast(s, a, aux)) return !(s.steps < c.steplim);
            if (s.pt <= 0) s.status |= KR_STATUS_ERGO;
            const T two_r_rho = 2 * s.r * aux.inv_rhosq;
            if ((1 - two_r_rho) * s.pt + (two_r_rho * a * aux.sin2theta) * s.pphi < 0) s.status |= KR_STATUS_NEG_ENERGY;
            step_max = kr_abs((s.r - c.horizon) * fast_rcp(s.pr)) * c.inv_precision;
            {                                              // (switched off: phistep_eff = +inf, tstep_rlim_eff = -inf -- TraceConsts)
                const T step_phi = kr_abs(c.phistep_eff * fast_rcp(s.pphi));
                if (step_phi < step_max) step_max = step_phi;
            }
            if (s.r < c.tstep_rlim_eff) {
                const T step_t = kr_abs(c.max_tstep * fast_rcp(s.pt));
                if (step_t < step_max) step_max = step_t;
            }
        } else {
        T rhosq, sin2theta;
        bool flipped;
        if constexpr (sizeof(T) == 8) {
            // wave-uniform: every lane's data from its last accepted stage is valid (else all recompute -- same bits either way)
            if (__builtin_amdgcn_ballot_w64(!s.fsal_valid) == 0) flipped = k1_from_last_stage<T, LeanDefault<T>::value>(s, a, rhosq, sin2theta);
            else flipped = k1_with_flips<T, true>(s, a, rhosq, sin2theta);
        } else {
            flipped = k1_with_flips<T, true>(s, a, rhosq, sin2theta);
        }
        if (flipped) return !(s.steps < c.steplim);
        // flags (:1403-1410): same rhosq / sin2theta values as k1's
        if (s.pt <= 0) s.status |= KR_STATUS_ERGO;
        if ((1 - 2 * s.r / rhosq) * s.pt + (2 * a * s.r * sin2theta / rhosq) * s.pphi < 0) s.status |= KR_STATUS_NEG_ENERGY;
        // outer cap (:1421-1434): horizon / phi / t, no MIN_STEP floor afterwards
        step_max = kr_abs((s.r - c.horizon) / s.pr) / c.precision;
        {                                                  // max_phistep > 0: otherwise the quotient is inf / NaN and the comparison false
            const T step_phi = kr_abs(c.phistep_eff / s.pphi);
            if (step_phi < step_max) step_max = step_phi;
        }
        if (s.r < c.tstep_rlim_eff) {                      // max_tstep > 0 && r < maxtstep_rlim
            const T step_t = kr_abs(c.max_tstep / s.pt);
            if (step_t < step_max) step_max = step_t;
        }
        }
        if (s.step > step_max) s.step = step_max;
        s.theta_prev = s.theta;
    }
    const T pt1 = s.pt, pr1 = s.pr, ptheta1 = s.ptheta, pphi1 = s.pphi;
    const T r = s.r, theta = s.theta;

    // trial step with boundary clamps (:1442-1453 / :1745-1755)
    T h_try = s.step;
    bool clamped = false;
    if (!USE_DEST) {
        if (theta + ptheta1 * h_try > c.theta_hi) {      // thetalim > 0 && ...
            const T h_th = kr_abs((c.theta_hi - theta) / ptheta1);      // (= thetalim: the clamp only fires for thetalim > 0)
            if (h_th < h_try) { h_try = h_th; clamped = true; }
        }
    } else {
        if (r + pr1 * h_try > c.rlim) { h_try = kr_abs((c.rlim - r) / pr1); clamped = true; }      // rlim > 0 && ...
        const T h_dest = dest_step_limit(c, r, theta, ptheta1);
        if (h_dest < h_try) { h_try = h_dest; clamped = true; }
    }
    ++attempts;

    // stages 2..6; the b- and e-weighted sums are accumulated in stage order, which is the reference's
    // left-to-right order (:1493-1496, :1508-1509), so only (pr_i, ptheta_i) stay live across stages
    T pt_i, pphi_i;
    T pr2, ptheta2, pr3, ptheta3, pr4, ptheta4, pr5, ptheta5, pr6, ptheta6;
    T sum_t = D::b1 * pt1, sum_phi = D::b1 * pphi1;

    eval<T, FAST, kRk45CrSincos>(pt_i, pr2, ptheta2, pphi_i, s, r + h_try * D::a21 * pr1,
             theta + h_try * D::a21 * ptheta1, a);

    eval<T, FAST, kRk45CrSincos>(pt_i, pr3, ptheta3, pphi_i, s, r + h_try * (D::a31 * pr1 + D::a32 * pr2),
             theta + h_try * (D::a31 * ptheta1 + D::a32 * ptheta2), a);
    sum_t = sum_t + D::b3 * pt_i;
    sum_phi = sum_phi + D::b3 * pphi_i;

    eval<T, FAST, kRk45CrSincos>(pt_i, pr4, ptheta4, pphi_i, s,
             r + h_try * (D::a41 * pr1 + D::a42 * pr2 + D::a43 * pr3),
             theta + h_try * (D::a41 * ptheta1 + D::a42 * ptheta2 + D::a43 * ptheta3), a);
    sum_t = sum_t + D::b4 * pt_i;
    sum_phi = sum_phi + D::b4 * pphi_i;

    eval<T, FAST, kRk45CrSincos>(pt_i, pr5, ptheta5, pphi_i, s,
             r + h_try * (D::a51 * pr1 + D::a52 * pr2 + D::a53 * pr3 + D::a54 * pr4),
             theta + h_try * (D::a51 * ptheta1 + D::a52 * ptheta2 + D::a53 * ptheta3 + D::a54 * ptheta4), a);
    sum_t = sum_t + D::b5 * pt_i;
    sum_phi = sum_phi + D::b5 * pphi_i;

    eval<T, FAST, kRk45CrSincos>(pt_i, pr6, ptheta6, pphi_i, s,
             r + h_try * (D::a61 * pr1 + D::a62 * pr2 + D::a63 * pr3 + D::a64 * pr4 + D::a65 * pr5),
             theta + h_try * (D::a61 * ptheta1 + D::a62 * ptheta2 + D::a63 * ptheta3 + D::a64 * ptheta4 + D::a65 * ptheta5), a);
    sum_t = sum_t + D::b6 * pt_i;
    sum_phi = sum_phi + D::b6 * pphi_i;

    // 5th-order solution (:1493-1499); the polar reflection mutates thetadot_sign even if the trial is rejected
    const T inc_r = h_try * (D::b1 * pr1 + D::b3 * pr3 + D::b4 * pr4 + D::b5 * pr5 + D::b6 * pr6);
    const T inc_theta = h_try * (D::b1 * ptheta1 + D::b3 * ptheta3 + D::b4 * ptheta4 + D::b5 * ptheta5 + D::b6 * ptheta6);
    T r_new = r + inc_r;
    T theta_new = theta + inc_theta;
    T t_new = s.t + h_try * sum_t;
    T phi_new = s.phi + h_try * sum_phi;
    const bool inside_poles = !(theta_new < T(0)) && !(theta_new > T(kPi));
    reflect_poles(theta_new, phi_new, s.thetadot_sign);

    T pt7, pr7, ptheta7, pphi7;
    Lane<T> last;                   // (only its f_* members are written, and only on the strict double path)
    if constexpr (!FAST && sizeof(T) == 8)
        momentum_impl<T, LeanDefault<T>::value, kRk45CrSincos>(pt7, pr7, ptheta7, pphi7, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r_new, theta_new, a, &last);
    else
        eval<T, FAST, kRk45CrSincos>(pt7, pr7, ptheta7, pphi7, s, r_new, theta_new, a);

    // error norm over (r, theta) and the step controller (:1508-1519)
    const T err_r = h_try * (D::e1 * pr1 + D::e3 * pr3 + D::e4 * pr4 + D::e5 * pr5 + D::e6 * pr6 + D::e7 * pr7);
    const T err_theta = h_try * (D::e1 * ptheta1 + D::e3 * ptheta3 + D::e4 * ptheta4 + D::e5 * ptheta5 + D::e6 * ptheta6 + D::e7 * ptheta7);
    const T sc_r = c.tol * (T(1) + std_max(kr_abs(r), kr_abs(r_new)));
    const T sc_theta = c.tol * (T(1) + std_max(kr_abs(theta), kr_abs(theta_new)));
    // The norm itself is never stored: it only decides -- accept (<= 1), the controller's factor (5 whenever <= 1.8e-4, below), the creep test
    // (<= 0.5).  A ray whose step is set by a cap rather than by its error (the polar-axis ray's 100 000 steps, which bound every RK45 launch)
    // sits orders of magnitude below 1.8e-4: two raw reciprocals (2^-22) show that with a 1 % margin, every decision is then known, and the two
    // IEEE quotients and the IEEE root (36 instructions of a lone wave's ~1000 per trial) are left out.  Wave-uniform; NaN takes the exact path.
    T err_norm;
    bool surely_saturated = false;
    if constexpr (sizeof(T) == 8) {
        const double qr = (double) err_r * __builtin_amdgcn_rcp((double) sc_r), qt = (double) err_theta * __builtin_amdgcn_rcp((double) sc_theta);
        surely_saturated = __builtin_fma(qr, qr, qt * qt) <= 6.4e-8;                 // (1.8e-4)^2 x 2 = 6.48e-8
    }
    if (__builtin_amdgcn_ballot_w64(!surely_saturated) == 0) err_norm = T(1e-4);      // (stands for "some value <= 1.8e-4")
    else err_norm = kr_sqrt(T(0.5) * ((err_r / sc_r) * (err_r / sc_r) + (err_theta / sc_theta) * (err_theta / sc_theta)));

    // 0.9 (1 / max(err, 1e-10))^0.2 clamped to [0.1, 5] (:1517-1518) IS 5 whenever err <= 1.889e-4 (0.9 x^0.2 >= 5 from x = 5292 on); a ray
    // whose step is set by a cap rather than by its error -- the polar-axis ray's 100 000 steps -- is there at every step, and the
    // root costs ~40 instructions.  err <= 1.8e-4 leaves a 1 % margin for the root's rounding; the choice is a pure function of err.
    const bool saturated = err_norm <= T(1.8e-4);
    T fac = T(5.0);
    if (__builtin_amdgcn_ballot_w64(!saturated) != 0) {
        T f = T(0.9) * fifth_root_for_controller(T(1) / std_max(err_norm, T(1e-10)));
        f = std_max(T(0.1), std_min(T(5.0), f));
        fac = saturated ? T(5.0) : f;
    }
    const T step_new = h_try * fac;

    bool commit = false;
    if (err_norm <= T(1)) {
        if (!clamped) s.step = std_max(step_new, T(KR_MIN_STEP));
        commit = true;
    } else {
        ++rejects;
        s.step = std_max(step_new, T(KR_MIN_STEP));
        if (s.step <= T(KR_MIN_STEP)) {
            commit = true;                       // cannot shrink further: force-accept (:1533-1539)
        } else if (err_norm != err_norm) {
            // NaN error norm: the reference never leaves its retry loop here.  End the ray (documented extension).
            s.status |= KR_STATUS_NAN;
            s.in_retry = false;
            return true;
        }
    }
    if (!commit) {
        s.in_retry = true;
        return false;
    }
    s.in_retry = false;
    s.t = t_new; s.r = r_new; s.theta = theta_new; s.phi = phi_new;
    s.pt = pt7; s.pr = pr7; s.ptheta = ptheta7; s.pphi = pphi7;
    if constexpr (!FAST && sizeof(T) == 8) {
        s.f_sin2theta = last.f_sin2theta; s.f_rhosq = last.f_rhosq; s.f_delta = last.f_delta; s.f_pt = last.f_pt;
        s.f_thetadotsq = last.f_thetadotsq; s.f_abs_ptheta = last.f_abs_ptheta;
        s.fsal_valid = true;
    }

    // Fixed point.  A ray captured by the hole ends up with r - r_horizon ~ 1e-14: the outer cap makes the
    // step so small that r and theta no longer change in fp64, the ray never reaches r <= horizon, and the
    // reference spins until RK45_STEPLIM (every such ray costs exactly 100 000 steps).
    // If this whole outer step was ONE trial and left every fed-back variable (r, theta, running step, both
    // signs, both turning-point flags) bit-identical to its value on entry, then every later outer step is this
    // same pure function of the same inputs: it adds the same two increments to t and phi, sets the same status
    // bits, and counts one step.  Replaying only those two additions gives bit-identical results; t and phi are
    // accumulated one addition at a time, exactly as the full loop would round them.  (phi feeds back only
    // through FlatPlaneDestination::reached, so that stop kind is excluded.)
    if (fresh && inside_poles && s.r == r && s.theta == theta && s.step == step_in && s.rdot_sign == rs_in && s.thetadot_sign == ts_in &&
        s.r_was_positive == rwp_in && s.theta_was_positive == twp_in && !(s.r <= c.horizon) && (!USE_DEST || c.stop_kind != KR_STOP_FLATPLANE) &&
        s.steps < c.steplim) {
        const T dt = h_try * sum_t, dphi = h_try * sum_phi;
        const int32_t remaining = c.steplim - s.steps;
        if constexpr (sizeof(T) == 8) {
            // the `remaining` additions to t and to phi, each rounded as the loop would round it, in closed form per binade
            // (kr_replay.hpp; bit-identical to the loop, which used to hold the other 63 lanes of the wave for ~0.7 ms per captured ray)
            s.t = kr_replay_additions(s.t, dt, (long long) remaining);
            s.phi = kr_replay_additions(s.phi, dphi, (long long) remaining);
        } else {
            for (int32_t i = 0; i < remaining; ++i) {
                s.t = s.t + dt;
                s.phi = s.phi + dphi;
            }
        }
        s.steps = c.steplim;
        attempts += (uint32_t) remaining;
        if (!(err_norm <= T(1))) rejects += (uint32_t) remaining;
        stationary_steps += (uint32_t) remaining;
        return true;
    }

    // Creep.  Most captured rays do not reach that fixed point: r is stationary (its increment is a fraction of an ulp) but
    // the theta increment h Sum(b_i thetadot_i) stays near a whole number m >= 1 of ulps, so theta advances by exactly m ulps
    // per outer step, for ever -- 100 000 steps of seven evaluations each for a ray no application uses (its step count is
    // stored negative).  Over the ~1e5 ulps still to go theta changes by 1e-11 of itself, and so does every quantity of the
    // step.  Once the step has been of this kind KR_CREEP_RUN times in a row, with margins that 1e-11 cannot consume (trial
    // accepted at half the tolerance; increments at least 1e-6 ulp away from the rounding boundaries at 1/2 ulp and
    // m +- 1/2 ulps; no equator / pole / stop angle / binade boundary inside the range theta will cover), the lane switches to
    // creep mode (creep_step below): each further outer step evaluates only what can still change the ray's integer outputs
    // -- k1 with its turning-point tests and the two status flags -- and applies the known increments.
    if constexpr (sizeof(T) == 8) {
        bool creeping = false;
        if (c.rk45_extrapolate && fresh && inside_poles && s.r == r && s.rdot_sign == rs_in && s.thetadot_sign == ts_in && s.r_was_positive == rwp_in &&
            s.theta_was_positive == twp_in && !(s.r <= c.horizon) && err_norm <= T(0.5) && !clamped &&
            (!USE_DEST || c.stop_kind != KR_STOP_FLATPLANE) && theta > T(0)) {
            const long long b0 = (long long) __builtin_bit_cast(unsigned long long, (double) theta);
            const long long b1 = (long long) __builtin_bit_cast(unsigned long long, (double) s.theta);
            const long long m = b1 - b0;
            const double ulp_th = __builtin_bit_cast(double, (unsigned long long) b0 & 0x7FF0000000000000ull) * 2.220446049250313e-16;
            const double ulp_r = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, (double) r) & 0x7FF0000000000000ull) * 2.220446049250313e-16;
            const long long am = m < 0 ? -m : m;
            if (am >= 1 && am <= 65536 && ((b0 ^ b1) >> 52) == 0 && __builtin_fabs((double) inc_theta - (double) m * ulp_th) <= 0.499999 * ulp_th &&
                __builtin_fabs((double) inc_r) <= 0.499999 * ulp_r) {
                creeping = true;
                s.creep_run = ((int32_t) m == s.creep_m) ? s.creep_run + 1 : 1;
                s.creep_m = (int32_t) m;
                const long long remaining = (long long) c.steplim - s.steps;
                if (s.creep_run >= KR_CREEP_RUN && remaining > 0) {
                    const long long b_end = b1 + m * remaining;
                    const double th_end = __builtin_bit_cast(double, (unsigned long long) b_end);
                    const double lo = __builtin_fmin((double) s.theta, th_end), hi = __builtin_fmax((double) s.theta, th_end);
                    auto outside = [&](double x) { return !(x >= lo && x <= hi); };      // x is not a value theta will take
                    if (((b1 ^ b_end) >> 52) == 0 && lo > 0.0 && hi < kPi && outside(kPi2) && outside(__builtin_fabs((double) c.thetalim)) &&
                        outside(__builtin_fabs((double) c.sp0)) && outside(__builtin_fabs((double) c.sp2))) {
                        s.creep_mode = true;
                        s.creep_dt = h_try * sum_t;
                        s.creep_dphi = h_try * sum_phi;
                    }
                }
            }
        }
        if (!creeping) { s.creep_run = 0; s.creep_m = 0; }
    }

    if (crossed_equator(s.theta_eq_prev, s.theta)) ++s.eq_cross;   // once per accepted outer step (:1542-1544)
    s.theta_eq_prev = s.theta;

    if (s.r <= c.horizon) { s.status |= KR_STATUS_HORIZON; return true; }
    if (USE_DEST) {
        if (dest_reached(c, s.r, s.theta, s.phi, s.theta_prev)) { s.status |= KR_STATUS_DEST; return true; }
    }
    return !loop_cond<T, USE_DEST>(s, c);
}

// Epilogue shared by all propagators (raytracer.cpp:315-339, :945-969, :1231-1253, :1574-1597, :1872-1893):
// final status bits and the value of rays[i].steps to store.
template <typename T, bool USE_DEST>
KR_DEV int32_t finish_status(Lane<T>& s, const TraceConsts<T>& c)
{
    if (s.steps >= c.steplim)
        s.status |= KR_STATUS_STEPLIM;
    else if (s.r >= c.rlim)
        s.status |= KR_STATUS_RLIM;
    else if (!USE_DEST && (s.theta >= c.theta_hi || s.theta <= c.theta_lo))       // (tl > 0 && theta >= tl) || (tl < 0 && theta <= |tl|)
        s.status |= KR_STATUS_DEST;
    int32_t out_steps = s.steps0;
    if (s.steps > 0) out_steps += s.steps;
    if (s.status & KR_STATUS_STEPLIM) out_steps = -out_steps;
    return out_steps;
}

}  // namespace kr
