// kr_device.hpp -- device-side Kerr null-geodesic arithmetic for gfx950 (MI355X).
//
// One ray per work-item; the whole ray state lives in VGPRs.  Everything here is a pure
// __device__ function of registers: no memory traffic, no LDS, no cross-lane ops.  The per-step
// semantics follow the reference propagators (file:line cited per function, paths relative to the
// reference tree); arithmetic is written with the reference's association so that, built with
// -ffp-contract=off, a step differs from the CPU result only through the device libm (sin, cos,
// pow: <= 1-2 ulp vs glibc), never through re-ordering.
//
// Layout: kr_arith.hpp (scalar building blocks) -> this file (per-launch constants, per-lane state, the strict derivative evaluation, the
// fixed-step integrators, the stop surfaces, the epilogue) -> kr_fast.hpp (the fast-arithmetic evaluation, included below) -> kr_rk45.hpp
// (the adaptive integrator, included at the end).
//
// T = double is the product precision; T = float mirrors the reference's second instantiation
// (raytracer.cpp:1896-1897).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kr_trace.h"
#include "kr_arith.hpp"

namespace kr {

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
    // fast path: the words of max_tstep, and the high word of "no cap" -- a number >= 8.9e307 that shares max_tstep's low word (step_fixed)
    uint32_t tstep_lo, tstep_on_hi, tstep_off_hi;
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
    int32_t steps0;         // rays[i].steps on entry (>= 0 for a traced ray); its SIGN BIT is this call's "always evaluate the NEG_ENERGY flag" mark (energy_guard)
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
    // RK45, strict arithmetic: what the accepted trial's last stage (k7) already knows about the point the next step's k1 is taken at
    bool fsal_valid;
    T f_sin2theta, f_rhosq, f_delta, f_pt, f_thetadotsq, f_abs_ptheta;
};

// The NEG_ENERGY flag, (1 - 2r/rho^2) tdot + (2 a r sin^2/rho^2) phidot < 0 (raytracer.cpp:264-273 / :874-887 / :1403-1410), is the conserved energy k
// evaluated from tdot and phidot.  Its two terms are <= ~(r^2 + a^2)^2 (k + |h| / r) / (rho^2 Delta) in size, so their rounding errors (1e-16 of the terms)
// stay far below k -- and the sum cannot come out negative, in the reference or here -- as long as the ray is not within 1e-6 of the horizon, k > 0 and
// |h| <= 1e6 k.  The per-ray half of that (k, h: constants of the ray) is decided ONCE, when a lane takes the ray, and kept in the sign bit of steps0;
// the per-step test is then "r - r_h > 1e-6 and the mark is clear" -- two compares, as before the |h| clause existed.
template <typename T> KR_DEV void energy_guard_set(Lane<T>& s)
{
    if (!(s.k > T(0)) || !(kr_abs(s.h) <= T(1e6) * s.k)) s.steps0 |= (int32_t) 0x80000000;       // (NaN k or h: marked)
}
template <typename T> KR_DEV bool energy_flag_needed(const Lane<T>& s, T r_minus_horizon) { return !(r_minus_horizon > T(1e-6)) || s.steps0 < 0; }
template <typename T> KR_DEV int32_t steps_on_entry(const Lane<T>& s) { return s.steps0 & 0x7fffffff; }

// The five denominators of one derivative evaluation (kerr.h:308-334: rho^2 Delta, sin^2 x that, sin, rho^4, rho^2) are products of one another, so
// in double precision three reciprocals are taken from the hardware -- of rho^2 Delta, sin and rho^2 -- and the other two are put together from them
// and polished (lean_recip_from): two quarter-rate instructions less per evaluation on the strict path, whose quotients stay the correctly rounded
// ones (tests/test_gpu_primitives.py; every ray record of 2e6-ray Euler / RK4 and 1e6-ray RK45 traces unchanged, profiles/r04_ab_experiments.txt).
// Every quotient is formed against the denominator AS THE REFERENCE ROUNDS IT (rho^2 x rho^2; sin^2 x (rho^2 Delta) or (sin^2 rho^2) Delta).
// (1 / rho^2 is NOT taken from 1 / (rho^2 Delta): exactly on the horizon, Delta = 0, the reference's thetadot^2 is still finite.)
template <typename T> struct StageRecips;
template <> struct StageRecips<double> {
    double rhosq_delta, sin_theta, sin2_rhosq_delta, rho4, rhosq;
    double y_rhosq_delta, y_sin, y_sin2_rhosq_delta, y_rho4, y_rhosq;
    KR_DEV StageRecips(double rhosq_delta_, double sin_theta_, double sin2_rhosq_delta_, double rhosq_, double delta)
        : rhosq_delta(rhosq_delta_), sin_theta(sin_theta_), sin2_rhosq_delta(sin2_rhosq_delta_), rho4(rhosq_ * rhosq_), rhosq(rhosq_)
    {
        y_rhosq_delta = lean_recip(rhosq_delta);
        y_sin = lean_recip(sin_theta);
        y_rhosq = lean_recip(rhosq);
        y_sin2_rhosq_delta = lean_recip_from(sin2_rhosq_delta, y_sin * (y_sin * y_rhosq_delta));      // (in this order: (1 / sin)^2 alone overflows for theta < 1e-154, long before the quotient does)
        y_rho4 = lean_recip_from(rho4, y_rhosq * y_rhosq);
    }
    KR_DEV double over_rhosq_delta(double x) const { return lean_div_y(x, rhosq_delta, y_rhosq_delta); }
    KR_DEV double over_sin(double x) const { return lean_div_y(x, sin_theta, y_sin); }
    KR_DEV double over_sin2_rhosq_delta(double x) const { return lean_div_y(x, sin2_rhosq_delta, y_sin2_rhosq_delta); }
    KR_DEV double over_rho4(double x) const { return lean_div_y(x, rho4, y_rho4); }
    KR_DEV double over_rhosq(double x) const { return lean_div_y(x, rhosq, y_rhosq); }
};
template <> struct StageRecips<float> {
    float rhosq_delta, sin_theta, sin2_rhosq_delta, rho4, rhosq, y_rhosq;
    KR_DEV StageRecips(float rhosq_delta_, float sin_theta_, float sin2_rhosq_delta_, float rhosq_, float)
        : rhosq_delta(rhosq_delta_), sin_theta(sin_theta_), sin2_rhosq_delta(sin2_rhosq_delta_), rho4(rhosq_ * rhosq_), rhosq(rhosq_), y_rhosq(0.0f) {}
    KR_DEV float over_rhosq_delta(float x) const { return x / rhosq_delta; }
    KR_DEV float over_sin(float x) const { return x / sin_theta; }
    KR_DEV float over_sin2_rhosq_delta(float x) const { return x / sin2_rhosq_delta; }
    KR_DEV float over_rho4(float x) const { return x / rho4; }
    KR_DEV float over_rhosq(float x) const { return x / rhosq; }
};

// momentum_from_consts, src/include/kerr.h:300-335
// SMALL_ANGLE (double): the caller vouches for |theta| < KR_SMALL_ANGLE_LIMIT (step_fixed's side-launch stages): kr_sincos_f64's small-angle branch
// without its test.
template <typename T, bool LONE = false, bool SMALL_ANGLE = false>
KR_DEV void momentum_impl(T& pt, T& pr, T& ptheta, T& pphi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T r, T theta, T a, Lane<T>* keep = nullptr)
{
    T sin_theta, cos_theta;
    if constexpr (SMALL_ANGLE && sizeof(T) == 8) kr_sincos_small_f64(theta, sin_theta, cos_theta);
    else kr_sincos<LONE>(theta, sin_theta, cos_theta);
    const T sin2theta = sin_theta * sin_theta;
    const T rhosq = r * r + (a * cos_theta) * (a * cos_theta);
    const T delta = r * r - 2 * r + a * a;
    const T rhosq_delta = rhosq * delta;
    const StageRecips<T> y(rhosq_delta, sin_theta, sin2theta * rhosq_delta, rhosq, delta);
    pt = (rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h;
    pt = y.over_rhosq_delta(pt);

    pphi = 2 * a * r * sin2theta * k + (rhosq - 2 * r) * h;
    pphi = y.over_sin2_rhosq_delta(pphi);

    const T hcs = y.over_sin(h * cos_theta);
    T thetadotsq = Q + (k * a * cos_theta + hcs) * (k * a * cos_theta - hcs);
    thetadotsq = y.over_rho4(thetadotsq);
    const T abs_ptheta = sq(kr_abs(thetadotsq));
    ptheta = abs_ptheta * thetadot_sign;
    if (keep) {      // (see k1_from_last_stage)
        keep->f_sin2theta = sin2theta; keep->f_rhosq = rhosq; keep->f_delta = delta; keep->f_pt = pt; keep->f_thetadotsq = thetadotsq; keep->f_abs_ptheta = abs_ptheta;
    }

    T rdotsq = k * pt - h * pphi - rhosq * ptheta * ptheta;
    rdotsq = y.over_rhosq(rdotsq * delta);
    pr = sq(kr_abs(rdotsq)) * rdot_sign;
}

template <typename T>
KR_DEV void momentum(T& pt, T& pr, T& ptheta, T& pphi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T r, T theta, T a)
{
    momentum_impl<T>(pt, pr, ptheta, pphi, k, h, Q, rdot_sign, thetadot_sign, r, theta, a);
}

// k1 at the current position with turning-point logic; identical in all five reference propagators
// (raytracer.cpp:177-222, :805-849, :1086-1130, :1370-1398, :1680-1708).  RK45_ASSOC selects the RK45
// bodies' association of the phidot denominator ((sin2theta*rhosq)*delta, :1375 vs :818).
// Returns true when the reference would `continue` (theta turning point: sign flipped, nothing moves).
template <typename T, bool RK45_ASSOC, bool LONE = false>
KR_DEV bool k1_with_flips(Lane<T>& s, T a, T& rhosq_o, T& sin2theta_o, T* y_rhosq_o = nullptr)
{
    const T r = s.r, theta = s.theta, k = s.k, h = s.h;
    T sin_theta, cos_theta;
    kr_sincos<LONE>(theta, sin_theta, cos_theta);
    const T sin2theta = sin_theta * sin_theta;
    const T rhosq = r * r + (a * cos_theta) * (a * cos_theta);
    const T delta = r * r - 2 * r + a * a;
    const T rhosq_delta = rhosq * delta;
    const StageRecips<T> y(rhosq_delta, sin_theta, RK45_ASSOC ? sin2theta * rhosq * delta : sin2theta * rhosq_delta, rhosq, delta);
    s.pt = y.over_rhosq_delta((rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h);
    s.pphi = y.over_sin2_rhosq_delta(2 * a * r * sin2theta * k + (rhosq - 2 * r) * h);

    const T hcs = y.over_sin(h * cos_theta);
    T thetadotsq = s.Q + (k * a * cos_theta + hcs) * (k * a * cos_theta - hcs);
    thetadotsq = y.over_rho4(thetadotsq);

    if (thetadotsq < 0 && s.theta_was_positive) {
        s.thetadot_sign = -s.thetadot_sign;
        s.theta_was_positive = false;
        return true;
    }
    if (thetadotsq >= 0) s.theta_was_positive = true;

    s.ptheta = sq(kr_abs(thetadotsq)) * s.thetadot_sign;

    T rdotsq = k * s.pt - h * s.pphi - rhosq * s.ptheta * s.ptheta;
    const T y_rhosq = y.y_rhosq;                // (the caller's two flag quotients over rho^2 reuse it: step_fixed)
    rdotsq = y.over_rhosq(rdotsq * delta);
    if constexpr (LONE) {
        // (the same bookkeeping as selects: the if / else costs a wave that owns its SIMD a taken branch around a one-instruction else arm on every step)
        const bool flip = rdotsq <= 0 && s.r_was_positive;
        s.rdot_sign = flip ? -s.rdot_sign : s.rdot_sign;
        s.rdot_flips += flip ? 1 : 0;
        s.r_was_positive = !flip && (s.r_was_positive || rdotsq > 0);
    } else if (rdotsq <= 0 && s.r_was_positive) {
        s.rdot_sign = -s.rdot_sign;
        s.r_was_positive = false;
        s.rdot_flips++;
    } else if (rdotsq > 0) {
        s.r_was_positive = true;
    }
    s.pr = sq(kr_abs(rdotsq)) * s.rdot_sign;

    rhosq_o = rhosq;
    sin2theta_o = sin2theta;
    if (y_rhosq_o) *y_rhosq_o = y_rhosq;
    return false;
}

// RK45: the k1 of a step that follows an ACCEPTED trial is taken at the point that trial's last stage (k7) was evaluated at, and most
// of it is the same arithmetic on the same operands: sin^2, rho^2, Delta, tdot, thetadot^2 and |thetadot| come out bit for bit as k7
// had them (momentum_impl above and k1_impl evaluate identical expressions; the reference recomputes them, :1370-1398).  What differs
// is phidot -- the RK45 bodies associate its denominator as (sin^2 rho^2) Delta, kerr.h as sin^2 (rho^2 Delta) -- and rdot^2, which
// depends on it; those, the turning-point logic and the signs are done here as k1_impl does them.  ~70-115 of a trial step's ~1400
// instructions, on every accepted step; exact.  (Strict arithmetic, double precision.)
template <typename T>
KR_DEV bool k1_from_last_stage(Lane<T>& s, T a, T& rhosq_o, T& sin2theta_o, T* y_rhosq_o = nullptr)
{
    const T r = s.r, k = s.k, h = s.h;
    const T sin2theta = s.f_sin2theta, rhosq = s.f_rhosq, delta = s.f_delta;
    s.pt = s.f_pt;
    s.pphi = dv(2 * a * r * sin2theta * k + (rhosq - 2 * r) * h, sin2theta * rhosq * delta);
    const T thetadotsq = s.f_thetadotsq;
    if (thetadotsq < 0 && s.theta_was_positive) {
        s.thetadot_sign = -s.thetadot_sign;
        s.theta_was_positive = false;
        return true;
    }
    if (thetadotsq >= 0) s.theta_was_positive = true;
    s.ptheta = s.f_abs_ptheta * s.thetadot_sign;
    T rdotsq = k * s.pt - h * s.pphi - rhosq * s.ptheta * s.ptheta;
    const T y_rhosq = dv_recip(rhosq);                 // (shared with the caller's two flag quotients over rho^2, as in k1_with_flips)
    rdotsq = dv_y(rdotsq * delta, rhosq, y_rhosq);
    if (rdotsq <= 0 && s.r_was_positive) {
        s.rdot_sign = -s.rdot_sign;
        s.r_was_positive = false;
        s.rdot_flips++;
    } else if (rdotsq > 0) {
        s.r_was_positive = true;
    }
    s.pr = sq(kr_abs(rdotsq)) * s.rdot_sign;
    rhosq_o = rhosq;
    sin2theta_o = sin2theta;
    if (y_rhosq_o) *y_rhosq_o = y_rhosq;
    return false;
}

}  // namespace kr

#include "kr_fast.hpp"

namespace kr {

// one derivative evaluation on either path
template <typename T, bool FAST, bool LONE = false>
KR_DEV void eval(T& pt, T& pr, T& ptheta, T& pphi, const Lane<T>& s, T r, T theta, T a)
{
    if constexpr (FAST) momentum_fast(pt, pr, ptheta, pphi, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r, theta, a);
    else momentum_impl<T, LONE>(pt, pr, ptheta, pphi, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r, theta, a);
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
template <typename T, bool LONE = false>
KR_DEV void reflect_poles(T& theta, T& phi, int32_t& thetadot_sign)
{
    // a pole crossing is rare: one wave-uniform test, and the per-lane selects only in a wave that has one
    if constexpr (LONE) {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(theta < T(0) || theta > T(kPi)) == 0, true)) return;     // (the usual path falls through)
    } else {
        if (__builtin_amdgcn_ballot_w64(theta < T(0) || theta > T(kPi)) == 0) return;
    }
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
template <typename T, bool RK4, bool USE_DEST, bool FAST, bool LONE = false>
KR_DEV bool step_fixed(Lane<T>& s, const TraceConsts<T>& c)
{
    const T a = c.a;
    ++s.steps;

    T step;
    T pt1, pr1, ptheta1, pphi1;
    FastAux aux;
    if constexpr (FAST) {
        // (carrying sin / cos of the base point from step to step by angle addition was measured and rejected: it removes 7 of 406 vector instructions
        // per RK4 step but costs registers the stage code needs -- 63.2 ms against 62.7 at 1e7 rays, Euler 31.6 against 30.9; profiles/r03_ab_experiments.txt)
        if (k1_with_flips_fast(s, a, aux)) return !(s.steps < c.steplim);
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
            // (|phidot| floored at 1e-300: with phidot == 0 exactly -- a = 0 and h = 0 -- the shared quotient below would be 0 x inf = NaN, v_min would
            // ignore it and the TIME cap would be dropped with it, where the reference's "step > |max_tstep / tdot|" applies it; floored, the
            // quotient is min(dt 1e-300, dphi |tdot|) / (|tdot| 1e-300) = dt / |tdot|)
            const T apt = kr_abs(pt1), aphi = abs_floor(pphi1);
            // the time cap applies inside maxtstep_rlim only: outside, "dt" is a huge number with max_tstep's LOW word (TraceConsts::tstep_off_hi),
            // so that the choice is one select on the high word instead of two on a 64-bit pair
            const T dt_eff = __builtin_bit_cast(double, ((unsigned long long) (unsigned) ((s.r < c.tstep_rlim_eff) ? c.tstep_on_hi : c.tstep_off_hi) << 32) | c.tstep_lo);
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
        if (__builtin_amdgcn_ballot_w64(pt1 <= 0) != 0) {        // (inside the ergosphere only: one compare and a scalar branch instead of compare, or, select)
            if (pt1 <= 0) s.status |= KR_STATUS_ERGO;
        }
        // (1 - 2r/rho^2) tdot + (2 a r sin^2/rho^2) phidot IS the conserved energy k (= -p_t): analytically it cannot turn negative, and
        // numerically only where its two terms (~ k / Delta) are 1e15 times k.  Away from the horizon, for k > 0, the test is skipped
        // (wave-uniform; a NaN k or r takes the evaluation, whose comparison is then false as in the reference).
        if (__builtin_amdgcn_ballot_w64(energy_flag_needed(s, dr)) != 0) {
            const T two_r_rho = 2 * s.r * aux.inv_rhosq;
            if ((1 - two_r_rho) * pt1 + (two_r_rho * a * aux.sin2theta) * pphi1 < 0) s.status |= KR_STATUS_NEG_ENERGY;
        }
    } else {
    T rhosq, sin2theta, y_rhosq;
    if (k1_with_flips<T, false, LONE>(s, a, rhosq, sin2theta, &y_rhosq)) return !(s.steps < c.steplim);   // r, theta unchanged
    pt1 = s.pt; pr1 = s.pr; ptheta1 = s.ptheta; pphi1 = s.pphi;

    // step-size heuristic (:224-243 / :855-871 / :1136-1151)
    step = div_const(kr_abs(dv(s.r - c.horizon, pr1)), c.precision, c.inv_precision, c.inv_ok);
    {
        const T q_th = kr_abs(dv(s.theta, ptheta1));
        if (step > div_const(q_th, c.precision, c.inv_precision, c.inv_ok)) step = div_const(q_th, c.theta_precision, c.inv_theta_precision, c.inv_ok);
    }
    if (s.r < c.tstep_rlim_eff) {                       // max_tstep > 0 && r < maxtstep_rlim  (TraceConsts)
        const T st = kr_abs(dv(c.max_tstep, pt1));
        if (step > st) step = st;
    }
    {                                                   // max_phistep > 0: otherwise the quotient is inf / NaN and the comparison false
        const T sp = kr_abs(dv(c.phistep_eff, pphi1));
        if (step > sp) step = sp;
    }
    if ((double) step < KR_MIN_STEP) step = T(KR_MIN_STEP);
    if (s.r + pr1 * step > c.rlim) step = kr_abs(dv(c.rlim - s.r, pr1));
    if (!USE_DEST) {
        if (s.theta + ptheta1 * step > c.theta_hi) step = kr_abs(dv(c.theta_hi - s.theta, ptheta1));
    }

    // flags (:264-273 / :874-887); neither ends the ray
    if (pt1 <= 0) s.status |= KR_STATUS_ERGO;
    // (1 - 2r/rho^2) tdot + (2 a r sin^2/rho^2) phidot is the conserved energy k (= -p_t) evaluated from tdot and phidot: its two terms are
    // bounded (energy_guard_set) so that with r - r_horizon > 1e-6 (Delta > 1e-9 for every a < 0.99999), k > 0 and |h| <= 1e6 k the sum cannot come out
    // negative, in the reference or here.  The flag is therefore only evaluated -- with the reference's operations -- by waves in which some ray is that
    // close to the horizon or carries the mark.
    const bool flag_somewhere = __builtin_amdgcn_ballot_w64(energy_flag_needed(s, s.r - c.horizon)) != 0;
    if (sizeof(T) == 4 || (LONE ? __builtin_expect(flag_somewhere, false) : flag_somewhere)) {       // (LONE: the evaluation out of line, the usual path falls through)
        if ((1 - dv_y(2 * s.r, rhosq, y_rhosq)) * pt1 + dv_y(2 * a * s.r * sin2theta, rhosq, y_rhosq) * pphi1 < 0) s.status |= KR_STATUS_NEG_ENERGY;
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
                } else if constexpr (kNear) {
                    const T theta_stage = s.theta + dtheta;
                    within = within && (kr_abs(theta_stage) < T(KR_SMALL_ANGLE_LIMIT));
                    momentum_impl<T, LONE, true>(pt, pr, ptheta, pphi, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r_stage, theta_stage, a);
                } else {
                    eval<T, false, LONE>(pt, pr, ptheta, pphi, s, r_stage, s.theta + dtheta, a);
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
        if constexpr (FAST) {
            if (!stages(std::true_type{})) stages(std::false_type{});
        } else if constexpr (LONE && sizeof(T) == 8) {
            // Strict path on a wave that owns its SIMD (polar-axis rays: the critical path of the pass).  Nothing covers such a wave's branch bubbles,
            // and kr_sincos_f64 costs two compare-dependent branches per stage to pick its small-angle formula.  While the base angle is small on
            // every lane the three stages take that formula unasked, record whether their angles were small too, and are redone with the full
            // routine if not: the same values, one branch instead of six.
            bool done = false;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(kr_abs(s.theta) < T(KR_SMALL_ANGLE_LIMIT))) == 0, 1)) done = stages(std::true_type{});
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!done) != 0, 0)) {
                if (!done) stages(std::false_type{});
            }
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
    reflect_poles<T, LONE>(s.theta, s.phi, s.thetadot_sign);

    if (s.r <= c.horizon) { s.status |= KR_STATUS_HORIZON; return true; }
    if (USE_DEST) {
        if (dest_reached(c, s.r, s.theta, s.phi, theta_prev)) { s.status |= KR_STATUS_DEST; return true; }
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
    int32_t out_steps = steps_on_entry(s);
    if (s.steps > 0) out_steps += s.steps;
    if (s.status & KR_STATUS_STEPLIM) out_steps = -out_steps;
    return out_steps;
}

}  // namespace kr

#include "kr_rk45.hpp"
