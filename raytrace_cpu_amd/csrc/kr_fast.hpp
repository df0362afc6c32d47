// kr_fast.hpp -- the fast-arithmetic evaluation of the geodesic equations (the unflagged rays of the hybrid launch; every ray with
// KR_FLAG_FAST_MATH): separated potentials, one reciprocal per evaluation, Newton-refined v_rcp_f64 / v_rsq_f64, every fused multiply-add written
// out, sin / cos of the Runge-Kutta stage angles by angle addition.  Included by kr_device.hpp (after Lane<T>).
#pragma once

namespace kr {

// ==== fast-arithmetic path (kr_params.flags & KR_FLAG_FAST_MATH, double only) =====================================
// On gfx950 an IEEE fp64 division costs ~67 SIMD-cycles per wave-instruction and an IEEE sqrt ~92, against ~5.4
// for an FMA (scripts/microbench/fp64_peak.hip); the reference's formulation has 5 divisions + 2 square roots per
// derivative evaluation and 6-9 more divisions in the step heuristic, i.e. about half of an RK4 step.  This path
// evaluates the SAME formulas with one reciprocal per derivative evaluation (1/(rho^2 Delta sin^2 theta), from which
// 1/(rho^2 Delta), 1/rho^2 and 1/sin^2 follow by multiplication), reciprocal-multiply in the heuristic, Newton-refined
// v_rcp_f64 / v_rsq_f64 (<= ~1 ulp) and FMA contraction.  Results differ from the strict path by a few ulp per
// operation -- the same order as the libm difference that already separates the strict path from the CPU -- and are
// held to the same parity tolerances (tests/test_gpu_parity.py runs both).
KR_DEV double fast_rcp(double x)
{
    // v_rcp_f64 is good to ~2^-23; with e = 1 - x y one CUBIC step y (1 + e + e^2) = (1/x)(1 - e^3) lands at 2^-69 before its own rounding
    // (<= 1 ulp, tests/test_gpu_primitives.py) in three fused operations -- two Newton steps, the textbook route to the same accuracy, take four
    const double y = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, y, 1.0);
    const double p = __builtin_fma(e, e, e);
    return __builtin_fma(y, p, y);
}

// One Newton step (relative error ~2^-46): for the step-size heuristic only, whose quotients end up under min() / as a step length
// (a step that is 1e-14 longer moves the sample point along the same trajectory; the landing steps hit r_max / theta_max to 1e-16).
KR_DEV double fast_rcp_heur(double x)
{
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
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
// (Horner steps with the coefficient as a scalar-register operand, kr_fma3s: these constants are live for part of a step only)
constexpr double kNearLimit = 0.07;
// largest |d| for which sincos_near() uses the angle addition from theta0 (computed once per step, shared by its stages)
KR_DEV double sincos_near_limit(double theta0)
{
    return __builtin_fmin(kNearLimit, 0.5 * __builtin_fmin(__builtin_fabs(theta0), __builtin_fabs(kPi - theta0)));
}

KR_DEV void sincos_near(double s0, double c0, double d, double& s, double& c)   // valid for |d| <= sincos_near_limit(theta0)
{
    // |d| <= 0.07 (the step heuristic keeps a whole RK4 step within theta / 50 <= 0.063): sin d through d^9 (next term d^11 / 11! <= 5e-21),
    // cos d - 1 through d^8 (next d^10 / 10! <= 8e-19 of a sum of magnitude ~1): one Horner step less on each side than the 1/8 version
    const double d2 = d * d;
    double ps = kr_fma3s(d2, 1.0 / 362880.0, -1.0 / 5040.0);
    ps = kr_fma3s(ps, d2, 1.0 / 120.0);
    ps = kr_fma3s(ps, d2, -1.0 / 6.0);
    const double sd = __builtin_fma(d * d2, ps, d);                 // sin d
    double pc = kr_fma3s(d2, 1.0 / 40320.0, -1.0 / 720.0);
    pc = kr_fma3s(pc, d2, 1.0 / 24.0);
    pc = __builtin_fma(pc, d2, -0.5);
    const double cm = d2 * pc;                                       // cos d - 1
    s = __builtin_fma(c0, sd, __builtin_fma(s0, cm, s0));
    c = __builtin_fma(-s0, sd, __builtin_fma(c0, cm, c0));
}

// k1 with the turning-point logic (see k1_with_flips) on the fast path
KR_DEV bool k1_with_flips_fast(Lane<double>& s, double a, FastAux& aux)
{
    double sn, c;
    kr_sincos_fast_f64(s.theta, sn, c);
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

}  // namespace kr
