/*
 * oracle/kr_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement (plain C, fp64) of the reference's
 * Kerr geodesic hot path, used only as the checker in tests/, __graft_entry__.smoke() and as
 * bench.py's cpu_baseline leg.  Nothing under raytrace_cpu_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  Built with `gcc -O2 -ffp-contract=off` this file is checked BITWISE
 * (tests/test_oracle_vs_ref.py, tests/test_oracle_golden.py) against
 *   (a) oracle/_ref/libkr_ref.so = the reference's own raytracer.cpp/pointsource.cpp/imageplane.cpp
 *       compiled where they lie (oracle/Makefile), when /root/reference is present, and
 *   (b) the fixtures under tests/golden/ that tests/golden/make_golden.py captured from (a).
 * The reference ships no known-answer vectors of its own (its tests compare integrators with each
 * other, src/tests/), so (a)/(b) are the pins.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference tree).
 * Expression trees (association, order of operations, which libm call) are kept exactly as in the
 * reference so that IEEE results are bit-identical; do not "simplify" the arithmetic here.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/kr_trace.h"

typedef kr_ray_f64 ray_t;

/* ---------------------------------------------------------------------------------------------- */
/* src/include/kerr.h:14-20 */
double kro_kerr_horizon(double a) { return 1 + 1 * sqrt((1 - a) * (1 + a)); }

/* src/include/kerr.h:23-32.  A and B are `const float` in the reference, and the last sqrt resolves to
 * the float overload (kerr.h is included after `using namespace std`), so r_isco is a float value. */
double kro_kerr_isco(double a, int sign)
{
    const float A = (float) (1. + pow(1. - a * a, 1. / 3.) * (pow(1. + a, 1. / 3.) + pow(1. - a, 1. / 3.)));
    const float AA = A * A;
    const float B = (float) sqrt(3. * a * a + AA);
    const float inner = (3 - A) * (3 + A + 2 * B);
    const float res = 3 + B - sign * sqrtf(inner);
    return res;
}

/* src/include/kerr.h:35-38 */
double kro_disc_velocity(double r, double a, int sign) { return 1 / (a + sign * pow(r, 3. / 2.)); }

/* src/include/kerr.h:300-335  momentum_from_consts */
static inline void momentum_from_consts(double* pt, double* pr, double* ptheta, double* pphi, double k, double h,
                                        double Q, int rdot_sign, int thetadot_sign, double r, double theta, double a)
{
    const double sin_theta = sin(theta);
    const double cos_theta = cos(theta);
    const double sin2theta = sin_theta * sin_theta;
    const double rhosq = r * r + (a * cos_theta) * (a * cos_theta);
    const double delta = r * r - 2 * r + a * a;
    const double rhosq_delta = rhosq * delta;

    double tdot = (rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h;
    tdot /= rhosq_delta;

    double phidot = 2 * a * r * sin2theta * k + (rhosq - 2 * r) * h;
    phidot /= sin2theta * rhosq_delta;

    double thetadotsq = Q + (k * a * cos_theta + h * cos_theta / sin_theta) * (k * a * cos_theta - h * cos_theta / sin_theta);
    thetadotsq = thetadotsq / (rhosq * rhosq);
    const double thetadot = sqrt(fabs(thetadotsq)) * thetadot_sign;

    double rdotsq = k * tdot - h * phidot - rhosq * thetadot * thetadot;
    rdotsq = rdotsq * delta / rhosq;

    *pt = tdot;
    *pphi = phidot;
    *ptheta = thetadot;
    *pr = sqrt(fabs(rdotsq)) * rdot_sign;
}

/* ---------------------------------------------------------------------------------------------- */
/* stop surfaces, src/raytracer/ray_destination.h */

/* reached(r, theta, phi, prev_theta): FlatDisc :90-94 (via the default overload :52-54),
 * DiscWithISCO :130-142, FlatPlane :184-190 */
static int dest_reached(const kr_params* p, double r, double theta, double phi, double prev_theta)
{
    switch (p->stop_kind) {
        case KR_STOP_FLATDISC: {
            const double theta_lim = p->stop_params[0];
            if (theta_lim > 0) return theta >= theta_lim;
            if (theta_lim < 0) return theta <= -theta_lim;
            return 0;
        }
        case KR_STOP_DISC_ISCO: {
            const double r_isco = p->stop_params[0], r_out = p->stop_params[1], theta_lim = p->stop_params[2];
            if (r < r_isco) return 0;
            if (r_out > 0 && r > r_out) return 0;
            if (theta_lim > 0)
                return (prev_theta < theta_lim && theta >= theta_lim) || (prev_theta > theta_lim && theta <= theta_lim);
            if (theta_lim < 0) {
                const double tl = -theta_lim;
                return (prev_theta > tl && theta <= tl) || (prev_theta < tl && theta >= tl);
            }
            return 0;
        }
        case KR_STOP_FLATPLANE: {
            const double incl = p->stop_params[0], phi0 = p->stop_params[1], z_s = p->stop_params[2];
            const double proj = r * (sin(theta) * sin(incl) * cos(phi - phi0) + cos(theta) * cos(incl));
            return proj <= -z_s;
        }
    }
    return 0;
}

/* step_limit(): base :55-57, FlatDisc :95-101, DiscWithISCO :143-151 */
static double dest_step_limit(const kr_params* p, double r, double theta, double ptheta)
{
    double theta_lim;
    switch (p->stop_kind) {
        case KR_STOP_FLATDISC:
            theta_lim = p->stop_params[0];
            break;
        case KR_STOP_DISC_ISCO: {
            const double r_isco = p->stop_params[0], r_out = p->stop_params[1];
            if (r < r_isco) return DBL_MAX;
            if (r_out > 0 && r > r_out) return DBL_MAX;
            theta_lim = p->stop_params[2];
            break;
        }
        default:
            return DBL_MAX;
    }
    if (theta_lim > 0 && ptheta > 0 && theta < theta_lim) return (theta_lim - theta) / ptheta;
    if (theta_lim < 0 && ptheta < 0 && theta > -theta_lim) return (-theta_lim - theta) / ptheta;
    return DBL_MAX;
}

/* loop condition of the theta-limit overloads, raytracer.cpp:172 / :799 / :1362-1364 */
static inline int theta_cond(double thetalim, double theta)
{
    return (thetalim > 0 && theta < thetalim) || (thetalim < 0 && theta > fabs(thetalim)) || thetalim == 0;
}

/* std::max / std::min semantics (they matter for NaN operands) */
static inline double std_max(double a, double b) { return (a < b) ? b : a; }
static inline double std_min(double a, double b) { return (b < a) ? b : a; }

/* ---------------------------------------------------------------------------------------------- */
/* k1 at the current position with the turning-point logic; shared by all five propagators
 * (raytracer.cpp:177-222, :805-849, :1086-1130, :1370-1398, :1680-1708).
 * `rk45_assoc` selects the RK45 bodies' association of the phidot denominator
 * ((sin2theta*rhosq)*delta, :1375) instead of sin2theta*(rhosq*delta) (:818).
 * Returns 1 when the iteration must `continue` (theta flip). */
typedef struct {
    double t, r, theta, phi, pt, pr, ptheta, pphi;
    int rdot_sign, thetadot_sign, rdot_flips, equatorial_crossings;
    int r_was_positive, theta_was_positive;
    int status;
} state_t;

static inline int k1_with_flips(state_t* s, double k, double h, double Q, double a, int rk45_assoc, double* rhosq_out,
                                double* sin2theta_out)
{
    const double r = s->r, theta = s->theta;
    const double sin_theta = sin(theta);
    const double cos_theta = cos(theta);
    const double sin2theta = sin_theta * sin_theta;
    const double rhosq = r * r + (a * cos_theta) * (a * cos_theta);
    const double delta = r * r - 2 * r + a * a;

    if (rk45_assoc) {
        s->pt = ((rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h) / (rhosq * delta);
        s->pphi = (2 * a * r * sin2theta * k + (rhosq - 2 * r) * h) / (sin2theta * rhosq * delta);
    } else {
        const double rhosq_delta = rhosq * delta;
        s->pt = (rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h;
        s->pt /= rhosq_delta;
        s->pphi = 2 * a * r * sin2theta * k + (rhosq - 2 * r) * h;
        s->pphi /= sin2theta * rhosq_delta;
    }

    double thetadotsq = Q + (k * a * cos_theta + h * cos_theta / sin_theta) * (k * a * cos_theta - h * cos_theta / sin_theta);
    thetadotsq = thetadotsq / (rhosq * rhosq);

    if (thetadotsq < 0 && s->theta_was_positive) {
        s->thetadot_sign *= -1;
        s->theta_was_positive = 0;
        return 1;
    }
    if (thetadotsq >= 0) s->theta_was_positive = 1;

    s->ptheta = sqrt(fabs(thetadotsq)) * s->thetadot_sign;

    double rdotsq = k * s->pt - h * s->pphi - rhosq * s->ptheta * s->ptheta;
    rdotsq = rdotsq * delta / rhosq;

    if (rdotsq <= 0 && s->r_was_positive) {
        s->rdot_sign *= -1;
        s->r_was_positive = 0;
        s->rdot_flips++;
    } else if (rdotsq > 0) {
        s->r_was_positive = 1;
    }
    s->pr = sqrt(fabs(rdotsq)) * s->rdot_sign;

    *rhosq_out = rhosq;
    *sin2theta_out = sin2theta;
    return 0;
}

static inline void load_state(state_t* s, const ray_t* ray)
{
    s->t = ray->t; s->r = ray->r; s->theta = ray->theta; s->phi = ray->phi;
    s->pt = ray->pt; s->pr = ray->pr; s->ptheta = ray->ptheta; s->pphi = ray->pphi;
    s->rdot_sign = ray->rdot_sign; s->thetadot_sign = ray->thetadot_sign;
    s->rdot_flips = ray->rdot_flips; s->equatorial_crossings = ray->equatorial_crossings;
    s->r_was_positive = 0;      /* raytracer.cpp:137-138, :767-768: per-call, not stored in Ray */
    s->theta_was_positive = 1;
    s->status = ray->status;
}

/* epilogue common to every propagator: raytracer.cpp:315-339, :945-969, :1231-1253, :1574-1597, :1872-1893 */
static inline int store_state(ray_t* ray, const state_t* s, int steps, int steplim, double rlim, int theta_overload,
                              double thetalim)
{
    int status = s->status;
    if (steps >= steplim)
        status |= KR_STATUS_STEPLIM;
    else if (s->r >= rlim)
        status |= KR_STATUS_RLIM;
    else if (theta_overload && ((thetalim > 0 && s->theta >= thetalim) || (thetalim < 0 && s->theta <= fabs(thetalim))))
        status |= KR_STATUS_DEST;

    ray->status = status;
    ray->t = s->t; ray->r = s->r; ray->theta = s->theta; ray->phi = s->phi;
    ray->pt = s->pt; ray->pr = s->pr; ray->ptheta = s->ptheta; ray->pphi = s->pphi;
    ray->rdot_sign = s->rdot_sign; ray->thetadot_sign = s->thetadot_sign;
    ray->rdot_flips = s->rdot_flips; ray->equatorial_crossings = s->equatorial_crossings;

    if (steps > 0) ray->steps += steps;
    if (ray->status & KR_STATUS_STEPLIM) ray->steps = -ray->steps;
    return steps;
}

/* polar reflection, raytracer.cpp:282-283 / :914-915 */
static inline void reflect_poles(double* theta, double* phi, int* thetadot_sign)
{
    if (*theta < 0.0) { *theta = -*theta; *thetadot_sign = -*thetadot_sign; *phi += M_PI; }
    if (*theta > M_PI) { *theta = 2.0 * M_PI - *theta; *thetadot_sign = -*thetadot_sign; *phi += M_PI; }
}

/* ---------------------------------------------------------------------------------------------- */
/* Euler  raytracer.cpp:129-340  and  RK4  :755-970 (theta limit) / :1036-1254 (RayDestination) */
static int propagate_fixed(ray_t* ray, const kr_params* p, int steplim)
{
    const int rk4 = (p->integrator == KR_RK4);
    const int use_dest = (p->stop_kind != KR_STOP_THETA);
    const double a = p->spin, horizon = p->horizon, rlim = p->r_max, thetalim = p->theta_max;
    const double precision = p->precision, theta_precision = p->theta_precision;
    const double max_tstep = p->max_tstep, maxtstep_rlim = p->maxtstep_rlim, max_phistep = p->max_phistep;
    const double k = ray->k, h = ray->h, Q = ray->Q;

    state_t s;
    load_state(&s, ray);
    int steps = 0;

    while (s.r < rlim && (use_dest || theta_cond(thetalim, s.theta)) && steps < steplim) {
        ++steps;

        double rhosq, sin2theta;
        if (k1_with_flips(&s, k, h, Q, a, 0, &rhosq, &sin2theta)) continue;
        const double pt1 = s.pt, pr1 = s.pr, ptheta1 = s.ptheta, pphi1 = s.pphi;

        /* step heuristic  :224-243 / :855-871 / :1136-1151 */
        double step = fabs((s.r - horizon) / pr1) / precision;
        if (step > fabs(s.theta / ptheta1) / precision) step = fabs(s.theta / ptheta1) / theta_precision;
        if (max_tstep > 0 && s.r < maxtstep_rlim && step > fabs(max_tstep / pt1)) step = fabs(max_tstep / pt1);
        if (max_phistep > 0 && step > fabs(max_phistep / pphi1)) step = fabs(max_phistep / pphi1);
        if (step < KR_MIN_STEP) step = KR_MIN_STEP;
        if (rlim > 0 && s.r + pr1 * step > rlim) step = fabs((rlim - s.r) / pr1);
        if (!use_dest && thetalim > 0 && s.theta + ptheta1 * step > thetalim) step = fabs((thetalim - s.theta) / ptheta1);

        /* flags  :264-273 / :874-887 */
        if (pt1 <= 0) s.status |= KR_STATUS_ERGO;
        if ((1 - 2 * s.r / rhosq) * pt1 + (2 * a * s.r * sin2theta / rhosq) * pphi1 < 0) s.status |= KR_STATUS_NEG_ENERGY;

        const double theta_prev = s.theta;
        if (!rk4) {
            /* Euler update  :276-280 */
            s.t += pt1 * step;
            s.r += pr1 * step;
            s.theta += ptheta1 * step;
            if ((theta_prev < M_PI_2 && s.theta >= M_PI_2) || (theta_prev > M_PI_2 && s.theta <= M_PI_2)) ++s.equatorial_crossings;
            s.phi += pphi1 * step;
        } else {
            /* k2..k4 with k1's signs  :889-905 */
            double pt2, pr2, ptheta2, pphi2, pt3, pr3, ptheta3, pphi3, pt4, pr4, ptheta4, pphi4;
            momentum_from_consts(&pt2, &pr2, &ptheta2, &pphi2, k, h, Q, s.rdot_sign, s.thetadot_sign,
                                 s.r + (step / 2) * pr1, s.theta + (step / 2) * ptheta1, a);
            momentum_from_consts(&pt3, &pr3, &ptheta3, &pphi3, k, h, Q, s.rdot_sign, s.thetadot_sign,
                                 s.r + (step / 2) * pr2, s.theta + (step / 2) * ptheta2, a);
            momentum_from_consts(&pt4, &pr4, &ptheta4, &pphi4, k, h, Q, s.rdot_sign, s.thetadot_sign,
                                 s.r + step * pr3, s.theta + step * ptheta3, a);
            /* weighted update  :908-912 */
            s.t += (step / 6) * (pt1 + 2 * pt2 + 2 * pt3 + pt4);
            s.r += (step / 6) * (pr1 + 2 * pr2 + 2 * pr3 + pr4);
            s.theta += (step / 6) * (ptheta1 + 2 * ptheta2 + 2 * ptheta3 + ptheta4);
            if ((theta_prev < M_PI_2 && s.theta >= M_PI_2) || (theta_prev > M_PI_2 && s.theta <= M_PI_2)) ++s.equatorial_crossings;
            s.phi += (step / 6) * (pphi1 + 2 * pphi2 + 2 * pphi3 + pphi4);
        }
        reflect_poles(&s.theta, &s.phi, &s.thetadot_sign);

        if (s.r <= horizon) { s.status |= KR_STATUS_HORIZON; break; }
        if (use_dest && dest_reached(p, s.r, s.theta, s.phi, theta_prev)) { s.status |= KR_STATUS_DEST; break; }
    }
    return store_state(ray, &s, steps, steplim, rlim, !use_dest, thetalim);
}

/* ---------------------------------------------------------------------------------------------- */
/* RK45 / DOPRI5  raytracer.cpp:1260-1598 (theta limit) / :1600-1894 (RayDestination) */
static int propagate_rk45(ray_t* ray, const kr_params* p, int steplim, int64_t* attempts, int64_t* rejects)
{
    /* tableau  :1316-1330 */
    static const double a21 = 1.0 / 5;
    static const double a31 = 3.0 / 40, a32 = 9.0 / 40;
    static const double a41 = 44.0 / 45, a42 = -56.0 / 15, a43 = 32.0 / 9;
    static const double a51 = 19372.0 / 6561, a52 = -25360.0 / 2187, a53 = 64448.0 / 6561, a54 = -212.0 / 729;
    static const double a61 = 9017.0 / 3168, a62 = -355.0 / 33, a63 = 46732.0 / 5247, a64 = 49.0 / 176, a65 = -5103.0 / 18656;
    static const double b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84;
    static const double e1 = 71.0 / 57600, e3 = -71.0 / 16695, e4 = 71.0 / 1920, e5 = -17253.0 / 339200, e6 = 22.0 / 525, e7 = -1.0 / 40;
    static const double safety = 0.9, fac_max = 5.0, fac_min = 0.1;

    const int use_dest = (p->stop_kind != KR_STOP_THETA);
    const double a = p->spin, horizon = p->horizon, rlim = p->r_max, thetalim = p->theta_max;
    const double precision = p->precision, theta_precision = p->theta_precision;
    const double max_tstep = p->max_tstep, maxtstep_rlim = p->maxtstep_rlim, max_phistep = p->max_phistep;
    const double tol = p->rk45_tol;
    const double k = ray->k, h = ray->h, Q = ray->Q;

    state_t s;
    load_state(&s, ray);
    int steps = 0;
    double theta_eq_prev = s.theta;

    /* seed the running step  :1341-1359 (no flip logic, no boundary clips) */
    {
        const double r = s.r, theta = s.theta;
        const double sin_theta = sin(theta), cos_theta = cos(theta), sin2theta = sin_theta * sin_theta;
        const double rhosq = r * r + (a * cos_theta) * (a * cos_theta);
        const double delta = r * r - 2 * r + a * a;
        s.pt = ((rhosq * (r * r + a * a) + 2 * a * a * r * sin2theta) * k - 2 * a * r * h) / (rhosq * delta);
        s.pphi = (2 * a * r * sin2theta * k + (rhosq - 2 * r) * h) / (sin2theta * rhosq * delta);
        const double thetadotsq = (Q + (k * a * cos_theta + h * cos_theta / sin_theta) * (k * a * cos_theta - h * cos_theta / sin_theta)) / (rhosq * rhosq);
        s.ptheta = sqrt(fabs(thetadotsq)) * s.thetadot_sign;
        const double rdotsq = (k * s.pt - h * s.pphi - rhosq * s.ptheta * s.ptheta) * delta / rhosq;
        s.pr = sqrt(fabs(rdotsq)) * s.rdot_sign;
    }
    double step = fabs((s.r - horizon) / s.pr) / precision;
    if (fabs(s.ptheta) > 0 && step > fabs(s.theta / s.ptheta) / theta_precision) step = fabs(s.theta / s.ptheta) / theta_precision;
    if (max_tstep > 0 && s.r < maxtstep_rlim && step > fabs(max_tstep / s.pt)) step = fabs(max_tstep / s.pt);
    if (max_phistep > 0 && step > fabs(max_phistep / s.pphi)) step = fabs(max_phistep / s.pphi);
    if (step < KR_MIN_STEP) step = KR_MIN_STEP;

    while (s.r < rlim && (use_dest || theta_cond(thetalim, s.theta)) && steps < steplim) {
        ++steps;

        double rhosq_k1, sin2theta_k1;
        if (k1_with_flips(&s, k, h, Q, a, 1, &rhosq_k1, &sin2theta_k1)) continue;
        const double pt1 = s.pt, pr1 = s.pr, ptheta1 = s.ptheta, pphi1 = s.pphi;
        const double r = s.r, theta = s.theta, phi = s.phi, t = s.t;

        /* flags  :1403-1410 (same expressions as k1's rhosq / sin2theta) */
        if (pt1 <= 0) s.status |= KR_STATUS_ERGO;
        if ((1 - 2 * r / rhosq_k1) * pt1 + (2 * a * r * sin2theta_k1 / rhosq_k1) * pphi1 < 0) s.status |= KR_STATUS_NEG_ENERGY;

        /* outer cap  :1421-1434 */
        {
            double step_max = fabs((r - horizon) / pr1) / precision;
            if (max_phistep > 0) {
                const double step_phi = fabs(max_phistep / pphi1);
                if (step_phi < step_max) step_max = step_phi;
            }
            if (max_tstep > 0 && r < maxtstep_rlim) {
                const double step_t = fabs(max_tstep / pt1);
                if (step_t < step_max) step_max = step_t;
            }
            if (step > step_max) step = step_max;
        }

        const double theta_prev = theta;
        int accepted = 0;
        while (!accepted) {
            double h_try = step;
            int clamped = 0;
            if (!use_dest) {
                /* :1449-1453 */
                if (thetalim > 0 && theta + ptheta1 * h_try > thetalim) {
                    const double h_th = fabs((thetalim - theta) / ptheta1);
                    if (h_th < h_try) { h_try = h_th; clamped = 1; }
                }
            } else {
                /* :1747-1755 */
                if (rlim > 0 && r + pr1 * h_try > rlim) { h_try = fabs((rlim - r) / pr1); clamped = 1; }
                const double h_dest = dest_step_limit(p, r, theta, ptheta1);
                if (h_dest < h_try) { h_try = h_dest; clamped = 1; }
            }
            if (attempts) ++*attempts;

            double pt2, pr2, ptheta2, pphi2, pt3, pr3, ptheta3, pphi3, pt4, pr4, ptheta4, pphi4;
            double pt5, pr5, ptheta5, pphi5, pt6, pr6, ptheta6, pphi6, pt7, pr7, ptheta7, pphi7;
            momentum_from_consts(&pt2, &pr2, &ptheta2, &pphi2, k, h, Q, s.rdot_sign, s.thetadot_sign,
                                 r + h_try * a21 * pr1, theta + h_try * a21 * ptheta1, a);
            momentum_from_consts(&pt3, &pr3, &ptheta3, &pphi3, k, h, Q, s.rdot_sign, s.thetadot_sign,
                                 r + h_try * (a31 * pr1 + a32 * pr2), theta + h_try * (a31 * ptheta1 + a32 * ptheta2), a);
            momentum_from_consts(&pt4, &pr4, &ptheta4, &pphi4, k, h, Q, s.rdot_sign, s.thetadot_sign,
                                 r + h_try * (a41 * pr1 + a42 * pr2 + a43 * pr3),
                                 theta + h_try * (a41 * ptheta1 + a42 * ptheta2 + a43 * ptheta3), a);
            momentum_from_consts(&pt5, &pr5, &ptheta5, &pphi5, k, h, Q, s.rdot_sign, s.thetadot_sign,
                                 r + h_try * (a51 * pr1 + a52 * pr2 + a53 * pr3 + a54 * pr4),
                                 theta + h_try * (a51 * ptheta1 + a52 * ptheta2 + a53 * ptheta3 + a54 * ptheta4), a);
            momentum_from_consts(&pt6, &pr6, &ptheta6, &pphi6, k, h, Q, s.rdot_sign, s.thetadot_sign,
                                 r + h_try * (a61 * pr1 + a62 * pr2 + a63 * pr3 + a64 * pr4 + a65 * pr5),
                                 theta + h_try * (a61 * ptheta1 + a62 * ptheta2 + a63 * ptheta3 + a64 * ptheta4 + a65 * ptheta5), a);

            /* 5th-order solution  :1493-1499 */
            double r_new = r + h_try * (b1 * pr1 + b3 * pr3 + b4 * pr4 + b5 * pr5 + b6 * pr6);
            double theta_new = theta + h_try * (b1 * ptheta1 + b3 * ptheta3 + b4 * ptheta4 + b5 * ptheta5 + b6 * ptheta6);
            double t_new = t + h_try * (b1 * pt1 + b3 * pt3 + b4 * pt4 + b5 * pt5 + b6 * pt6);
            double phi_new = phi + h_try * (b1 * pphi1 + b3 * pphi3 + b4 * pphi4 + b5 * pphi5 + b6 * pphi6);
            reflect_poles(&theta_new, &phi_new, &s.thetadot_sign);   /* mutates the sign even if rejected, :1498-1499 */

            momentum_from_consts(&pt7, &pr7, &ptheta7, &pphi7, k, h, Q, s.rdot_sign, s.thetadot_sign, r_new, theta_new, a);

            /* error norm + controller  :1508-1519 */
            const double err_r = h_try * (e1 * pr1 + e3 * pr3 + e4 * pr4 + e5 * pr5 + e6 * pr6 + e7 * pr7);
            const double err_theta = h_try * (e1 * ptheta1 + e3 * ptheta3 + e4 * ptheta4 + e5 * ptheta5 + e6 * ptheta6 + e7 * ptheta7);
            const double sc_r = tol * (1.0 + std_max(fabs(r), fabs(r_new)));
            const double sc_theta = tol * (1.0 + std_max(fabs(theta), fabs(theta_new)));
            const double err_norm = sqrt(0.5 * ((err_r / sc_r) * (err_r / sc_r) + (err_theta / sc_theta) * (err_theta / sc_theta)));

            double fac = safety * pow(1.0 / std_max(err_norm, 1e-10), 0.2);
            fac = std_max(fac_min, std_min(fac_max, fac));
            const double step_new = h_try * fac;

            if (err_norm <= 1.0) {
                s.t = t_new; s.r = r_new; s.theta = theta_new; s.phi = phi_new;
                s.pt = pt7; s.pr = pr7; s.ptheta = ptheta7; s.pphi = pphi7;
                if (!clamped) step = std_max(step_new, KR_MIN_STEP);
                accepted = 1;
            } else {
                if (rejects) ++*rejects;
                step = std_max(step_new, KR_MIN_STEP);
                if (step <= KR_MIN_STEP) {
                    s.t = t_new; s.r = r_new; s.theta = theta_new; s.phi = phi_new;
                    s.pt = pt7; s.pr = pr7; s.ptheta = ptheta7; s.pphi = pphi7;
                    accepted = 1;
                } else if (err_norm != err_norm) {
                    /* The reference spins forever here (NaN never satisfies either exit).  The oracle
                     * mirrors the product's documented deviation: end the ray, flag KR_STATUS_NAN. */
                    s.status |= KR_STATUS_NAN;
                    return store_state(ray, &s, steps, steplim, rlim, !use_dest, thetalim);
                }
            }
        }
        /* :1542-1544 */
        if ((theta_eq_prev < M_PI_2 && s.theta >= M_PI_2) || (theta_eq_prev > M_PI_2 && s.theta <= M_PI_2)) ++s.equatorial_crossings;
        theta_eq_prev = s.theta;

        if (s.r <= horizon) { s.status |= KR_STATUS_HORIZON; break; }
        if (use_dest && dest_reached(p, s.r, s.theta, s.phi, theta_prev)) { s.status |= KR_STATUS_DEST; break; }
    }
    return store_state(ray, &s, steps, steplim, rlim, !use_dest, thetalim);
}

/* ---------------------------------------------------------------------------------------------- */
/* Raytracer<T>::run_raytrace, both overloads: raytracer.cpp:63-127 and :972-1034.
 * nthreads <= 0: OpenMP default.  Returns 0, or KR_EINVAL. */
int kro_trace_f64(const kr_params* p, kr_ray_f64* rays, int64_t n, int nthreads, kr_stats* stats)
{
    if (!p || (!rays && n > 0)) return KR_EINVAL;
    if (p->integrator < KR_EULER || p->integrator > KR_RK45) return KR_EINVAL;
    if (p->stop_kind < KR_STOP_THETA || p->stop_kind > KR_STOP_FLATPLANE) return KR_EINVAL;
    if (p->stop_kind != KR_STOP_THETA && p->integrator == KR_EULER) return KR_EINVAL;   /* assert, :983 */

    const int steplim = (p->steplim > 0) ? p->steplim : (p->integrator == KR_RK45) ? KR_RK45_STEPLIM : KR_STEPLIM;   /* :80 */
    int64_t traced = 0, steps_total = 0, attempts = 0, rejects = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : traced, steps_total, attempts, rejects)
    for (int64_t i = 0; i < n; i++) {
        if (rays[i].steps < 0) continue;           /* :116 */
        if (rays[i].steps >= steplim) continue;    /* :117 */
        int64_t att = 0, rej = 0;
        int st;
        if (p->integrator == KR_RK45)
            st = propagate_rk45(&rays[i], p, steplim, &att, &rej);
        else
            st = propagate_fixed(&rays[i], p, steplim);
        traced += 1;
        steps_total += st;
        attempts += att;
        rejects += rej;
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->rays_total = n;
        stats->rays_traced = traced;
        stats->steps_total = steps_total;
        stats->rk45_attempts = attempts;
        stats->rk45_rejects = rejects;
    }
    return KR_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* Kerr metric in the (e2nu, e2psi, omega) form used by redshift_start / ray_redshift / calculate_constants
 * (raytracer.cpp:370-388, :491-509, :564-582, :632-639) */
typedef struct { double rhosq, delta, sigmasq, e2nu, e2psi, omega, g[16]; } metric_t;

static inline void kerr_metric_at(metric_t* m, double r, double theta, double a)
{
    m->rhosq = r * r + (a * cos(theta)) * (a * cos(theta));
    m->delta = r * r - 2 * r + a * a;
    m->sigmasq = (r * r + a * a) * (r * r + a * a) - a * a * m->delta * sin(theta) * sin(theta);
    m->e2nu = m->rhosq * m->delta / m->sigmasq;
    m->e2psi = m->sigmasq * sin(theta) * sin(theta) / m->rhosq;
    m->omega = 2 * a * r / m->sigmasq;
    for (int i = 0; i < 16; i++) m->g[i] = 0;
    m->g[0 * 4 + 0] = m->e2nu - m->omega * m->omega * m->e2psi;
    m->g[0 * 4 + 3] = m->omega * m->e2psi;
    m->g[3 * 4 + 0] = m->g[0 * 4 + 3];
    m->g[1 * 4 + 1] = -m->rhosq / m->delta;
    m->g[2 * 4 + 2] = -m->rhosq;
    m->g[3 * 4 + 3] = -m->e2psi;
}

static inline double energy_dot(const double* g, const double* et, const double* p)
{
    double e = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) e += g[i * 4 + j] * et[i] * p[j];
    return e;
}

/* Raytracer<T>::redshift_start  raytracer.cpp:342-417.  V is a by-value parameter that the loop body
 * overwrites when it is -1, so the orbital velocity of the FIRST ray is reused for all later rays. */
void kro_redshift_start_f64(double spin, double V, int reverse, int projradius, kr_ray_f64* rays, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        ray_t* ray = &rays[i];
        double p[4];
        const double a = reverse ? -1 * spin : spin;
        metric_t m;
        kerr_metric_at(&m, ray->r, ray->theta, a);

        if (V == -1 && projradius)
            V = 1 / (a + ray->r * sin(ray->theta) * sqrt(ray->r * sin(ray->theta)));
        else if (V == -1)
            V = 1 / (a + ray->r * sqrt(ray->r));

        const double et[4] = {(1 / sqrt(m.e2nu)) / sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu), 0, 0,
                              (1 / sqrt(m.e2nu)) * V / sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu)};

        momentum_from_consts(&p[0], &p[1], &p[2], &p[3], ray->k, ray->h, ray->Q, ray->rdot_sign, ray->thetadot_sign,
                             ray->r, ray->theta, spin);
        if (reverse) { p[1] *= -1; p[2] *= -1; p[3] *= -1; }

        ray->emit = 0;
        for (int ii = 0; ii < 4; ii++)
            for (int jj = 0; jj < 4; jj++) ray->emit += m.g[ii * 4 + jj] * et[ii] * p[jj];
    }
}

/* Raytracer<T>::ray_redshift(V, ...)  raytracer.cpp:480-553 */
static double ray_redshift_V(double spin, double V, int reverse, int projradius, int motion, const ray_t* ray)
{
    const double r = ray->r, theta = ray->theta;
    double p[4];
    const double a = reverse ? -1 * spin : spin;
    metric_t m;
    kerr_metric_at(&m, r, theta, a);

    double et[4] = {0, 0, 0, 0};
    if (motion == 0) {
        if (V == -1 && projradius)
            V = 1 / (a + r * sin(theta) * sqrt(r * sin(theta)));
        else if (V == -1)
            V = 1 / (a + r * sqrt(r));
        et[0] = (1 / sqrt(m.e2nu)) / sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu);
        et[3] = (1 / sqrt(m.e2nu)) * V / sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu);
    } else if (motion == 1) {
        if (V < 0) V = fabs(V) * (r * r - 2 * r + spin + spin) / (r * r + spin * spin);   /* sic: spin+spin, :531 */
        et[0] = 1. / sqrt(m.g[0 * 4 + 0] + m.g[1 * 4 + 1] * V * V);
        et[1] = V * et[0];
    }

    momentum_from_consts(&p[0], &p[1], &p[2], &p[3], ray->k, ray->h, ray->Q, ray->rdot_sign, ray->thetadot_sign, r, theta, spin);
    if (reverse) { p[1] *= -1; p[2] *= -1; p[3] *= -1; }

    const double recv = energy_dot(m.g, et, p);
    return reverse ? recv / ray->emit : ray->emit / recv;
}

/* Raytracer<T>::redshift(V, reverse, projradius, motion)  raytracer.cpp:420-447 */
void kro_redshift_f64(double spin, double V, int reverse, int projradius, int motion, kr_ray_f64* rays, int64_t n)
{
    for (int64_t i = 0; i < n; i++) rays[i].redshift = ray_redshift_V(spin, V, reverse, projradius, motion, &rays[i]);
}

/* Raytracer<T>::redshift(RayDestination*, reverse)  raytracer.cpp:450-477, :556-600 with the default
 * RayDestination::four_velocity (ray_destination.h:59-78; velocity() = -1 in every concrete class) */
void kro_redshift_dest_f64(double spin, int reverse, kr_ray_f64* rays, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        ray_t* ray = &rays[i];
        const double r = ray->r, theta = ray->theta;
        double et[4], p[4];
        {
            double V = -1;
            const double rhosq = r * r + (spin * cos(theta)) * (spin * cos(theta));
            const double delta = r * r - 2 * r + spin * spin;
            const double sigmasq = (r * r + spin * spin) * (r * r + spin * spin) - spin * spin * delta * sin(theta) * sin(theta);
            const double e2nu = rhosq * delta / sigmasq;
            const double e2psi = sigmasq * sin(theta) * sin(theta) / rhosq;
            const double omega = 2 * spin * r / sigmasq;
            if (V == -1) V = 1 / (spin + r * sqrt(r));
            const double gamma_factor = 1 / sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);
            et[0] = gamma_factor / sqrt(e2nu);
            et[1] = 0;
            et[2] = 0;
            et[3] = gamma_factor * V / sqrt(e2nu);
        }
        metric_t m;
        kerr_metric_at(&m, r, theta, spin);
        momentum_from_consts(&p[0], &p[1], &p[2], &p[3], ray->k, ray->h, ray->Q, ray->rdot_sign, ray->thetadot_sign, r, theta, spin);
        if (reverse) { p[1] *= -1; p[2] *= -1; p[3] *= -1; }
        const double recv = energy_dot(m.g, et, p);
        ray->redshift = reverse ? recv / ray->emit : ray->emit / recv;
    }
}

/* Raytracer<T>::range_phi  raytracer.cpp:603-622 */
void kro_range_phi_f64(double lo, double hi, kr_ray_f64* rays, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        if (fabs(rays[i].phi) > 1000 || rays[i].phi != rays[i].phi || !(rays[i].steps > 0)) continue;
        while (rays[i].phi >= hi) rays[i].phi -= 2 * M_PI;
        while (rays[i].phi < lo) rays[i].phi += 2 * M_PI;
    }
}

/* Raytracer<T>::calculate_momentum  raytracer.cpp:704-753 (same expression tree as momentum_from_consts) */
void kro_calculate_momentum_f64(double spin, kr_ray_f64* rays, int64_t n)
{
    for (int64_t i = 0; i < n; i++)
        momentum_from_consts(&rays[i].pt, &rays[i].pr, &rays[i].ptheta, &rays[i].pphi, rays[i].k, rays[i].h, rays[i].Q,
                             rays[i].rdot_sign, rays[i].thetadot_sign, rays[i].r, rays[i].theta, spin);
}

/* ---------------------------------------------------------------------------------------------- */
/* Raytracer<T>::calculate_constants  raytracer.cpp:625-676 */
static void calculate_constants(ray_t* ray, double spin, double alpha, double beta, double V, double E)
{
    const double r = ray->r, th = ray->theta;
    const double rhosq = r * r + (spin * cos(th)) * (spin * cos(th));
    const double delta = r * r - 2 * r + spin * spin;
    const double sigmasq = (r * r + spin * spin) * (r * r + spin * spin) - spin * spin * delta * sin(th) * sin(th);

    const double e2nu = rhosq * delta / sigmasq;
    const double e2psi = sigmasq * sin(th) * sin(th) / rhosq;
    const double omega = 2 * spin * r / sigmasq;

    const double et0 = (1 / sqrt(e2nu)) / sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);
    const double et3 = (1 / sqrt(e2nu)) * V / sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);
    const double e10 = (V - omega) * sqrt(e2psi / e2nu) / sqrt(e2nu - (V - omega) * (V - omega) * e2psi);
    const double e13 = (1 / sqrt(e2nu * e2psi)) * (e2nu + V * omega * e2psi - omega * omega * e2psi) / sqrt(e2nu - (V - omega) * (V - omega) * e2psi);
    const double e22 = -1 / sqrt(rhosq);
    const double e31 = sqrt(delta / rhosq);

    const double rdotprime[4] = {E, E * sin(alpha) * cos(beta), E * sin(alpha) * sin(beta), E * cos(alpha)};

    const double tdot = rdotprime[0] * et0 + rdotprime[1] * e10;
    const double phidot = rdotprime[0] * et3 + rdotprime[1] * e13;
    const double rdot = rdotprime[3] * e31;
    const double thetadot = rdotprime[2] * e22;

    ray->k = (1 - 2 * r / rhosq) * tdot + (2 * spin * r * sin(th) * sin(th) / rhosq) * phidot;

    ray->h = phidot * ((r * r + spin * spin) * (r * r + spin * spin * cos(th) * cos(th) - 2 * r) * sin(th) * sin(th) +
                       2 * spin * spin * r * sin(th) * sin(th) * sin(th) * sin(th));
    ray->h = ray->h - 2 * spin * r * ray->k * sin(th) * sin(th);
    ray->h = ray->h / (r * r + spin * spin * cos(th) * cos(th) - 2 * r);

    ray->Q = rhosq * rhosq * thetadot * thetadot -
             (spin * ray->k * cos(th) + ray->h / tan(th)) * (spin * ray->k * cos(th) - ray->h / tan(th));

    ray->rdot_sign = (rdot >= 0) ? 1 : -1;
    ray->thetadot_sign = (thetadot > 0) ? 1 : -1;
    ray->rdot_flips = 0;
    ray->equatorial_crossings = 0;
}

/* ray-count arithmetic of the PointSource ctor: an int-truncated PRODUCT of doubles, pointsource.cpp:12,16-17 */
int64_t kro_pointsource_count(const kr_pointsource* s, int32_t* n_cosalpha, int32_t* n_beta)
{
    const int nrays = (int) ((((s->cosalphamax - s->cosalpha0) / s->dcosalpha) + 1) * (((s->betamax - s->beta0) / s->dbeta) + 1));
    if (n_cosalpha) *n_cosalpha = (int) (((s->cosalphamax - s->cosalpha0) / s->dcosalpha) + 1);
    if (n_beta) *n_beta = (int) (((s->betamax - s->beta0) / s->dbeta) + 1);
    return nrays;
}

/* Raytracer ctor (steps=-1,status=0; raytracer.cpp:45-49) + PointSource::init_pointsource (pointsource.cpp:30-64).
 * Fields the reference leaves indeterminate are zeroed here. */
int kro_pointsource_init_f64(const kr_pointsource* s, kr_ray_f64* rays, int64_t n)
{
    int32_t n_cosalpha, n_beta;
    const int64_t need = kro_pointsource_count(s, &n_cosalpha, &n_beta);
    if (n < need) return KR_EINVAL;
    memset(rays, 0, (size_t) n * sizeof(ray_t));
    for (int64_t i = 0; i < n; i++) { rays[i].steps = -1; rays[i].status = 0; }

    for (int i = 0; i < n_cosalpha; i++)
        for (int j = 0; j < n_beta; j++) {
            const int ix = i * n_beta + j;
            const double cosalpha = s->cosalpha0 + i * s->dcosalpha;
            const double beta = s->beta0 + j * s->dbeta;
            if (cosalpha >= s->cosalphamax || beta >= s->betamax) { rays[ix].steps = -1; continue; }
            const double alpha = acos(cosalpha);
            ray_t* ray = &rays[ix];
            ray->alpha = cosalpha;     /* sic: cos(alpha), pointsource.cpp:48 */
            ray->beta = beta;
            ray->t = s->pos[0]; ray->r = s->pos[1]; ray->theta = s->pos[2]; ray->phi = s->pos[3];
            ray->pt = 0; ray->pr = 0; ray->ptheta = 0; ray->pphi = 0;
            ray->steps = 0;
            calculate_constants(ray, s->spin, alpha, beta, s->V, s->E);
        }
    return KR_OK;
}

/* imageplane.cpp:12-14 */
int64_t kro_imageplane_count(const kr_imageplane* s, int32_t* nx, int32_t* ny)
{
    const int nrays = (int) ((((s->xmax - s->x0) / s->dx) + 1) * (((s->ymax - s->y0) / s->dy) + 1));
    if (nx) *nx = (int) (((s->xmax - s->x0) / s->dx) + 1);
    if (ny) *ny = (int) (((s->ymax - s->y0) / s->dy) + 1);
    return nrays;
}

/* ImagePlane ctor + init_image_plane (imageplane.cpp:11-121) incl. calculate_constants_from_p
 * (raytracer.cpp:678-701), whose k/h/Q are then overwritten (imageplane.cpp:100-113). */
int kro_imageplane_init_f64(const kr_imageplane* s, kr_ray_f64* rays, int64_t n)
{
    int32_t Nx, Ny;
    const int64_t need = kro_imageplane_count(s, &Nx, &Ny);
    if (n < need) return KR_EINVAL;
    memset(rays, 0, (size_t) n * sizeof(ray_t));
    for (int64_t i = 0; i < n; i++) { rays[i].steps = -1; rays[i].status = 0; }

    const double a = -1 * s->spin;                    /* imageplane.cpp:12 */
    const double D = s->dist, incl = s->inc_deg * M_PI / 180, phi0 = s->phi0;
    const double x0 = s->x0, y0 = s->y0, dy = s->dy;

    for (int i = 0; i < Nx; i++) {
        const double x = x0 + i * dy;                 /* sic: dy, imageplane.cpp:43 */
        for (int j = 0; j < Ny; j++) {
            const int ix = i * Ny + j;
            const double y = y0 + j * dy;

            const double t = 0;
            const double r = sqrt(D * D + x * x + y * y);
            const double theta = acos((D * cos(incl) + y * sin(incl)) / r);
            const double phi = phi0 + atan2(x, D * sin(incl) - y * cos(incl));

            const double pr = D / r;
            const double ptheta = sin(acos(D / r)) / r;
            const double pphi = x * sin(incl) / (x * x + (D * sin(incl) - y * cos(incl)) * (D * sin(incl) - y * cos(incl)));

            const double rhosq = r * r + (a * cos(theta)) * (a * cos(theta));
            const double delta = r * r - 2 * r + a * a;
            const double sigmasq = (r * r + a * a) * (r * r + a * a) - a * a * delta * sin(theta) * sin(theta);
            const double e2nu = rhosq * delta / sigmasq;
            const double e2psi = sigmasq * sin(theta) * sin(theta) / rhosq;
            const double omega = 2 * a * r / sigmasq;

            const double g00 = e2nu - omega * omega * e2psi;
            const double g03 = omega * e2psi;
            const double g11 = -rhosq / delta;
            const double g22 = -rhosq;
            const double g33 = -e2psi;

            const double A = g00;
            const double B = 2 * g03 * pphi;
            const double C = g11 * pr * pr + g22 * ptheta * ptheta + g33 * pphi * pphi;
            double pt = (-B + sqrt(B * B - 4 * A * C)) / (2 * A);
            if (pt < 0) pt = (-B - sqrt(B * B - 4 * A * C)) / (2 * A);

            ray_t* ray = &rays[ix];
            ray->t = t; ray->r = r; ray->theta = theta; ray->phi = phi;
            ray->pt = pt; ray->pr = pr; ray->ptheta = ptheta; ray->pphi = pphi;

            /* calculate_constants_from_p(ix, pt, pr, ptheta, pphi) is evaluated by the reference but every
             * output (k, h, Q) is overwritten just below; it has no side effects, so only the overwrite stays. */
            ray->rdot_sign = -1;
            ray->thetadot_sign = 1;
            ray->k = 1;

            const double b = sqrt(x * x + y * y);
            double beta = asin(y / b);
            if (x < 0) beta = M_PI - beta;

            const double h = -1. * b * sin(incl) * cos(beta);
            const double ltheta = b * sin(beta);
            const double Q = (ltheta * ltheta) - (a * cos(theta)) * (a * cos(theta)) + ((h / tan(theta))) * ((h / tan(theta)));

            ray->h = h;
            ray->Q = Q;
            ray->thetadot_sign = (ltheta >= 0) ? 1 : -1;
            ray->steps = 0;
            ray->alpha = x;
            ray->beta = y;
            /* rdot_flips / equatorial_crossings are left indeterminate by the reference; zeroed here */
        }
    }
    return KR_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* emissivity reducer, src/emissivity/emissivity.cpp:96-126 (raw accumulators; the divisions of :128-134
 * are left to the caller) */
void kro_reduce_emissivity_f64(const kr_emis_bins* b, const kr_ray_f64* rays, int64_t n, int64_t* count, double* flux,
                               double* emis, double* sum_redshift, double* sum_time, int64_t* disc_count)
{
    for (int ir = 0; ir < b->nr; ir++) { count[ir] = 0; flux[ir] = 0; emis[ir] = 0; sum_redshift[ir] = 0; sum_time[ir] = 0; }
    int64_t dc = 0;
    for (int64_t i = 0; i < n; i++) {
        const ray_t* ray = &rays[i];
        if (ray->steps > 0) {
            const double z = ray->r * cos(ray->theta);     /* cartesian(), kerr.h:55 */
            if (z < 1E-2 && ray->redshift > 0 && ray->r >= b->r_isco) {
                const int ir = b->logbin ? (int) (log(ray->r / b->r_min) / log(b->dr)) : (int) ((ray->r - b->r_min) / b->dr);
                if (ir >= 0 && ir < b->nr) {
                    ++count[ir];
                    flux[ir] += 1 / (b->num_primary_rays * pow(ray->redshift, 1));
                    emis[ir] += 1 / pow(ray->redshift, b->gamma);
                    sum_redshift[ir] += ray->redshift;
                    sum_time[ir] += ray->t;
                }
                ++dc;
            }
        }
    }
    if (disc_count) *disc_count = dc;
}

/* imageplane_disc_image.cpp:20-28 */
static double powerlaw3(double r, double q1, double rb1, double q2, double rb2, double q3)
{
    if (r < rb1) return pow(r, -1 * q1);
    else if (r < rb2) return pow(rb1, q2 - q1) * pow(r, -1 * q2);
    else return pow(rb1, q2 - q1) * pow(rb2, q3 - q2) * pow(r, -1 * q3);
}

/* image reducer, src/imageplane/imageplane_disc_image.cpp:122-161 (raw sums; the divisions of :165-174
 * are left to the caller).  Planes are [ix*img_ny + iy] like the reference's Array2D. */
void kro_reduce_image_f64(const kr_image_bins* b, const kr_ray_f64* rays, int64_t n, int32_t* nrays, double* flux,
                          double* rr, double* phi, double* enshift, double* time, double* emis, int64_t* disc_count)
{
    const int64_t npix = (int64_t) b->img_nx * b->img_ny;
    for (int64_t i = 0; i < npix; i++) { nrays[i] = 0; flux[i] = 0; rr[i] = 0; phi[i] = 0; enshift[i] = 0; time[i] = 0; emis[i] = 0; }
    int64_t dc = 0;
    for (int64_t i = 0; i < n; i++) {
        const ray_t* ray = &rays[i];
        if (ray->steps > 0) {
            const double z = ray->r * cos(ray->theta);
            if (z < 1E-2 && ray->r >= b->r_isco && ray->r < b->r_disc && ray->redshift > 0) {
                const double x = ray->alpha, y = ray->beta;
                int ix = (int) ((x - b->x0) / b->img_dx);
                int iy = (int) ((y - b->y0) / b->img_dy);
                if (b->flip_image) iy = b->img_ny - iy - 1;
                if (ix >= 0 && ix < b->img_nx && iy >= 0 && iy < b->img_ny) {
                    const int64_t px = (int64_t) ix * b->img_ny + iy;
                    ++nrays[px];
                    const double e = powerlaw3(ray->r, b->q1, b->rb1, b->q2, b->rb2, b->q3);
                    flux[px] += e / pow(ray->redshift, 3);
                    rr[px] += ray->r;
                    phi[px] += ray->phi;
                    enshift[px] += 1. / ray->redshift;
                    time[px] += ray->t;
                    emis[px] += e;
                    ++dc;
                }
            }
        }
    }
    if (disc_count) *disc_count = dc;
}

/* returning-radiation classification: the loop body of src/return_radiation/disc_source_photonfrac_r.cpp:97-126
 * (stale app: `ray_cosalpha(ray)` / `ray_beta(ray)` / map_results() are read here from rays[].alpha (= cos alpha,
 * pointsource.cpp:48), rays[].beta and the record itself).  out = {ray_count, return, escape, lost}. */
void kro_reduce_return_f64(const kr_return_bins* b, const kr_ray_f64* rays, int64_t n, double out[4])
{
    double ray_count = 0, return_count = 0, escape_count = 0, lost_count = 0;
    for (int64_t i = 0; i < n; i++) {
        const ray_t* ray = &rays[i];
        if (ray->steps > 0) {
            const double alpha = acos(ray->alpha);
            const double beta = ray->beta;
            double ray_weight = b->plane_iso ? fabs(sin(alpha) * sin(beta)) : 1;
            if (b->limb) ray_weight *= 1 + 2.06 * (fabs(sin(alpha) * sin(beta)));
            ray_count += b->weight_norm ? ray_weight : 1;
            if (ray->theta >= M_PI_2 && ray->r >= b->r_isco && ray->r < b->r_disc) {
                if (fabs(ray->r - b->source_r) > 0.1 * b->source_r || fabs(ray->phi - b->source_phi) > 0.1) return_count += ray_weight;
            } else if (ray->r > b->r_esc) {
                escape_count += ray_weight;
            } else if (ray->r < b->r_isco) {
                lost_count += ray_weight;
            }
        }
    }
    out[0] = ray_count; out[1] = return_count; out[2] = escape_count; out[3] = lost_count;
}

int kro_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
