// kr_rk45.hpp -- the adaptive integrator: one wave iteration = at most one DOPRI5 TRIAL step per lane (reference raytracer.cpp:1260-1598, :1600-1894),
// the closed-form replay of stationary captured rays and the creep mode.  Included by kr_device.hpp (after step_fixed's helpers).
#pragma once

#include "kr_replay.hpp"

namespace kr {

constexpr int kCreepRun = 8;      // consecutive creeping outer steps (same ulp count) before the rest of a captured ray is extrapolated

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
    kr_sincos(theta, sin_theta, cos_theta);
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
template <typename T, bool USE_DEST, bool FAST, bool LONE = false>
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
        confirmed = !k1_with_flips<T, true, LONE>(s, a, rhosq, sin2theta);
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
template <typename T, bool USE_DEST, bool FAST, bool LONE = false>
KR_DEV bool step_rk45(Lane<T>& s, const TraceConsts<T>& c, uint32_t& attempts, uint32_t& rejects, uint32_t& stationary_steps, uint32_t& creep_steps,
                      int replay_batch)
{
    using D = Dopri<T>;
    const T a = c.a;

    if constexpr (sizeof(T) == 8) {
        if (s.creep_mode) {
            // replay_batch > 1 when every ray of the wave is in creep mode (the tail of a launch): several outer steps per wave iteration
            for (int u = 0; u < replay_batch; ++u) {
                const int rc = creep_step<T, USE_DEST, FAST, LONE>(s, c, attempts, creep_steps);
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
        T rhosq, sin2theta, y_rhosq;
        bool flipped;
        if constexpr (sizeof(T) == 8) {
            // wave-uniform: every lane's data from its last accepted stage is valid (else all recompute -- same bits either way)
            if (__builtin_amdgcn_ballot_w64(!s.fsal_valid) == 0) flipped = k1_from_last_stage<T>(s, a, rhosq, sin2theta, &y_rhosq);
            else flipped = k1_with_flips<T, true, LONE>(s, a, rhosq, sin2theta, &y_rhosq);
        } else {
            flipped = k1_with_flips<T, true, LONE>(s, a, rhosq, sin2theta, &y_rhosq);
        }
        if (flipped) return !(s.steps < c.steplim);
        // flags (:1403-1410): same rhosq / sin2theta values as k1's.  The NEG_ENERGY test is the conserved energy k evaluated from tdot and phidot: away
        // from the horizon, for k > 0 and |h| <= 1e6 k, its rounding cannot turn the sum negative (energy_guard_set) -- evaluated, with the reference's
        // operations and k1's refined reciprocal of rho^2, only by waves that hold a ray within 1e-6 of the horizon or one that carries the mark
        if (s.pt <= 0) s.status |= KR_STATUS_ERGO;
        if (sizeof(T) == 4 || __builtin_amdgcn_ballot_w64(energy_flag_needed(s, s.r - c.horizon)) != 0) {
            if ((1 - dv_y(2 * s.r, rhosq, y_rhosq)) * s.pt + dv_y(2 * a * s.r * sin2theta, rhosq, y_rhosq) * s.pphi < 0) s.status |= KR_STATUS_NEG_ENERGY;
        }
        // outer cap (:1421-1434): horizon / phi / t, no MIN_STEP floor afterwards.  The first quotient keeps the compiler's IEEE division: with
        // rdot == 0 it must be +inf (so that the other two caps still apply), where the lean chain gives NaN; for the other two, inf and NaN
        // alike make "cap < step_max" false.  / precision through the host's correctly rounded reciprocal (div_const: same bits).
        step_max = div_const(kr_abs((s.r - c.horizon) / s.pr), c.precision, c.inv_precision, c.inv_ok);
        {                                                  // max_phistep > 0: otherwise the quotient is inf / NaN and the comparison false
            const T step_phi = kr_abs(dv(c.phistep_eff, s.pphi));
            if (step_phi < step_max) step_max = step_phi;
        }
        if (s.r < c.tstep_rlim_eff) {                      // max_tstep > 0 && r < maxtstep_rlim
            const T step_t = kr_abs(dv(c.max_tstep, s.pt));
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
    T pr2, ptheta2, pr3, ptheta3, pr4, ptheta4, pr5, ptheta5, pr6, ptheta6;
    T sum_t, sum_phi;
    // (small: the side launches' optimistic form of the five stages, as in step_fixed -- the small-angle sine / cosine unasked while the base angle
    // is small on every lane, the stages redone with the full routine if one of their angles was not: two branches instead of ten)
    auto stages = [&](auto small) -> bool {
        constexpr bool kSmall = decltype(small)::value;
        bool within = true;
        T pt_i, pphi_i;
        auto stage = [&](T& pr_o, T& ptheta_o, T r_stage, T theta_stage) {
            if constexpr (kSmall) {
                within = within && (kr_abs(theta_stage) < T(KR_SMALL_ANGLE_LIMIT));
                momentum_impl<T, LONE, true>(pt_i, pr_o, ptheta_o, pphi_i, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r_stage, theta_stage, a);
            } else {
                eval<T, FAST, LONE>(pt_i, pr_o, ptheta_o, pphi_i, s, r_stage, theta_stage, a);
            }
        };
        sum_t = D::b1 * pt1;
        sum_phi = D::b1 * pphi1;

        stage(pr2, ptheta2, r + h_try * D::a21 * pr1, theta + h_try * D::a21 * ptheta1);

        stage(pr3, ptheta3, r + h_try * (D::a31 * pr1 + D::a32 * pr2), theta + h_try * (D::a31 * ptheta1 + D::a32 * ptheta2));
        sum_t = sum_t + D::b3 * pt_i;
        sum_phi = sum_phi + D::b3 * pphi_i;

        stage(pr4, ptheta4, r + h_try * (D::a41 * pr1 + D::a42 * pr2 + D::a43 * pr3), theta + h_try * (D::a41 * ptheta1 + D::a42 * ptheta2 + D::a43 * ptheta3));
        sum_t = sum_t + D::b4 * pt_i;
        sum_phi = sum_phi + D::b4 * pphi_i;

        stage(pr5, ptheta5, r + h_try * (D::a51 * pr1 + D::a52 * pr2 + D::a53 * pr3 + D::a54 * pr4),
              theta + h_try * (D::a51 * ptheta1 + D::a52 * ptheta2 + D::a53 * ptheta3 + D::a54 * ptheta4));
        sum_t = sum_t + D::b5 * pt_i;
        sum_phi = sum_phi + D::b5 * pphi_i;

        stage(pr6, ptheta6, r + h_try * (D::a61 * pr1 + D::a62 * pr2 + D::a63 * pr3 + D::a64 * pr4 + D::a65 * pr5),
              theta + h_try * (D::a61 * ptheta1 + D::a62 * ptheta2 + D::a63 * ptheta3 + D::a64 * ptheta4 + D::a65 * ptheta5));
        sum_t = sum_t + D::b6 * pt_i;
        sum_phi = sum_phi + D::b6 * pphi_i;
        return within;
    };
    if constexpr (LONE && !FAST && sizeof(T) == 8) {
        bool done = false;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(kr_abs(theta) < T(KR_SMALL_ANGLE_LIMIT))) == 0, 1)) done = stages(std::true_type{});
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!done) != 0, 0)) {
            if (!done) stages(std::false_type{});
        }
    } else {
        stages(std::false_type{});
    }

    // 5th-order solution (:1493-1499); the polar reflection mutates thetadot_sign even if the trial is rejected
    const T inc_r = h_try * (D::b1 * pr1 + D::b3 * pr3 + D::b4 * pr4 + D::b5 * pr5 + D::b6 * pr6);
    const T inc_theta = h_try * (D::b1 * ptheta1 + D::b3 * ptheta3 + D::b4 * ptheta4 + D::b5 * ptheta5 + D::b6 * ptheta6);
    T r_new = r + inc_r;
    T theta_new = theta + inc_theta;
    T t_new = s.t + h_try * sum_t;
    T phi_new = s.phi + h_try * sum_phi;
    const bool inside_poles = !(theta_new < T(0)) && !(theta_new > T(kPi));
    reflect_poles<T, LONE>(theta_new, phi_new, s.thetadot_sign);

    T pt7, pr7, ptheta7, pphi7;
    Lane<T> last;                   // (only its f_* members are written, and only on the strict double path)
    if constexpr (!FAST && sizeof(T) == 8)
        momentum_impl<T, LONE>(pt7, pr7, ptheta7, pphi7, s.k, s.h, s.Q, s.rdot_sign, s.thetadot_sign, r_new, theta_new, a, &last);
    else
        eval<T, FAST, LONE>(pt7, pr7, ptheta7, pphi7, s, r_new, theta_new, a);

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
    // (the compiler's IEEE quotients and root stay here: an infinite scale or norm -- a diverging trial -- must come out as 0 / inf, as in the reference,
    // where the lean chains would say NaN and end the ray)

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
    // step.  Once the step has been of this kind kCreepRun times in a row, with margins that 1e-11 cannot consume (trial
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
                if (s.creep_run >= kCreepRun && remaining > 0) {
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

}  // namespace kr
