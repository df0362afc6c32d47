"""Shared comparison logic for the GPU parity tests (TEST INFRASTRUCTURE).

What "parity" means here (stated once, used by every test):

* The HIP path is built with -ffp-contract=off and uses IEEE-correct fp64 +,-,*,/,sqrt, so a step differs
  from the CPU reference only through the device libm (sin, cos: <= 1-2 ulp from glibc).  A 1-ulp input
  difference grows along the trajectory; for ordinary rays it stays near 1e-13 relative, for rays that orbit
  near the photon sphere it is amplified exponentially, exactly as between two CPU libms.
* per ray: integer outputs (status, rdot_flips, equatorial_crossings) equal; step count EQUAL for Euler / RK4 on the strict
  and hybrid arithmetic, within +-2 for RK45 and for the opt-in fast arithmetic (steps_slack_for);
  t, r, theta, phi, redshift within RAY_RTOL relative (absolute for |x| < 1): 1e-9 for the fixed-step
  integrators, 1e-7 for RK45, whose step controller divides by an error estimate that is a cancellation of
  O(1) terms down to O(tol) and therefore turns 1 ulp into ~1e-10..1e-8 of step size.  These bands are the
  reference's OWN rounding-noise envelope: perturbing Q by 1 ulp on the CPU oracle moves disc-hit radii by
  median 3e-14 / p99 6e-7 (RK4) and median 7e-12 / p99 2e-8 / max 8e-7 (RK45) on the ps_h10 case
  (tests/test_oracle_sensitivity.py pins those numbers);
  at most CHAOTIC_FRAC of the traced rays may violate that (photon-sphere rays);
* rays that end on the horizon or at the step limit: t and phi are NOT compared (both diverge at the horizon,
  dt/dlambda ~ 1/Delta, so the last steps amplify 1 ulp without bound; every consumer drops these rays:
  steps < 0 or r < r_isco), r / theta / momenta-free integer outputs are;
* per bin (emissivity profile, BASELINE.json north star): ray count exact (+-BIN_COUNT_SLACK when a chaotic ray
  moved), flux / emis / sum_redshift / sum_time within BIN_RTOL = 1e-6 relative on bins whose counts agree.
"""
import numpy as np

from raytrace_cpu_amd import capi

RAY_RTOL = 1e-9
RAY_RTOL_RK45 = 1e-7
CHAOTIC_FRAC = 0.01
BIN_RTOL = 1e-6
BIN_COUNT_SLACK = 1

FLOAT_FIELDS = ("t", "r", "theta", "phi", "pt", "pr", "ptheta", "pphi")
INT_FIELDS = ("status", "rdot_sign", "thetadot_sign", "rdot_flips", "equatorial_crossings")


def same_records(a, b):
    """Every field of two ray arrays bit for bit -- except that a NaN equals a NaN whatever its sign and payload.  The rays of the beta = -pi column
    that run into the polar axis end with every coordinate NaN (in the reference as well), and WHICH NaN an instruction sequence hands on depends on
    which operand the compiler put a negation on, not on the arithmetic; the side launch's kernels and the ordinary strict ones are different
    instruction sequences for the same operations."""
    def same(x, y):
        if x.dtype.kind == "f":
            u = f"u{x.dtype.itemsize}"
            return bool(np.all((x.view(u) == y.view(u)) | (np.isnan(x) & np.isnan(y))))
        return bool(np.array_equal(x, y))
    return a.dtype == b.dtype and a.shape == b.shape and all(same(a[f], b[f]) for f in a.dtype.names)


def rtol_for(params):
    return RAY_RTOL_RK45 if params.integrator == capi.RK45 else RAY_RTOL


def steps_slack_for(params, flags=0):
    """Allowed |delta steps| per ray: the fixed-step integrators on the strict / hybrid arithmetic must take exactly the
    reference's number of steps (a ray that does not counts as bad, i.e. against the chaotic-ray allowance); RK45's step
    controller and the opt-in fast arithmetic may move a step boundary by one or two."""
    return 2 if (params.integrator == capi.RK45 or (flags & capi.FLAG_FAST_MATH)) else 0


def compare_rays(got, want, rtol=RAY_RTOL, check_redshift=False, steps_slack=2):
    """Per-ray comparison.  Returns dict(n_traced, n_bad, frac_bad, worst) where a ray is 'bad' if any integer
    output differs, |steps| differs by more than steps_slack, or a float output is off by more than rtol."""
    assert len(got) == len(want)
    untouched = want["steps"] == -1
    assert (got["steps"][untouched] == -1).all(), "a never-initialised ray was modified"
    live = ~untouched
    bad = np.zeros(len(got), dtype=bool)
    for f in INT_FIELDS:
        bad |= live & (got[f] != want[f])
    dsteps = np.abs(np.abs(got["steps"].astype(np.int64)) - np.abs(want["steps"].astype(np.int64)))
    bad |= live & (dsteps > steps_slack)
    bad |= live & (np.sign(got["steps"]) != np.sign(want["steps"]))
    worst = 0.0
    worst_any = np.zeros(len(got))                  # per ray: its largest relative error over the compared float fields
    fields = FLOAT_FIELDS + (("redshift",) if check_redshift else ())
    sunk = (want["status"] & (capi.STATUS_HORIZON | capi.STATUS_STEPLIM)) != 0
    for f in fields:
        g, w = got[f], want[f]
        if f not in ("r", "theta"):
            g, w = np.where(sunk, 0.0, g), np.where(sunk, 0.0, w)
        both_nan = np.isnan(g) & np.isnan(w)
        scale = np.maximum(np.abs(w), 1.0)
        with np.errstate(invalid="ignore"):
            err = np.abs(g - w) / scale
        err = np.where(both_nan, 0.0, err)
        err = np.where(np.isnan(err), np.inf, err)
        err = np.where((g == w), 0.0, err)          # equal infinities
        bad |= live & (err > rtol)
        worst_any = np.maximum(worst_any, np.where(live, err, 0.0))
        ok = live & ~bad
        if ok.any():
            worst = max(worst, float(err[ok].max()))
    n_live = int(live.sum())
    int_bad = np.zeros(len(got), dtype=bool)
    for f in INT_FIELDS:
        int_bad |= live & (got[f] != want[f])
    term_bad = live & (terminal_bits(got["status"]) != terminal_bits(want["status"]))
    # rays that carry the reference's bits in every compared output (t, r, theta, phi and the integer fields)
    bits = live & ~int_bad & (got["steps"] == want["steps"])
    for f in FLOAT_FIELDS:
        bits &= (got[f].view(np.int64) == want[f].view(np.int64)) | (np.isnan(got[f]) & np.isnan(want[f]))
    # how far out the rays beyond the bar are (finite errors only; a NaN / inf mismatch counts as inf), and which rays they are
    wb = worst_any[bad]
    worst_bad = float(wb.max()) if len(wb) else 0.0
    return {"worst_bad": worst_bad, "median_bad": float(np.median(wb)) if len(wb) else 0.0, "frac_bit_identical": float(bits.sum()) / max(n_live, 1), "n_traced": n_live, "n_bad": int(bad.sum()), "frac_bad": float(bad.sum()) / max(n_live, 1), "worst_ok": worst,
            "n_steps_differ": int((live & (dsteps > 0)).sum()), "n_int_fields_differ": int(int_bad.sum()),
            "frac_terminal_status_differs": float(term_bad.sum()) / max(n_live, 1), "bad_index": np.flatnonzero(bad)}


_BIN_EXCLUSIONS = []


def compare_bins(got, want, rtol=BIN_RTOL, slack=BIN_COUNT_SLACK, max_excluded=None, label=""):
    """got/want: dicts with count, flux, emis, sum_redshift, sum_time.  Returns list of problems (empty = pass).
    A bin whose count differs by one (a chaotic ray landed next door) cannot be held to 1e-6 in its sums and is left out of the sum check -- but how
    many bins that may be is bounded: max_excluded (default: 2 % of the non-empty bins, at least 2: one moved ray changes two bins), and the number
    is recorded with the round's margins (gpurun_out/parity_margins.json)."""
    problems = []
    dc = np.abs(got["count"].astype(np.int64) - want["count"].astype(np.int64))
    if (dc > slack).any():
        problems.append(("count", int(dc.max()), int(np.argmax(dc))))
    same = dc == 0
    nonempty = int((want["count"] > 0).sum())
    excluded = int((~same & (want["count"] > 0)).sum())
    bound = max(2, nonempty // 50) if max_excluded is None else max_excluded
    _BIN_EXCLUSIONS.append({"label": label, "bins_nonempty": nonempty, "bins_excluded_from_the_sum_check": excluded, "allowed": bound})
    if excluded > bound:
        problems.append(("excluded_bins", excluded, bound))
    for k in ("flux", "emis", "sum_redshift", "sum_time"):
        g, w = got[k][same], want[k][same]
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.abs(g - w) / np.maximum(np.abs(w), 1e-300)
        rel = np.where(g == w, 0.0, rel)
        if (rel > rtol).any():
            problems.append((k, float(rel.max()), int(np.argmax(rel))))
    return problems


def noise_envelope_frac(params, init, rtol):
    """The reference algorithm's own rounding-noise envelope for this run: the fraction of rays that move by more
    than rtol (or change an integer output) when Q is perturbed by 1 ulp on the CPU oracle.  Runs whose end
    point is not converged onto a surface (e.g. RK45 + FlatPlane, which has no step_limit) are ill-conditioned
    in the reference itself; a GPU libm cannot be asked to agree better than the reference agrees with itself."""
    import oracle_lib as ol
    base, _ = ol.oracle_trace(params, init)
    pert = init.copy()
    pert["Q"] = np.nextafter(pert["Q"], np.inf)
    out, _ = ol.oracle_trace(params, pert)
    return compare_rays(out, base, rtol=rtol)["frac_bad"]


MAX_BAD_FRAC = 0.05
# The fast arithmetic on the Keplerian-source fixture: on the 320-ray grid 4.4-4.7 % of the rays sit beyond 1e-9 (reference's own 1-ulp envelope
# 2.8-3.4 %) -- a coarse-grid artefact, the same geometry on 5040 rays leaves 1.13-1.15 % (envelope 0.9-1.0 %; profiles/r03_parity_margins.json).
# The large fixture is therefore held to twice its measured share, so that a regression of the arithmetic shows; the small one keeps the global cap.
HYBRID_CAP = {"ps_kep5k": 0.023}
STRICT_BAD_FRAC = 1e-3      # fixed-step integrators on the strict arithmetic: measured 0 bad rays on every PointSource fixture, 2 of 289 on the
STRICT_BAD_RAYS = 2.5       # image-plane grid that contains the x = 0 column and the NaN pixel (profiles/r02_parity_margins.json)


def is_unconverged_endpoint(params):
    """RK45 + FlatPlaneDestination: the destination has no step_limit(), so the adaptive step overshoots the plane by a
    rounding-dependent amount and the end POSITION of every ray is ill-conditioned in the reference itself (12 % of its rays
    move by more than 1e-7 under a 1-ulp change of Q, tests/test_oracle_sensitivity.py)."""
    return params.integrator == capi.RK45 and params.stop_kind == capi.STOP_FLATPLANE


def allowed_bad_frac(params, init, rtol, envelope=None):
    """1 % (photon-sphere rays) + 3 x the reference's own 1-ulp noise envelope for this run, never more than MAX_BAD_FRAC = 5 %;
    the one run whose end points are not converged (is_unconverged_endpoint) keeps the uncapped envelope bar for positions and
    is held to >= 99 % agreement of the terminal status instead (test_trace_vs_golden)."""
    env = noise_envelope_frac(params, init, rtol) if envelope is None else envelope
    bar = CHAOTIC_FRAC + 3 * env
    return bar if is_unconverged_endpoint(params) else min(bar, MAX_BAD_FRAC)


def allowed_bad_frac_strict(params, n_traced):
    """Euler / RK4 with flags = 0: the only difference from the CPU is sin / cos in the last place, and the fixtures show what that
    costs -- nothing.  The bar is therefore fixed, not derived from the noise envelope: 1e-3 of the rays, or 2 rays on a small grid."""
    assert params.integrator != capi.RK45
    return max(STRICT_BAD_FRAC, STRICT_BAD_RAYS / max(n_traced, 1))


# ---- measured margins: every comparison that goes through record_margin ends up in gpurun_out/parity_margins.json (copied to
# profiles/ per round), so that how close the HIP path runs to each bar is on record, not just pass / fail
_MARGINS = []


def record_margin(test, case, res, allowed=None, envelope=None, **extra):
    row = {"test": test, "case": case, "n_traced": res["n_traced"], "n_bad": res["n_bad"], "frac_bad": res["frac_bad"], "worst_ok": res["worst_ok"],
           "worst_bad": res.get("worst_bad"), "median_bad": res.get("median_bad"), "bad_index": [int(i) for i in res.get("bad_index", [])[:64]],
           "frac_bit_identical": res.get("frac_bit_identical"),
           "n_steps_differ": res.get("n_steps_differ"), "n_int_fields_differ": res.get("n_int_fields_differ"),
           "frac_terminal_status_differs": res.get("frac_terminal_status_differs"), "allowed_bad_frac": allowed, "noise_envelope_frac": envelope}
    row.update(extra)
    _MARGINS.append(row)
    return row


def dump_margins(path):
    import json
    import os
    if not _MARGINS:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"what": "measured per-ray parity margins of the HIP path against the reference fixtures / the oracle (tests/parity.py)", "rows": _MARGINS,
                   "bins_excluded_from_sum_checks": _BIN_EXCLUSIONS}, f, indent=1)


def knife_edge_mask(init, imageplane):
    """Rays whose fate IN THE REFERENCE is decided by rounding noise: a PointSource ray emitted at beta = -pi has
    sin(beta) = -1.2e-16, an ImagePlane ray on the x = 0 column has cos(beta) = 6e-17 and one on the y = 0 row has
    l_theta = 0, so their theta-motion / axial angular momentum h is a pure cancellation residue.  The strict path reproduces
    that residue bit for bit and therefore the reference's outcome; KR_FLAG_FAST_MATH changes the residue (any
    re-association does) and such a ray can end somewhere else entirely.  Fast-math parity is asserted on all OTHER
    rays, at the unchanged tolerances; this set has measure zero in the ray grid's continuum limit (one grid column)."""
    if imageplane:
        return (np.abs(init["alpha"]) < 1e-9) | (np.abs(init["beta"]) < 1e-9)
    return np.abs(np.sin(init["beta"])) < 1e-9


def drop_rays(rays, mask):
    out = rays.copy()
    out["steps"][mask] = -1
    return out


def terminal_bits(status):
    return status & (capi.STATUS_DEST | capi.STATUS_HORIZON | capi.STATUS_RLIM | capi.STATUS_STEPLIM | capi.STATUS_NAN)
