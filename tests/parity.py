"""Shared comparison logic for the GPU parity tests (TEST INFRASTRUCTURE).

What "parity" means here (stated once, used by every test):

* The HIP path is built with -ffp-contract=off and uses IEEE-correct fp64 +,-,*,/,sqrt, so a step differs
  from the CPU reference only through the device libm (sin, cos: <= 1-2 ulp from glibc).  A 1-ulp input
  difference grows along the trajectory; for ordinary rays it stays near 1e-13 relative, for rays that orbit
  near the photon sphere it is amplified exponentially, exactly as between two CPU libms.
* per ray: integer outputs (status, rdot_flips, equatorial_crossings) equal and |steps| within +-2;
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


def rtol_for(params):
    return RAY_RTOL_RK45 if params.integrator == capi.RK45 else RAY_RTOL


def compare_rays(got, want, rtol=RAY_RTOL, check_redshift=False):
    """Per-ray comparison.  Returns dict(n_traced, n_bad, frac_bad, worst) where a ray is 'bad' if any integer
    output differs, |steps| differs by more than 2, or a float output is off by more than rtol."""
    assert len(got) == len(want)
    untouched = want["steps"] == -1
    assert (got["steps"][untouched] == -1).all(), "a never-initialised ray was modified"
    live = ~untouched
    bad = np.zeros(len(got), dtype=bool)
    for f in INT_FIELDS:
        bad |= live & (got[f] != want[f])
    bad |= live & (np.abs(np.abs(got["steps"].astype(np.int64)) - np.abs(want["steps"].astype(np.int64))) > 2)
    bad |= live & (np.sign(got["steps"]) != np.sign(want["steps"]))
    worst = 0.0
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
        ok = live & ~bad
        if ok.any():
            worst = max(worst, float(err[ok].max()))
    n_live = int(live.sum())
    return {"n_traced": n_live, "n_bad": int(bad.sum()), "frac_bad": float(bad.sum()) / max(n_live, 1), "worst_ok": worst,
            "bad_index": np.flatnonzero(bad)}


def compare_bins(got, want, rtol=BIN_RTOL, slack=BIN_COUNT_SLACK):
    """got/want: dicts with count, flux, emis, sum_redshift, sum_time.  Returns list of problems (empty = pass)."""
    problems = []
    dc = np.abs(got["count"].astype(np.int64) - want["count"].astype(np.int64))
    if (dc > slack).any():
        problems.append(("count", int(dc.max()), int(np.argmax(dc))))
    same = dc == 0
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


def allowed_bad_frac(params, init, rtol):
    return CHAOTIC_FRAC + 3 * noise_envelope_frac(params, init, rtol)


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
