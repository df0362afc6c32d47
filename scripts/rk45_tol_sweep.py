#!/usr/bin/env python3
"""BASELINE configs[2]: adaptive RK45 tolerance sweep (reference src/tests/emissivity_rk45_tol_sweep.py:38 tolerances,
src/tests/emissivity_rk45_plot.cpp:35-38 grid 0.01 x 0.01 = 125 863 allocated rays), source h = 5 (the reference's
sweep) and h = 10 (BASELINE), on one MI355X through the C ABI.  Prints one JSON document.
RK4 on the same grid is run beside it so the RK4-vs-RK45 emissivity deviation the reference's sweep plots can be formed."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from raytrace_cpu_amd import api, capi


class gc:                                   # the constants this sweep shares with the benchmark
    SPIN = bench.SPIN

    @staticmethod
    def emis_bins(spec, nr):
        n_primary = int(((spec.cosalphamax - spec.cosalpha0) / spec.dcosalpha) * ((spec.betamax - spec.beta0) / spec.dbeta))
        b = bench.emis_bins(capi, api.lib().kr_kerr_isco(bench.SPIN, 1), n_primary)
        b.dr = math.exp(math.log(bench.R_DISC / b.r_min) / nr)
        b.nr = nr
        return b

ARITH = {"hybrid": capi.FLAG_HYBRID, "strict": 0, "fast": capi.FLAG_FAST_MATH}[sys.argv[1] if len(sys.argv) > 1 else "hybrid"]
TOLS = [1e-6, 3e-7, 1e-7, 3e-8, 1e-8, 3e-9, 1e-9, 3e-10, 1e-10]
out = {"arithmetic": sys.argv[1] if len(sys.argv) > 1 else "hybrid", "grid": "dcosalpha = dbeta = 0.01, cos(alpha) in [-0.995, 0.995), beta in [-pi, pi)", "device": api.device_info(), "runs": []}
for h in (5.0, 10.0):
    spec = capi.PointSourceSpec()
    for i, v in enumerate([0.0, h, 1e-3, 0.0]): spec.pos[i] = v
    spec.V, spec.spin, spec.tol, spec.E = 0.0, gc.SPIN, 100.0, 1.0
    spec.cosalpha0, spec.cosalphamax, spec.dcosalpha = -0.995, 0.995, 0.01
    spec.beta0, spec.betamax, spec.dbeta = -math.pi, math.pi, 0.01
    init = api.pointsource_init(spec)
    api.redshift_start(gc.SPIN, 0.0, 0, 0, init)
    bins = gc.emis_bins(spec, nr=30)
    def run(method, tol):
        p = capi.default_params(gc.SPIN); p.integrator, p.rk45_tol, p.flags = method, tol, ARITH
        best = None
        for _ in range(2):
            rays, st = api.trace(p, init)
            if best is None or st["kernel_ms"] < best[1]["kernel_ms"]: best = (rays, st)
        rays, st = best
        api.range_phi(rays); api.redshift(gc.SPIN, -1.0, 0, 0, rays)
        return st, api.reduce_emissivity(bins, rays)
    st4, h4 = run(capi.RK4, 1e-8)
    out["runs"].append({"h": h, "integrator": "rk4", "rays": st4["rays_traced"], "steps": st4["steps_total"], "kernel_ms": st4["kernel_ms"], "steps_per_sec": st4["steps_total"] / st4["kernel_ms"] * 1e3})
    for tol in TOLS:
        st, hh = run(capi.RK45, tol)
        ok = (h4["count"] >= 100) & (hh["count"] >= 100)
        dev = np.abs(hh["emis"][ok] / h4["emis"][ok] - 1)
        out["runs"].append({"h": h, "integrator": "rk45", "tol": tol, "rays": st["rays_traced"], "steps": st["steps_total"], "attempts": st["rk45_attempts"], "rejects": st["rk45_rejects"],
                            "stationary_steps": st["rk45_stationary_steps"], "extrapolated_steps": st["rk45_extrapolated_steps"], "kernel_ms": st["kernel_ms"], "steps_per_sec": st["steps_total"] / st["kernel_ms"] * 1e3,
                            "attempts_per_sec": st["rk45_attempts"] / st["kernel_ms"] * 1e3, "emis_dev_vs_rk4_rms": float(np.sqrt(np.mean(dev ** 2))), "emis_dev_vs_rk4_max": float(dev.max())})
print(json.dumps(out, indent=1))
