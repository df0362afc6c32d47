#!/usr/bin/env python3
"""GPU box: the beta = -pi column of the headline grid (the hybrid launch's strict side launch) traced with the fast and with the strict
arithmetic: which of its rays would come out differently -- and are the long polar-axis crawlers among them?
usage: scripts/gpu_axis_column.py [rays=1e7] [integrator=rk4]"""
import ctypes as C, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from raytrace_cpu_amd import api, capi

rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
integ = {"rk4": capi.RK4, "euler": capi.EULER, "rk45": capi.RK45}[sys.argv[2] if len(sys.argv) > 2 else "rk4"]
spec = bench.make_spec(capi, bench.grid_spacing_for(rays_n))
spec.beta0, spec.betamax, spec.dbeta = -math.pi, -math.pi + 1e-9, 1.0          # one column
init = api.pointsource_init(spec)
api.redshift_start(bench.SPIN, 0.0, 0, 0, init)
res = {}
for flags, mode in ((0, "strict"), (capi.FLAG_FAST_MATH, "fast")):
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.flags = integ, bench.R_MAX, flags
    out, st = api.trace(p, init.copy())
    res[mode] = out
    print(mode, "kernel_ms", round(st["kernel_ms"], 2), "longest", st["longest_ray_steps"], "rays", st["rays_traced"], flush=True)
s, f = res["strict"], res["fast"]
valid = s["steps"] != -1
ints = np.zeros(len(s), dtype=bool)
for k in ("status", "steps", "rdot_flips", "equatorial_crossings", "rdot_sign", "thetadot_sign"):
    ints |= s[k] != f[k]
with np.errstate(invalid="ignore", divide="ignore"):
    rel = np.maximum(np.abs(f["r"] - s["r"]) / np.maximum(np.abs(s["r"]), 1e-300), np.abs(f["theta"] - s["theta"]) / np.maximum(np.abs(s["theta"]), 1e-300))
steps = np.abs(s["steps"].astype(np.int64))
order = np.argsort(-steps)
print(json.dumps({"rays": int(valid.sum()), "integer_fields_differ": int((valid & ints).sum()), "beyond_1e-9": int((valid & ~ints & (rel > 1e-9)).sum()),
                  "h_values": [float(x) for x in np.unique(init["h"])[:5]],
                  "longest20": [{"row": int(i), "steps_strict": int(s["steps"][i]), "steps_fast": int(f["steps"][i]), "ints_differ": bool(ints[i]), "rel": float(rel[i]), "status": int(s["status"][i])} for i in order[:20]],
                  "differing_rows_by_steps": sorted(((int(steps[i]), int(i)) for i in np.flatnonzero(valid & ints)), reverse=True)[:20]}))
