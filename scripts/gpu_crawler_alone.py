#!/usr/bin/env python3
"""GPU box: how long does the headline's longest ray (row 147 of the beta = -pi column: 39 280 steps) take when its wave carries nothing else,
and when it shares the wave with its neighbours?  A wave executes a full step whenever ANY of its lanes takes one; a theta-flip iteration
(k1 only, `continue`) is cheap only if no other lane steps in that iteration.  usage: scripts/gpu_crawler_alone.py [rays=1e7] [integrator=rk4]"""
import json, math, os, sys
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
p = capi.default_params(bench.SPIN)
p.integrator, p.r_max, p.flags = integ, bench.R_MAX, 0
full, st = api.trace(p, init.copy())
steps = np.abs(full["steps"].astype(np.int64))
order = np.argsort(-steps)
sets = {"longest ray alone": order[:1], "the 2 longest": order[:2], "the 8 longest": order[:8], "the 32 longest": order[:32], "the 64 longest": order[:64],
        "rows 128-191 (its wave in the side launch)": np.arange(128, 192), "whole column": np.arange(len(init))}
for name, idx in sets.items():
    ms = []
    for _ in range(3):
        out, st = api.trace(p, init[np.sort(idx)].copy())
        ms.append(st["kernel_ms"])
    print(json.dumps({"set": name, "rays": int(len(idx)), "longest": int(st["longest_ray_steps"]), "steps_total": int(st["steps_total"]), "kernel_ms": round(min(ms), 2),
                      "us_per_step_of_longest": round(1e3 * min(ms) / st["longest_ray_steps"], 3)}), flush=True)
