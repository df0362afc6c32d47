#!/usr/bin/env python3
"""GPU box: the longest strict RK45 ray of the headline source traced ALONE (one ray in the launch), with its 63 neighbours of the beta = -pi column,
and with the whole column: how much does a long ray pay for the other control paths (creep mode, replay, retries) of the lanes it shares a wave with?"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, bench
from raytrace_cpu_amd import api, capi
method = {"rk4": capi.RK4, "rk45": capi.RK45}[sys.argv[1] if len(sys.argv) > 1 else "rk45"]
spec = bench.make_spec(capi, bench.grid_spacing_for(1e7))
spec.beta0, spec.betamax, spec.dbeta = -math.pi, -math.pi + 1e-9, 1.0          # the beta = -pi column: 3162 rays
rays = api.pointsource_init(spec)
api.redshift_start(bench.SPIN, 0.0, 0, 0, rays)
p = capi.default_params(bench.SPIN); p.integrator, p.r_max = method, bench.R_MAX
out, st = api.trace(p, rays)
steps = np.abs(out["steps"].astype(np.int64))
order = np.argsort(steps)[::-1]
print("column:", len(rays), "rays, kernel_ms", round(st["kernel_ms"], 1), "longest", steps[order[:5]].tolist(), "rays >= 50000 steps:", int((steps >= 50000).sum()), ">= 20000:", int((steps >= 20000).sum()))
i = int(order[0])
for label, idx in (("longest ray alone", [i]), ("the 8 longest", order[:8].tolist()), ("the 64 longest", order[:64].tolist()), ("64 consecutive incl. the longest", list(range(max(0, i - 32), max(0, i - 32) + 64)))):
    sub = rays[idx].copy()
    ts = []
    for _ in range(3):
        o, s2 = api.trace(p, sub)
        ts.append(s2["kernel_ms"])
    mx = int(np.abs(o["steps"]).max())
    print(f"{label}: kernel_ms {min(ts):.1f}  max steps {mx}  us per step of the longest {1e3 * min(ts) / mx:.3f}", flush=True)
