#!/usr/bin/env python3
"""GPU box: per-step latency of ONE wave alone on the chip (the kernel's critical path is its longest ray)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, bench
from raytrace_cpu_amd import api, capi
spec = bench.make_spec(capi, bench.grid_spacing_for(3e6))
rays = api.pointsource_init(spec)
api.redshift_start(bench.SPIN, 0.0, 0, 0, rays)
for name, method in (("rk4", capi.RK4), ("euler", capi.EULER)):
    p = capi.default_params(bench.SPIN); p.integrator, p.r_max = method, bench.R_MAX
    for n in (1, 64, 256):
        sub = rays[140292:140292 + n].copy()
        ts = []
        for _ in range(3):
            out, st = api.trace(p, sub)
            ts.append(st["kernel_ms"])
        mx = np.abs(out["steps"]).max()
        print(name, "n", n, "max steps", mx, "kernel_ms", min(ts), "us/step", 1e3 * min(ts) / mx)
