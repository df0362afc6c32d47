#!/usr/bin/env python3
"""GPU box: step-count distribution of the bench workload (who sets the kernel's tail?)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, bench
from raytrace_cpu_amd import api, capi
rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 3e6
method = {"euler": capi.EULER, "rk4": capi.RK4, "rk45": capi.RK45}[sys.argv[2] if len(sys.argv) > 2 else "rk4"]
lib = api.lib()
spec = bench.make_spec(capi, bench.grid_spacing_for(rays_n))
rays = api.pointsource_init(spec)
api.redshift_start(bench.SPIN, 0.0, 0, 0, rays)
p = capi.default_params(bench.SPIN); p.integrator, p.r_max = method, bench.R_MAX
out, st = api.trace(p, rays)
s = np.abs(out["steps"][out["steps"] != -1].astype(np.int64))
print("stats", st)
print("rays", len(s), "sum", s.sum(), "mean", s.mean(), "median", np.median(s), "p99", np.percentile(s, 99), "p99.9", np.percentile(s, 99.9), "p99.99", np.percentile(s, 99.99), "max", s.max())
top = np.sort(s)[-20:]
print("top20", top.tolist())
for thr in (1000, 2000, 5000, 10000, 20000, 50000, 100000):
    m = s > thr
    print(f"> {thr}: {m.sum()} rays, {s[m].sum()/s.sum()*100:.2f}% of steps")
big = np.argsort(np.abs(out["steps"].astype(np.int64)))[-5:]
for i in big: print(i, out["steps"][i], out["status"][i], out["r"][i], out["alpha"][i], out["beta"][i], out["rdot_flips"][i], out["equatorial_crossings"][i])
