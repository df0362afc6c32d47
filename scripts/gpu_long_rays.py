#!/usr/bin/env python3
"""GPU box: where do the long rays of the headline workload sit, and which of them does the hybrid classifier flag?
usage: scripts/gpu_long_rays.py [rays=1e7]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bench, parity
from raytrace_cpu_amd import api, capi
rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
spec = bench.make_spec(capi, bench.grid_spacing_for(rays_n))
total, n_ca, n_beta = api.pointsource_count(spec)
rays = api.pointsource_init(spec)
api.redshift_start(bench.SPIN, 0.0, 0, 0, rays)
ke = parity.knife_edge_mask(rays, False)
p = capi.default_params(bench.SPIN); p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, capi.FLAG_HYBRID
out, st = api.trace(p, rays)
steps = np.abs(out["steps"].astype(np.int64)); steps[out["steps"] == -1] = 0
print("stats", st, "n_beta", n_beta, "knife", int(ke.sum()))
col = np.arange(len(rays)) % n_beta
for thr in (2000, 5000, 10000, 20000, 30000):
    m = steps > thr
    cols, cnt = np.unique(col[m], return_counts=True)
    order = np.argsort(-cnt)[:8]
    print(f"> {thr}: {int(m.sum())} rays ({int((m & ke).sum())} flagged), steps share {steps[m].sum() / steps.sum() * 100:.2f}%, columns(top): " + ", ".join(f"{int(cols[i])}:{int(cnt[i])}" for i in order))
nk = ~ke
print("longest unflagged:", np.sort(steps[nk])[-10:].tolist())
big = np.argsort(np.where(nk, steps, 0))[-6:]
for i in big:
    print(int(i), "col", int(col[i]), "row", int(i // n_beta), "steps", int(steps[i]), "status", int(out["status"][i]), "h", float(rays["h"][i]), "cosalpha", float(rays["alpha"][i]), "beta", float(rays["beta"][i]))
h = np.abs(rays["h"])
for t in (1e-13, 1e-6, 1e-4, 1e-3, 3e-3, 1e-2):
    m = (h < t) & (out["steps"] != -1)
    print(f"|h| < {t:g}: {int(m.sum())} rays, max steps {int(steps[m].max()) if m.any() else 0}, holds {int((steps[m] > 10000).sum())} of the {int((steps > 10000).sum())} rays > 10000 steps")
