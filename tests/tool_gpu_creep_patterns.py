#!/usr/bin/env python3
"""GPU box tool: what do the RK45 rays that run into the step limit do per outer step late in their life?
Traces the bench source grid with steplim = K, K+1, ... (pure iteration) and histograms (d theta, d r) in ulps per step."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bench
from raytrace_cpu_amd import api, capi
rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
ARITH = {"fast": capi.FLAG_FAST_MATH, "strict": 0, "hybrid": capi.FLAG_HYBRID}[sys.argv[2] if len(sys.argv) > 2 else "fast"]
spec = bench.make_spec(capi, bench.grid_spacing_for(rays_n))
init = api.pointsource_init(spec)
api.redshift_start(bench.SPIN, 0.0, 0, 0, init)
p = capi.default_params(bench.SPIN); p.integrator, p.r_max = capi.RK45, bench.R_MAX
full, st = api.trace(capi.copy_params(p, flags=ARITH | capi.FLAG_RK45_ITERATE_ALL), init)
lim = np.flatnonzero((full["status"] & capi.STATUS_STEPLIM) != 0)
print("rays", len(init), "steplim rays", len(lim), st)
sub = init[lim].copy()
K = 60000
states = []
for k in range(K, K + 5):
    out, _ = api.trace(capi.copy_params(p, steplim=k, flags=ARITH | capi.FLAG_RK45_ITERATE_ALL), sub)
    states.append(out)
u = lambda a: a.view(np.uint64).astype(np.int64)
pat = collections.Counter()
for i in range(4):
    pass
dth = np.stack([u(states[i + 1]["theta"].copy()) - u(states[i]["theta"].copy()) for i in range(4)], 1)
dr = np.stack([u(states[i + 1]["r"].copy()) - u(states[i]["r"].copy()) for i in range(4)], 1)
for a, b in zip(dth, dr):
    pat[(tuple(int(x) for x in a), tuple(int(x) for x in b))] += 1
for k, v in pat.most_common(15):
    print(v, k)
# flips / crossings changing?
for f in ("rdot_flips", "equatorial_crossings", "thetadot_sign", "rdot_sign"):
    print(f, "changes between K and K+4:", int((states[4][f] != states[0][f]).sum()))

# extrapolation on: how many of the creeping rays does it take, and does it reproduce the iterated theta / integers exactly?
ext, st2 = api.trace(capi.copy_params(p, flags=ARITH), init)
print("with extrapolation:", {k: st2[k] for k in ("kernel_ms", "rk45_stationary_steps", "rk45_extrapolated_steps")}, "rays extrapolated ~", st2["rk45_extrapolated_steps"] / 96000.0)
creep = lim[(dth == 1).all(1)]
same_theta = (ext["theta"][creep].view(np.uint64) == full["theta"][creep].view(np.uint64))
ints = np.ones(len(init), dtype=bool)
for f in ("status", "steps", "rdot_flips", "equatorial_crossings", "rdot_sign", "thetadot_sign"):
    ints &= ext[f] == full[f]
print("creeping rays", len(creep), "theta bit-identical to iterating:", int(same_theta.sum()), "r identical:", int((ext["r"][creep] == full["r"][creep]).sum()),
      "all-ray integer fields identical:", bool(ints.all()), "max rel t diff on creeping rays %.2e" % np.max(np.abs(ext["t"][creep] - full["t"][creep]) / np.abs(full["t"][creep])))
other = np.setdiff1d(np.arange(len(init)), creep)
print("non-creeping rays bitwise identical:", all(np.array_equal(ext[f][other].view(np.uint8), full[f][other].view(np.uint8)) for f in ext.dtype.names if f not in ("emit", "redshift")))
for f in ("status", "steps", "rdot_flips", "equatorial_crossings", "rdot_sign", "thetadot_sign"):
    d = np.flatnonzero(ext[f] != full[f])
    if len(d):
        print(f, "differs on", len(d), "rays; in creeping set:", int(np.isin(d, creep).sum()), "examples", [(int(i), int(ext[f][i]), int(full[f][i])) for i in d[:5]])
# which creeping rays were NOT extrapolated?  (their t differs from iteration by exactly 0)
took = np.abs(ext["t"][creep] - full["t"][creep]) > 0
print("creeping rays with t != iterated (i.e. extrapolated):", int(took.sum()))
nt = creep[~took]
print("not extrapolated examples: theta", full["theta"][nt[:5]], "r-rh", full["r"][nt[:5]] - 1.0632139225171164, "thetadot_sign", full["thetadot_sign"][nt[:5]], "status", full["status"][nt[:5]])
tk = creep[took]
print("extrapolated examples:     theta", full["theta"][tk[:5]], "r-rh", full["r"][tk[:5]] - 1.0632139225171164, "thetadot_sign", full["thetadot_sign"][tk[:5]], "status", full["status"][tk[:5]])
print("theta range not-extrapolated", full["theta"][nt].min(), full["theta"][nt].max(), " extrapolated", full["theta"][tk].min(), full["theta"][tk].max())
