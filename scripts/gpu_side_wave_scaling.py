#!/usr/bin/env python3
"""GPU box: speed of the strict side launch's lone waves against how many of them are at work.
The beta = -pi column of the headline source (3162 rays, ~1300 of them 1e4..4e4 RK4 steps) is replicated R times into one array of
flagged rays and traced with flags = 0 ... through the split path (n >= 2^18 is forced by padding with unused slots), so that
R x 50 waves of the HOG instance run.  Reports strict_side_ms and us per step of the longest ray for R = 1, 2, 4, 8."""
import ctypes as C, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from raytrace_cpu_amd import api, capi
lib = api.lib(); vp = C.c_void_p
d = bench.grid_spacing_for(1e7)
s = bench.make_spec(capi, d)
s.beta0 = -math.pi; s.betamax = -math.pi + 0.5 * s.dbeta
col = api.pointsource_init(s)
api.redshift_start(bench.SPIN, 0.0, 0, 0, col)
live = col[col["steps"] == 0]
method = {"rk4": capi.RK4, "rk45": capi.RK45}[sys.argv[1] if len(sys.argv) > 1 else "rk4"]
p = capi.default_params(bench.SPIN); p.integrator, p.r_max, p.flags = method, bench.R_MAX, 0
for waves_target in (1, 4, 8, 16, 50, 100, 200, 400):
    nrays = waves_target * 64
    reps = -(-nrays // len(live))
    arr = np.concatenate([live] * reps)[:nrays].copy()
    if waves_target < 50:
        # few waves: take the longest rays so that every wave has a long one (rows 150.. of the column)
        order = np.argsort(-np.abs(live["h"])) if False else np.arange(len(live))
        arr = np.concatenate([live[147:147 + 64]] * waves_target)[:nrays].copy()
    pad = np.zeros(max(0, (1 << 18) - len(arr)), dtype=capi.RAY_F64); pad["steps"] = -1
    full = np.concatenate([arr, pad])
    dptr = vp(); capi.check(lib, lib.kr_malloc(C.byref(dptr), full.nbytes), "malloc")
    best = None
    for rep in range(2):
        capi.check(lib, lib.kr_memcpy_h2d(dptr, full.ctypes.data_as(vp), full.nbytes), "h2d")
        st = api.trace_dev(p, dptr.value, len(full))
        if best is None or st["strict_side_ms"] < best["strict_side_ms"]: best = st
    out = np.zeros(len(arr), dtype=capi.RAY_F64)
    capi.check(lib, lib.kr_memcpy_d2h(out.ctypes.data_as(vp), dptr, out.nbytes), "d2h")
    lib.kr_free(dptr)
    longest = int(np.abs(out["steps"]).max())
    print(f"waves {waves_target:4d} rays {nrays:6d} flagged {best['rays_strict_side']:6d} side {best['strict_side_ms']:8.2f} ms  longest ray {longest} steps  -> {1e3 * best['strict_side_ms'] / longest:.3f} us/step", flush=True)
