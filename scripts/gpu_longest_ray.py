#!/usr/bin/env python3
"""GPU box: which ray of the headline grid is the longest, per integrator, and does kr_stats agree?  usage: scripts/gpu_longest_ray.py [rays=1e7]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from raytrace_cpu_amd import api, capi

rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
lib = api.lib()
spec = bench.make_spec(capi, bench.grid_spacing_for(rays_n))
n, n_alpha, n_beta = api.pointsource_count(spec)
d = C.c_void_p()
capi.check(lib, lib.kr_malloc(C.byref(d), n * 144), "malloc")
host = np.zeros(n, dtype=capi.RAY_F64)
for integ, name in ((capi.RK4, "rk4"), (capi.EULER, "euler")):
    for flags, mode in ((capi.FLAG_HYBRID, "hybrid"), (0, "strict")):
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d, n, None), "init")
        p = capi.default_params(bench.SPIN)
        p.integrator, p.r_max, p.flags = integ, bench.R_MAX, flags
        st = api.trace_dev(p, d.value, n)
        capi.check(lib, lib.kr_memcpy_d2h(host.ctypes.data_as(C.c_void_p), d, host.nbytes), "d2h")
        steps = np.abs(host["steps"].astype(np.int64)); steps[host["steps"] == -1] = 0
        top = np.argsort(steps)[-5:][::-1]
        print(json.dumps({"integrator": name, "mode": mode, "kernel_ms": round(st["kernel_ms"], 2), "strict_side_ms": round(st["strict_side_ms"], 2), "main_ms": round(st["main_ms"], 2),
                          "stats_longest": st["longest_ray_steps"], "stats_longest_side": st["longest_ray_steps_strict_side"], "rays_strict_side": st["rays_strict_side"],
                          "top5": [{"ray": int(i), "row": int(i // n_beta), "col": int(i % n_beta), "steps": int(steps[i]), "status": int(host["status"][i]),
                                    "r": float(host["r"][i]), "theta": float(host["theta"][i])} for i in top]}), flush=True)
