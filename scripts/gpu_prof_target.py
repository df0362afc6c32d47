#!/usr/bin/env python3
"""GPU box: a workload for `rocprofv3 --pc-sampling-*` (scripts/pc_sample.sh): either the launch's critical ray ALONE on the strict side launch
(mode crawler: the longest ray of the beta = -pi column through KR_FLAG_HYBRID, i.e. classify -> HOG strict kernel, repeated), or a whole
1e7-ray launch in one arithmetic mode (mode main), repeated.  usage: gpu_pcs_target.py crawler|main rk4|rk45|euler [strict|fast|hybrid] [reps]"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from raytrace_cpu_amd import api, capi

mode = sys.argv[1]
integ = {"rk4": capi.RK4, "euler": capi.EULER, "rk45": capi.RK45}[sys.argv[2]]
arith = sys.argv[3] if len(sys.argv) > 3 else "strict"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
flags = {"strict": 0, "fast": capi.FLAG_FAST_MATH, "hybrid": capi.FLAG_HYBRID}[arith]
p = capi.default_params(bench.SPIN)
p.integrator, p.r_max = integ, bench.R_MAX
if mode == "crawler":
    spec = bench.make_spec(capi, bench.grid_spacing_for(1e7))
    spec.beta0, spec.betamax, spec.dbeta = -math.pi, -math.pi + 1e-9, 1.0          # one column
    init = api.pointsource_init(spec)
    api.redshift_start(bench.SPIN, 0.0, 0, 0, init)
    p.flags = 0
    full, st = api.trace(p, init.copy())
    longest = int(np.argmax(np.abs(full["steps"].astype(np.int64))))
    one = init[longest:longest + 1].copy()
    p.flags = capi.FLAG_HYBRID                      # flagged ray -> the HOG strict instance on a wave of its own
    t0 = time.time()
    for _ in range(reps):
        out, st = api.trace(p, one.copy())
    print(json.dumps({"mode": mode, "steps": int(st["longest_ray_steps"]), "strict_side_rays": int(st["rays_strict_side"]), "kernel_ms": st["kernel_ms"],
                      "us_per_step": 1e3 * st["kernel_ms"] / max(1, st["longest_ray_steps"]), "wall_s": time.time() - t0}), flush=True)
else:
    import ctypes as C
    lib = api.lib()
    spec = bench.make_spec(capi, bench.grid_spacing_for(1e7))
    n = api.pointsource_count(spec)[0]
    d = C.c_void_p()
    capi.check(lib, lib.kr_malloc(C.byref(d), n * 144), "malloc")
    p.flags = flags
    for _ in range(reps):
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d, n, None), "init")
        st = api.trace_dev(p, d.value, n)
    print(json.dumps({"mode": mode, "arith": arith, "kernel_ms": st["kernel_ms"], "main_ms": st.get("main_ms"), "side_ms": st.get("strict_side_ms"), "steps": int(st["steps_total"])}), flush=True)
    lib.kr_free(d)
