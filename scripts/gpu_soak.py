#!/usr/bin/env python3
"""GPU box: many passes in one process -- does a pass get slower, does device memory grow?  usage: scripts/gpu_soak.py [passes=300]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import bench
from raytrace_cpu_amd import api, capi

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lib = api.lib()
spec = bench.make_spec(capi, bench.grid_spacing_for(1e7))
n = api.pointsource_count(spec)[0]
d = C.c_void_p()
capi.check(lib, lib.kr_malloc(C.byref(d), n * 144), "malloc")
p = capi.default_params(bench.SPIN)
p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, capi.FLAG_HYBRID
free0 = torch.cuda.mem_get_info()[0]
ms, t0 = [], time.perf_counter()
streams = []
for k in range(passes):
    s = C.c_void_p()
    if k % 3 == 0:                                   # every third pass on a fresh caller stream that is destroyed afterwards
        capi.check(lib, lib.kr_stream_create(C.byref(s)), "stream")
    capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d, n, s), "init")
    st = api.trace_dev(p, d.value, n, stream=s.value)
    ms.append(st["kernel_ms"])
    if s:
        capi.check(lib, lib.kr_stream_destroy(s), "stream destroy")
    if k in (0, passes // 2, passes - 1):
        print(json.dumps({"pass": k, "kernel_ms": round(st["kernel_ms"], 2), "free_bytes_delta": int(torch.cuda.mem_get_info()[0] - free0), "steps": st["steps_total"]}), flush=True)
import numpy as np
print(json.dumps({"passes": passes, "wall_s": round(time.perf_counter() - t0, 1), "kernel_ms_first10": round(float(np.mean(ms[:10])), 2), "kernel_ms_last10": round(float(np.mean(ms[-10:])), 2),
                  "kernel_ms_max": round(float(np.max(ms)), 2)}))
capi.check(lib, lib.kr_shutdown(), "shutdown")
print(json.dumps({"after_kr_shutdown_free_bytes_delta": int(torch.cuda.mem_get_info()[0] - free0)}))
