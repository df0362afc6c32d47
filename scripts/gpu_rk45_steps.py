#!/usr/bin/env python3
"""GPU box: per-call timing of the split RK45 trace (whole call, strict side launch, main launch) over several repeats."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from raytrace_cpu_amd import api, capi
lib = api.lib()
method = {"euler": capi.EULER, "rk4": capi.RK4, "rk45": capi.RK45}[sys.argv[1] if len(sys.argv) > 1 else "rk45"]
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
spec = bench.make_spec(capi, bench.grid_spacing_for(1e7))
n = api.pointsource_count(spec)[0]
d = C.c_void_p()
capi.check(lib, lib.kr_malloc(C.byref(d), n * 144), "malloc")
p = capi.default_params(bench.SPIN); p.integrator, p.r_max, p.flags = method, bench.R_MAX, flags
stream = C.c_void_p()
if len(sys.argv) > 3 and sys.argv[3] == "stream":
    capi.check(lib, lib.kr_stream_create(C.byref(stream)), "stream")
print("stream", stream.value)
for i in range(8):
    capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d, n, stream), "init")
    capi.check(lib, lib.kr_synchronize(stream), "sync")
    t0 = time.perf_counter()
    st = api.trace_dev(p, d.value, n, stream=stream.value)
    wall = (time.perf_counter() - t0) * 1e3
    print(f"call {i}: wall {wall:7.1f} ms  kernel_ms {st['kernel_ms']:7.1f}  strict_side {st['strict_side_ms']:7.1f}  main {st['main_ms']:7.1f}  strict rays {st['rays_strict_side']}", flush=True)
