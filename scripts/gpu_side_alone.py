#!/usr/bin/env python3
"""GPU box: the strict side launch of the headline source WITHOUT a main launch beside it (only the beta = -pi column is traced, as a batch of one):
how much of the side launch's duration in the full trace is interference from the main launch's waves on the same CUs?"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from raytrace_cpu_amd import api, capi
lib = api.lib()
method = {"euler": capi.EULER, "rk4": capi.RK4, "rk45": capi.RK45}[sys.argv[1] if len(sys.argv) > 1 else "rk45"]
full = bench.make_spec(capi, bench.grid_spacing_for(1e7))
spec = bench.make_spec(capi, bench.grid_spacing_for(1e7))
spec.beta0, spec.betamax, spec.dbeta = -math.pi, -math.pi + 1e-9, 1.0          # one column: the rays of the side launch
n = api.pointsource_count(spec)[0]
d = C.c_void_p()
capi.check(lib, lib.kr_malloc(C.byref(d), n * 144), "malloc")
p = capi.default_params(bench.SPIN); p.integrator, p.r_max = method, bench.R_MAX
for i in range(4):
    capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d, n, None), "init")
    t = api.trace_batch_async([p], [d.value], [n], None)
    st = api.trace_wait(t[0])
    print(f"column alone ({n} rays): kernel_ms {st['kernel_ms']:.1f} strict_side {st['strict_side_ms']:.1f} main {st['main_ms']:.1f} strict rays {st['rays_strict_side']} steps {st['steps_total']}", flush=True)
