#!/usr/bin/env python3
"""GPU box: what slows a strict side launch (waves alone on their SIMDs) when other work runs on the chip?  One RK45 sweep point
(125 171 rays, h = 5: its side launch is ~0.35 s of one polar-axis ray) traced (a) alone, (b) beside an fp64 GEMM loop, (c) beside a
memory-bound elementwise loop, (d) beside another trace's main launch only (a fast-math 1e7-ray RK4 trace)."""
import ctypes as C, json, math, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from raytrace_cpu_amd import api, capi
lib = api.lib()
vp = C.c_void_p
spec = capi.PointSourceSpec()
for i, v in enumerate([0.0, 5.0, 1e-3, 0.0]): spec.pos[i] = v
spec.V, spec.spin, spec.tol, spec.E = 0.0, bench.SPIN, 100.0, 1.0
spec.cosalpha0, spec.cosalphamax, spec.dcosalpha = -0.995, 0.995, 0.01
spec.beta0, spec.betamax, spec.dbeta = -math.pi, math.pi, 0.01
n = api.pointsource_count(spec)[0]
rays = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
p = capi.default_params(bench.SPIN); p.integrator, p.flags = capi.RK45, capi.FLAG_HYBRID
s_trace = torch.cuda.Stream(); s_load = torch.cuda.Stream()

def point():
    capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, vp(rays.data_ptr()), n, vp(s_trace.cuda_stream)), "init")
    return api.trace_async(p, rays.data_ptr(), n, stream=s_trace.cuda_stream)

def run(load):
    out = []
    for _ in range(3):
        torch.cuda.synchronize()
        t = point()
        stop = time.perf_counter() + 0.6
        k = 0
        with torch.cuda.stream(s_load):
            while load is not None and time.perf_counter() < stop:
                load(); k += 1
                if k % 8 == 0: s_load.synchronize()
        st = api.trace_wait(t)
        out.append((round(st["strict_side_ms"], 1), round(st["main_ms"], 1), k))
    return out

A = torch.randn(4096, 4096, dtype=torch.float64, device="cuda"); B = torch.randn(4096, 4096, dtype=torch.float64, device="cuda")
X = torch.randn(1 << 28, dtype=torch.float32, device="cuda")
spec2 = bench.make_spec(capi, bench.grid_spacing_for(1e7)); n2 = api.pointsource_count(spec2)[0]
rays2 = torch.empty(n2 * 144, dtype=torch.uint8, device="cuda")
def other_trace(method):
    p2 = capi.default_params(bench.SPIN); p2.integrator, p2.r_max, p2.flags = method, bench.R_MAX, capi.FLAG_FAST_MATH
    def go():
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec2), 0, 1, 0.0, 0, 0, vp(rays2.data_ptr()), n2, vp(s_load.cuda_stream)), "init")
        api.trace_dev(p2, rays2.data_ptr(), n2, stream=s_load.cuda_stream, want_stats=False)
    return go
# code sizes (llvm-readelf): RK45 strict HOG 38 KB; fast-math main kernels: Euler 6 KB, RK4 12 KB, RK45 33 KB; the instruction cache is 64 KB per 2 CUs
res = {"alone": run(None), "beside fp64 GEMM 4096^3": run(lambda: torch.mm(A, B)), "beside fp32 elementwise (HBM-bound)": run(lambda: X.mul_(1.0001)),
       "beside fast-math Euler traces of 1e7 rays (6 KB of code)": run(other_trace(capi.EULER)),
       "beside fast-math RK4 traces of 1e7 rays (12 KB)": run(other_trace(capi.RK4)),
       "beside fast-math RK45 traces of 1e7 rays (33 KB)": run(other_trace(capi.RK45)), "alone again": run(None)}
print(json.dumps({"what": "(strict_side_ms, main_ms, load kernels issued) of one RK45 h=5 sweep point", **res}, indent=1))
