#!/usr/bin/env python3
"""GPU box: does a strict side launch slow down when more of them run at once?  K copies of one RK45 sweep point (h = 5, 125 171 rays,
hybrid) in one kr_trace_batch_async_f64 call, K = 1, 2, 4, 8, 16, 24: per-copy strict_side_ms / main_ms and the wall time."""
import ctypes as C, json, math, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from raytrace_cpu_amd import api, capi
lib = api.lib(); vp = C.c_void_p
spec = capi.PointSourceSpec()
for i, v in enumerate([0.0, 5.0, 1e-3, 0.0]): spec.pos[i] = v
spec.V, spec.spin, spec.tol, spec.E = 0.0, bench.SPIN, 100.0, 1.0
spec.cosalpha0, spec.cosalphamax, spec.dcosalpha = -0.995, 0.995, 0.01
spec.beta0, spec.betamax, spec.dbeta = -math.pi, math.pi, 0.01
n = api.pointsource_count(spec)[0]
flags = {"hybrid": capi.FLAG_HYBRID, "strict": 0}[sys.argv[1] if len(sys.argv) > 1 else "hybrid"]
ONE_STREAM = len(sys.argv) > 4 and sys.argv[4] == "one-stream"      # every copy on the same stream (what a merged batch wants)
p = capi.default_params(bench.SPIN); p.integrator, p.flags = capi.RK45, flags
KMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 24
bufs, streams = [], []
for _ in range(KMAX):
    d, s = vp(), vp()
    capi.check(lib, lib.kr_malloc(C.byref(d), n * 144), "malloc")
    if not (ONE_STREAM and streams): capi.check(lib, lib.kr_stream_create(C.byref(s)), "stream")
    else: s = streams[0]
    bufs.append(d); streams.append(s)
out = {}
for K in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,4,8,16,24,1".split(","))]:
    best = None
    for rep in range(2):
        for d, s in zip(bufs[:K], streams[:K]):
            capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d, n, s), "init")
        for s in streams[:K]: capi.check(lib, lib.kr_synchronize(s), "sync")
        t0 = time.perf_counter()
        tickets = api.trace_batch_async([p] * K, [d.value for d in bufs[:K]], [n] * K, [s.value for s in streams[:K]])
        st = [api.trace_wait(t) for t in tickets]
        wall = 1e3 * (time.perf_counter() - t0)
        if best is None or wall < best[0]: best = (wall, st)
    wall, st = best
    print(K, "wall %.0f ms" % wall, "side", [round(x["strict_side_ms"]) for x in st], "main", [round(x["main_ms"]) for x in st], "flagged", st[0]["rays_strict_side"], flush=True)
