import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bench
from raytrace_cpu_amd import api, capi
from test_gpu_concurrent import DeviceRays
lib = api.lib()
specs = [bench.make_spec(capi, 0.02), bench.make_spec(capi, 0.013), bench.make_spec(capi, 0.03), bench.make_spec(capi, 0.017)]
specs[1].pos[1], specs[2].pos[1] = 5.0, 20.0
for method, flags in ((capi.RK4, capi.FLAG_HYBRID), (capi.RK45, 0), (capi.EULER, capi.FLAG_HYBRID)):
    bufs = [DeviceRays(lib, s) for s in specs]
    params = []
    for i in range(len(specs)):
        p = capi.default_params(bench.SPIN); p.integrator, p.r_max, p.flags, p.rk45_tol = method, bench.R_MAX, flags, [1e-6, 1e-8, 1e-7, 1e-9][i]
        params.append(p)
    want = []
    for b, p in zip(bufs, params):
        b.init(); api.trace_dev(p, b.d.value, b.n); want.append(b.fetch())
    for b in bufs: b.init()
    tickets = api.trace_batch_async(params, [b.d.value for b in bufs], [b.n for b in bufs], None)
    stats = [api.trace_wait(t) for t in tickets]
    for i, (b, w) in enumerate(zip(bufs, want)):
        g = b.fetch()
        diff = np.zeros(len(g), dtype=bool)
        for f in g.dtype.names:
            diff |= (g[f].view(np.int64) != w[f].view(np.int64)) if g[f].dtype.kind == "f" else (g[f] != w[f])
        idx = np.flatnonzero(diff)
        print("method", method, "trace", i, "n", b.n, "differ", len(idx), "first", idx[:5], "untraced (steps==0) in batch:", int(((g["steps"] == 0)).sum()), "in single:", int((w["steps"] == 0).sum()),
              "flagged", stats[i]["rays_strict_side"], "traced", stats[i]["rays_traced"])
        if len(idx):
            j = idx[0]; print("   got ", [g[f][j] for f in ("r", "theta", "steps", "status")], " want", [w[f][j] for f in ("r", "theta", "steps", "status")], "beta", w["beta"][j])
    for b in bufs: b.free()
