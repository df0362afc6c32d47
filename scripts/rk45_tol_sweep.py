#!/usr/bin/env python3
"""BASELINE configs[2]: adaptive RK45 tolerance sweep (reference src/tests/emissivity_rk45_tol_sweep.py:38 tolerances,
src/tests/emissivity_rk45_plot.cpp:35-38 grid 0.01 x 0.01 = 125 863 allocated rays), source h = 5 (the reference's
sweep) and h = 10 (BASELINE), on one MI355X through the C ABI.  Prints one JSON document.
RK4 on the same grid is run beside it so the RK4-vs-RK45 emissivity deviation the reference's sweep plots can be formed.
After the point-by-point pass, all 18 points (2 heights x 9 tolerances) are run AT ONCE, device-resident, one ray buffer each, as ONE
merged batch on one stream (kr_trace_batch_async_f64: one strict side launch and one main launch over all 18 traces): their long-ray
tails run side by side, so the whole sweep should cost little more than its slowest point ("concurrent" in the output).
usage: rk45_tol_sweep.py [hybrid|strict|fast]"""
import ctypes as C, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from raytrace_cpu_amd import api, capi


class gc:                                   # the constants this sweep shares with the benchmark
    SPIN = bench.SPIN

    @staticmethod
    def emis_bins(spec, nr):
        n_primary = int(((spec.cosalphamax - spec.cosalpha0) / spec.dcosalpha) * ((spec.betamax - spec.beta0) / spec.dbeta))
        b = bench.emis_bins(capi, api.lib().kr_kerr_isco(bench.SPIN, 1), n_primary)
        b.dr = math.exp(math.log(bench.R_DISC / b.r_min) / nr)
        b.nr = nr
        return b

ARITH = {"hybrid": capi.FLAG_HYBRID, "strict": 0, "fast": capi.FLAG_FAST_MATH}[sys.argv[1] if len(sys.argv) > 1 else "hybrid"]
TOLS = [1e-6, 3e-7, 1e-7, 3e-8, 1e-8, 3e-9, 1e-9, 3e-10, 1e-10]
out = {"arithmetic": sys.argv[1] if len(sys.argv) > 1 else "hybrid", "grid": "dcosalpha = dbeta = 0.01, cos(alpha) in [-0.995, 0.995), beta in [-pi, pi)", "device": api.device_info(), "runs": []}
for h in (5.0, 10.0):
    spec = capi.PointSourceSpec()
    for i, v in enumerate([0.0, h, 1e-3, 0.0]): spec.pos[i] = v
    spec.V, spec.spin, spec.tol, spec.E = 0.0, gc.SPIN, 100.0, 1.0
    spec.cosalpha0, spec.cosalphamax, spec.dcosalpha = -0.995, 0.995, 0.01
    spec.beta0, spec.betamax, spec.dbeta = -math.pi, math.pi, 0.01
    init = api.pointsource_init(spec)
    api.redshift_start(gc.SPIN, 0.0, 0, 0, init)
    bins = gc.emis_bins(spec, nr=30)
    def run(method, tol):
        p = capi.default_params(gc.SPIN); p.integrator, p.rk45_tol, p.flags = method, tol, ARITH
        best = None
        for _ in range(2):
            rays, st = api.trace(p, init)
            if best is None or st["kernel_ms"] < best[1]["kernel_ms"]: best = (rays, st)
        rays, st = best
        api.range_phi(rays); api.redshift(gc.SPIN, -1.0, 0, 0, rays)
        return st, api.reduce_emissivity(bins, rays)
    st4, h4 = run(capi.RK4, 1e-8)
    out["runs"].append({"h": h, "integrator": "rk4", "rays": st4["rays_traced"], "steps": st4["steps_total"], "kernel_ms": st4["kernel_ms"], "steps_per_sec": st4["steps_total"] / st4["kernel_ms"] * 1e3})
    for tol in TOLS:
        st, hh = run(capi.RK45, tol)
        ok = (h4["count"] >= 100) & (hh["count"] >= 100)
        dev = np.abs(hh["emis"][ok] / h4["emis"][ok] - 1)
        out["runs"].append({"h": h, "integrator": "rk45", "tol": tol, "rays": st["rays_traced"], "steps": st["steps_total"], "attempts": st["rk45_attempts"], "rejects": st["rk45_rejects"],
                            "stationary_steps": st["rk45_stationary_steps"], "extrapolated_steps": st["rk45_extrapolated_steps"], "kernel_ms": st["kernel_ms"], "steps_per_sec": st["steps_total"] / st["kernel_ms"] * 1e3,
                            "attempts_per_sec": st["rk45_attempts"] / st["kernel_ms"] * 1e3, "emis_dev_vs_rk4_rms": float(np.sqrt(np.mean(dev ** 2))), "emis_dev_vs_rk4_max": float(dev.max())})

# ---- the same 18 points, all in flight together -------------------------------------------------------------------------
lib = api.lib()
vp = C.c_void_p
ONE_STREAM = vp()
capi.check(lib, lib.kr_stream_create(C.byref(ONE_STREAM)), "stream")      # a merged batch wants ONE stream (more streams only add queues to schedule)
points = []
for h in (5.0, 10.0):
    spec = capi.PointSourceSpec()
    for i, v in enumerate([0.0, h, 1e-3, 0.0]): spec.pos[i] = v
    spec.V, spec.spin, spec.tol, spec.E = 0.0, gc.SPIN, 100.0, 1.0
    spec.cosalpha0, spec.cosalphamax, spec.dcosalpha = -0.995, 0.995, 0.01
    spec.beta0, spec.betamax, spec.dbeta = -math.pi, math.pi, 0.01
    n = api.pointsource_count(spec)[0]
    bins = gc.emis_bins(spec, nr=30)
    for tol in TOLS:
        d_rays, d_hist = vp(), vp()
        capi.check(lib, lib.kr_malloc(C.byref(d_rays), n * capi.RAY_F64.itemsize), "malloc")
        capi.check(lib, lib.kr_malloc(C.byref(d_hist), (5 * bins.nr + 1) * 8), "malloc")
        stream = ONE_STREAM
        p = capi.default_params(gc.SPIN); p.integrator, p.rk45_tol, p.flags = capi.RK45, tol, ARITH
        points.append(dict(h=h, tol=tol, spec=spec, n=n, bins=bins, d_rays=d_rays, d_hist=d_hist, stream=stream, p=p))

def post(pt):
    capi.check(lib, lib.kr_post_emissivity_dev_f64(gc.SPIN, -1.0, 0, 0, 0, -math.pi, math.pi, C.byref(pt["bins"]), pt["d_rays"], pt["n"], pt["d_hist"], pt["stream"]), "post")

walls = []
for rnd in range(3):
    for pt in points:
        capi.check(lib, lib.kr_memset(pt["d_hist"], 0, (5 * pt["bins"].nr + 1) * 8), "memset")
    capi.check(lib, lib.kr_synchronize(None), "sync")
    t0 = time.perf_counter()
    for pt in points:
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(pt["spec"]), 0, 1, 0.0, 0, 0, pt["d_rays"], pt["n"], pt["stream"]), "init")
    # one batch: the strict side launches of all 18 points go to the device before any main launch (kr_trace_batch_async_f64)
    tickets = api.trace_batch_async([pt["p"] for pt in points], [pt["d_rays"].value for pt in points], [pt["n"] for pt in points], [pt["stream"].value for pt in points])
    for pt in points:
        post(pt)
    stats = [api.trace_wait(t) for t in tickets]
    capi.check(lib, lib.kr_synchronize(ONE_STREAM), "sync")
    walls.append(1e3 * (time.perf_counter() - t0))
serial = {(r["h"], r["tol"]): r["kernel_ms"] for r in out["runs"] if r["integrator"] == "rk45"}
hist_ok = True
for pt, st in zip(points, stats):
    hh = np.zeros(5 * pt["bins"].nr + 1)
    capi.check(lib, lib.kr_memcpy_d2h(hh.ctypes.data_as(vp), pt["d_hist"], hh.nbytes), "d2h")
    hist_ok = hist_ok and hh[-1] > 0 and st["steps_total"] == [r for r in out["runs"] if r["integrator"] == "rk45" and r["h"] == pt["h"] and r["tol"] == pt["tol"]][0]["steps"]
out["concurrent"] = {"points": len(points), "wall_ms_rounds": walls, "wall_ms": min(walls), "slowest_single_point_ms": max(serial.values()), "sum_of_single_points_ms": sum(serial.values()),
                     "wall_over_slowest_point": min(walls) / max(serial.values()), "per_point_span_ms": [st["kernel_ms"] for st in stats],
                     "per_point_strict_side_ms": [st["strict_side_ms"] for st in stats], "per_point_main_ms": [st["main_ms"] for st in stats],
                     "same_step_totals_as_point_by_point": bool(hist_ok)}
for pt in points:
    lib.kr_free(pt["d_rays"]); lib.kr_free(pt["d_hist"])
lib.kr_stream_destroy(ONE_STREAM)
print(json.dumps(out, indent=1))
