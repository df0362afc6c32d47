#!/usr/bin/env python3
"""A/B harness for trace-kernel variants (GPU box).  Variants are separate builds of the same library
(libkrtrace_<tag>.so, made by `python scripts/ab_kernels.py --build` in the build container); all are loaded into
ONE process and timed in interleaved rounds on the same device-resident rays (guide rule 24).
usage: scripts/ab_kernels.py [--build] [--rays 2e6] [--rounds 5] [--integrator rk4] tag[:flag,flag...] ..."""
import argparse, ctypes as C, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--rays", type=float, default=2e6)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--integrator", default="rk4")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()

from raytrace_cpu_amd import _build, capi
specs = []
for v in a.variants:
    tag, _, fl = v.partition(":")          # tag[@kr_flags][:compile,flags]
    specs.append((tag, [f for f in fl.split(",") if f]))
if a.build:
    for tag, fl in specs:
        print(_build.build(extra_flags=fl, tag=("" if tag.partition("@")[0] == "base" else tag.partition("@")[0])))
    sys.exit(0)

import bench
libs = {}
kr_flags = {}
for tag, _ in specs:
    name, _, fl = tag.partition("@")
    path = capi.LIB_PATH if name == "base" else capi.LIB_PATH.replace(".so", f"_{name}.so")
    libs[tag] = capi.load(path)
    kr_flags[tag] = int(fl or os.environ.get("KR_FLAGS", "0"))
method = {"euler": capi.EULER, "rk4": capi.RK4, "rk45": capi.RK45}[a.integrator]
spec = bench.make_spec(capi, bench.grid_spacing_for(a.rays))
first = libs[specs[0][0]]
n = first.kr_pointsource_count(C.byref(spec), None, None)
d_rays = C.c_void_p()
capi.check(first, first.kr_malloc(C.byref(d_rays), n * 144), "malloc")
p = capi.default_params(bench.SPIN); p.integrator, p.r_max = method, bench.R_MAX
times = {t: [] for t, _ in specs}; steps = {}; sides = {}
for rnd in range(a.rounds + 1):
    for tag, _ in specs:
        lib = libs[tag]
        capi.check(lib, lib.kr_pointsource_init_dev_f64(C.byref(spec), d_rays, n, None), "init")
        capi.check(lib, lib.kr_redshift_start_dev_f64(bench.SPIN, 0.0, 0, 0, d_rays, n, None), "rs")
        st = capi.Stats()
        p.flags = kr_flags[tag]
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, n, None, C.byref(st)), "trace")
        if rnd > 0:
            times[tag].append(st.kernel_ms)
            sides.setdefault(tag, []).append((st.strict_side_ms, st.main_ms))
        steps[tag] = st.steps_total
base = np.median(times[specs[0][0]])
for tag, fl in specs:
    t = np.array(times[tag])
    print(json.dumps({"variant": tag, "flags": fl, "rays": int(n), "steps": int(steps[tag]), "kr_flags": kr_flags[tag], "kernel_ms_median": float(np.median(t)), "kernel_ms_min": float(t.min()),
                      "steps_per_sec": steps[tag] / (np.median(t) * 1e-3), "speedup_vs_first": float(base / np.median(t)),
                      "strict_side_ms_median": float(np.median([x[0] for x in sides[tag]])), "main_ms_median": float(np.median([x[1] for x in sides[tag]]))}))
