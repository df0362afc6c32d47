#!/usr/bin/env python3
"""GPU box tool (not a pytest file): how robust is the hybrid launch's ray classification away from the BASELINE geometry?
For a set of sources (heights, spins, off-axis positions, orbiting sources, image planes at several inclinations) the compiled
reference traces ~1e6 rays on the host cores and the HIP path traces the same initial rays in hybrid and in strict mode; per
configuration: rays whose integer outcome (status, steps, flips, crossings) differs, rays beyond 1e-9, strict-side ray count.
usage: python tests/tool_gpu_hybrid_sweep.py [rays=1e6] [rk4|rk45|euler] [theta|flatdisc|isco]   -> one JSON line per configuration
(third argument: the stop surface -- the theta-limit overload, or run_raytrace(RayDestination*) with FlatDiscDestination / DiscWithISCODestination)"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from raytrace_cpu_amd import api, capi

rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
METHOD = {"rk4": capi.RK4, "rk45": capi.RK45, "euler": capi.EULER}[sys.argv[2] if len(sys.argv) > 2 else "rk4"]
STOP = sys.argv[3] if len(sys.argv) > 3 else "theta"
d = 1.99 / (math.sqrt(rays_n) - 1.0)
configs = []
for spin, pos, V, tag in [(0.998, [0, 10, 1e-3, 1.5707], 0.0, "lamp h=10 (BASELINE)"), (0.998, [0, 3, 1e-3, 0.0], 0.0, "lamp h=3"),
                          (0.998, [0, 20, 1e-3, 0.0], 0.0, "lamp h=20"), (0.5, [0, 5, 1e-3, 0.0], 0.0, "lamp h=5 a=0.5"),
                          (0.0, [0, 10, 1e-3, 0.0], 0.0, "lamp h=10 a=0"), (0.998, [0, 8, math.pi / 4, 0.3], 0.0, "off-axis theta=pi/4"),
                          (0.998, [0, 6, math.pi / 2 - 1e-6, 1.5707], None, "orbiting disc source r=6"), (0.9, [0, 4, 1.0, 0.0], 0.05, "off-axis rotating V=0.05")]:
    if V is None:
        V = api.lib().kr_disc_velocity(pos[1], spin, 1)
    configs.append(("ps", tag, ol.pointsource_spec(pos, V, spin, d, d * math.pi / 0.995, cosalpha0=-0.995, cosalphamax=0.995, beta0=-math.pi, betamax=math.pi), spin, V))
N = int(math.sqrt(rays_n)) | 1
for incl, spin in [(80.0, 0.998), (30.0, 0.998), (60.0, 0.5), (5.0, 0.9)]:
    configs.append(("ip", f"image plane incl={incl} a={spin} odd grid", ol.imageplane_spec(10000.0, incl, -30.0, 30.0, 60.0 / N, -30.0, 30.0, 60.0 / N, spin), spin, 0.0))
configs.append(("ip", "image plane incl=80 even grid (pixel at 0,0)", ol.imageplane_spec(10000.0, 80.0, -30.0, 30.0, 60.0 / (N - 1), -30.0, 30.0, 60.0 / (N - 1), 0.998), 0.998, 0.0))

if METHOD == capi.RK45:      # the reference's RK45 spends 1e5 steps on every captured ray: keep the sweep short (and off the (0,0) pixel, where it hangs)
    configs = [c for c in configs if c[1] in ("lamp h=10 (BASELINE)", "lamp h=3", "lamp h=5 a=0.5", "off-axis theta=pi/4", "image plane incl=80.0 a=0.998 odd grid", "image plane incl=30.0 a=0.998 odd grid")]
for kind, tag, spec, spin, V in configs:
    src = ol.RefSource(spec)
    if kind == "ps":
        src.lib.ref_redshift_start(src.h, V, 0, 0)
        p = capi.default_params(spin)
        p.integrator, p.r_max = METHOD, 1000.0
    else:
        src.lib.ref_redshift_start(src.h, 0.0, 1, 0)
        p = capi.default_params(-spin)
        p.integrator, p.r_max = METHOD, 11000.0
    if STOP == "flatdisc":
        p = capi.copy_params(p, stop_kind=capi.STOP_FLATDISC, stop_params=(math.pi / 2,))
    elif STOP == "isco":
        p = capi.copy_params(p, stop_kind=capi.STOP_DISC_ISCO, stop_params=(api.lib().kr_kerr_isco(spin, 1), 400.0 if kind == "ps" else 30.0, math.pi / 2))
    init = src.snapshot()
    t0 = time.perf_counter()
    src.run(p)
    cpu_s = time.perf_counter() - t0
    want = src.snapshot()
    src.close()
    valid = want["steps"] != -1
    # what Raytracer<T>::run_raytrace picks when KRTRACE_ARITHMETIC is unset (host/raytracer/raytracer.cpp::arithmetic_flags)
    row = {"config": tag, "stop": STOP, "rays": int(valid.sum()), "cpu_s": round(cpu_s, 2),
           "class_mirror_default": "strict" if (STOP != "theta" or METHOD == capi.RK45) else "hybrid"}
    modes = (("hybrid", capi.FLAG_HYBRID), ("strict", 0)) + ((("strict_iterate_all", capi.FLAG_RK45_ITERATE_ALL),) if METHOD == capi.RK45 else ())
    for mode, flags in modes:
        got, st = api.trace(capi.copy_params(p, flags=flags), init)
        ints = np.zeros(len(init), dtype=bool)
        for k in ("status", "steps", "rdot_flips", "equatorial_crossings"):
            ints |= got[k] != want[k]
        with np.errstate(invalid="ignore"):
            far = np.zeros(len(init), dtype=bool)
            for k in ("r", "theta"):
                far |= np.abs(got[k] - want[k]) > 1e-9 * np.maximum(np.abs(want[k]), 1e-300)
        nanmis = np.isnan(got["r"]) != np.isnan(want["r"])
        row[mode] = {"integer_fields_differ": int((valid & ints).sum()), "beyond_1e-9": int((valid & ~ints & far).sum()), "nan_mismatch": int((valid & nanmis).sum()),
                     "strict_side": int(st["rays_strict_side"]), "kernel_ms": round(st["kernel_ms"], 1)}
    print(json.dumps(row), flush=True)
