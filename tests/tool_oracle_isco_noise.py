#!/usr/bin/env python3
"""CPU tool (not a pytest file): how well does the REFERENCE ALGORITHM agree with itself on run_raytrace(DiscWithISCODestination*) away
from a = 0.998?  The oracle (bit-identical to the compiled reference, tests/test_oracle_golden.py) traces 1e6 lamp-post rays twice: as
they are, and with the Carter constant Q of every ray moved by ONE ulp.  Rays whose integer outputs (status, steps, rdot_flips,
equatorial_crossings) change, or whose end point moves by more than 1e-9, are rays whose result in the reference is decided at the
rounding level: they cross the equatorial plane inside the ISCO (the destination does not stop them there), whirl near the photon
sphere and come back.  No arithmetic that differs from the reference's by even one ulp per step reproduces them; this is why the class
mirror runs the RayDestination overloads on the strict (bit-carrying) arithmetic by default (raytrace_cpu_amd/host/raytracer/raytracer.cpp).
usage: python tests/tool_oracle_isco_noise.py [rays=1e6] [hybrid-dump.npz ...]   -> one JSON line per geometry (profiles/r03_isco_noise.jsonl)
(optional: gpurun_out/isco_diff_<tag>_hybrid.npz files from tests/tool_gpu_isco_diff.py, to count the overlap with the hybrid launch's set)"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from raytrace_cpu_amd import capi

rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
dumps = {os.path.basename(f).replace("isco_diff_", "").replace("_hybrid.npz", ""): f for f in sys.argv[2:]}
d = 1.99 / (math.sqrt(rays_n) - 1.0)
o = ol.oracle()
for spin, pos, tag in [(0.5, [0, 5, 1e-3, 0.0], "h5_a0.5"), (0.0, [0, 10, 1e-3, 0.0], "h10_a0"), (0.998, [0, 10, 1e-3, 1.5707], "h10_a0.998")]:
    spec = ol.pointsource_spec(pos, 0.0, spin, d, d * math.pi / 0.995, cosalpha0=-0.995, cosalphamax=0.995, beta0=-math.pi, betamax=math.pi)
    init = ol.oracle_pointsource(spec)
    o.kro_redshift_start_f64(spin, 0.0, 0, 0, ol.ptr(init), len(init))
    p = capi.default_params(spin)
    p.integrator, p.r_max = capi.RK4, 1000.0
    p = capi.copy_params(p, stop_kind=capi.STOP_DISC_ISCO, stop_params=(o.kro_kerr_isco(spin, 1), 400.0, math.pi / 2))
    want, _ = ol.oracle_trace(p, init, nthreads=os.cpu_count())
    pert = init.copy()
    pert["Q"] = np.nextafter(pert["Q"], np.inf)
    got, _ = ol.oracle_trace(p, pert, nthreads=os.cpu_count())
    valid = want["steps"] != -1
    ints = np.zeros(len(init), dtype=bool)
    for k in ("status", "steps", "rdot_flips", "equatorial_crossings"):
        ints |= got[k] != want[k]
    far = np.zeros(len(init), dtype=bool)
    with np.errstate(invalid="ignore"):
        for k in ("r", "theta"):
            far |= np.abs(got[k] - want[k]) > 1e-9 * np.maximum(np.abs(want[k]), 1e-300)
    noisy = valid & (ints | far)
    eq, st = want["equatorial_crossings"], want["status"]
    passed_inside = valid & (eq >= 1) & ~(((st & 1) == 1) & (eq == 1))        # crossed the plane at least once without being stopped there
    row = {"config": f"lamp {tag}", "stop": "isco", "integrator": "rk4", "rays": int(valid.sum()), "perturbation": "Q -> nextafter(Q) on every ray, CPU oracle vs CPU oracle",
           "integer_fields_differ": int((valid & ints).sum()), "beyond_1e-9": int((valid & ~ints & far).sum()),
           "rays_that_crossed_the_plane_unstopped": int(passed_inside.sum()), "noisy_rays_among_them": int((noisy & passed_inside).sum())}
    if tag in dumps:
        bad = np.zeros(len(init), dtype=bool)
        bad[np.load(dumps[tag])["idx"]] = True
        row["hybrid_launch_differs_on"] = int(bad.sum())
        row["of_which_also_noisy_under_the_1ulp_test"] = int((bad & noisy).sum())
    print(json.dumps(row), flush=True)
