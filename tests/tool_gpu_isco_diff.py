#!/usr/bin/env python3
"""GPU box tool (not a pytest file): WHICH rays of a run_raytrace(DiscWithISCODestination*) launch come out differently under the
hybrid launch's fast arithmetic?  For the two geometries profiles/r02_hybrid_sweep_rk4_isco.jsonl flags (lamp h=5 a=0.5, lamp h=10 a=0)
the compiled reference and the HIP path trace the same 1e6 rays; the initial records, the reference's results and the HIP results of
every ray that differs (integer fields, or r / theta beyond 1e-9) go to gpurun_out/isco_diff_<tag>.npz for analysis on the CPU.
usage: python tests/tool_gpu_isco_diff.py [rays=1e6] [mode=hybrid|strict|fast]"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from raytrace_cpu_amd import api, capi

rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
mode = sys.argv[2] if len(sys.argv) > 2 else "hybrid"
flags = {"hybrid": capi.FLAG_HYBRID, "strict": 0, "fast": capi.FLAG_FAST_MATH}[mode]
d = 1.99 / (math.sqrt(rays_n) - 1.0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for spin, pos, tag in [(0.5, [0, 5, 1e-3, 0.0], "h5_a0.5"), (0.0, [0, 10, 1e-3, 0.0], "h10_a0")]:
    spec = ol.pointsource_spec(pos, 0.0, spin, d, d * math.pi / 0.995, cosalpha0=-0.995, cosalphamax=0.995, beta0=-math.pi, betamax=math.pi)
    src = ol.RefSource(spec)
    src.lib.ref_redshift_start(src.h, 0.0, 0, 0)
    p = capi.default_params(spin)
    p.integrator, p.r_max = capi.RK4, 1000.0
    p = capi.copy_params(p, stop_kind=capi.STOP_DISC_ISCO, stop_params=(api.lib().kr_kerr_isco(spin, 1), 400.0, math.pi / 2))
    init = src.snapshot()
    src.run(p)
    want = src.snapshot()
    src.close()
    got, st = api.trace(capi.copy_params(p, flags=flags), init)
    valid = want["steps"] != -1
    ints = np.zeros(len(init), dtype=bool)
    for k in ("status", "steps", "rdot_flips", "equatorial_crossings"):
        ints |= got[k] != want[k]
    with np.errstate(invalid="ignore"):
        far = np.zeros(len(init), dtype=bool)
        for k in ("r", "theta"):
            far |= np.abs(got[k] - want[k]) > 1e-9 * np.maximum(np.abs(want[k]), 1e-300)
    bad = valid & (ints | far)
    idx = np.flatnonzero(bad)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"isco_diff_{tag}_{mode}.npz"), idx=idx, init=init[idx], want=want[idx], got=got[idx],
                        ints=ints[idx], status_all=want["status"], steps_all=want["steps"], r_all=want["r"])
    print(json.dumps({"config": tag, "mode": mode, "rays": int(valid.sum()), "integer_fields_differ": int((valid & ints).sum()),
                      "beyond_1e-9": int((valid & ~ints & far).sum()), "kernel_ms": round(st["kernel_ms"], 1)}), flush=True)
