#!/usr/bin/env python3
"""GPU box tool (not a pytest file): WHICH rays of a theta-limit Euler / RK4 launch come out with other integer fields under the hybrid launch's
fast arithmetic, and by how much?  Lamp post h = 10, a = 0 (profiles/r03_hybrid_sweep_euler.jsonl: 23 of 1e6) and h = 5, a = 0.5.
usage: python tests/tool_gpu_euler_diff.py [rays=1e6] [euler|rk4] [hybrid|fast]"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from raytrace_cpu_amd import api, capi

rays_n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
method = {"euler": capi.EULER, "rk4": capi.RK4}[sys.argv[2] if len(sys.argv) > 2 else "euler"]
mode = sys.argv[3] if len(sys.argv) > 3 else "hybrid"
flags = {"hybrid": capi.FLAG_HYBRID, "fast": capi.FLAG_FAST_MATH}[mode]
d = 1.99 / (math.sqrt(rays_n) - 1.0)
for spin, pos, tag in [(0.0, [0, 10, 1e-3, 0.0], "h10_a0"), (0.5, [0, 5, 1e-3, 0.0], "h5_a0.5"), (0.998, [0, 3, 1e-3, 0.0], "h3")]:
    spec = ol.pointsource_spec(pos, 0.0, spin, d, d * math.pi / 0.995, cosalpha0=-0.995, cosalphamax=0.995, beta0=-math.pi, betamax=math.pi)
    src = ol.RefSource(spec)
    src.lib.ref_redshift_start(src.h, 0.0, 0, 0)
    p = capi.default_params(spin)
    p.integrator, p.r_max = method, 1000.0
    init = src.snapshot()
    src.run(p)
    want = src.snapshot()
    src.close()
    got, st = api.trace(capi.copy_params(p, flags=flags), init)
    strict, _ = api.trace(capi.copy_params(p, flags=0), init)
    valid = want["steps"] != -1
    ints = np.zeros(len(init), dtype=bool)
    per_field = {}
    for k in ("status", "steps", "rdot_flips", "equatorial_crossings", "rdot_sign", "thetadot_sign"):
        m = valid & (got[k] != want[k])
        per_field[k] = int(m.sum())
        ints |= m
    idx = np.flatnonzero(ints)
    rows = []
    for i in idx[:40]:
        rows.append({"ray": int(i), "col": int(i % int(round(2 * math.pi / (d * math.pi / 0.995)))), "want": {k: int(want[k][i]) for k in ("status", "steps", "rdot_flips", "equatorial_crossings")},
                     "got": {k: int(got[k][i]) for k in ("status", "steps", "rdot_flips", "equatorial_crossings")}, "strict_equals_ref": bool(all(strict[k][i] == want[k][i] for k in ("status", "steps", "rdot_flips", "equatorial_crossings"))),
                     "r_want": float(want["r"][i]), "r_got": float(got["r"][i]), "theta_want": float(want["theta"][i]), "theta_got": float(got["theta"][i]), "h": float(init["h"][i]), "Q": float(init["Q"][i])})
    print(json.dumps({"config": tag, "mode": mode, "rays": int(valid.sum()), "differ": int(ints.sum()), "per_field": per_field, "rows": rows}), flush=True)
