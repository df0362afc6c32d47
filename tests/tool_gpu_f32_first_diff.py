#!/usr/bin/env python3
"""GPU box tool (not a pytest file): where does the float kernel first leave the reference's float build?  Traces the ps_h10 float fixture
with steplim = 1, 2, 3, ... on both sides and reports, for the rays that end up different, the first step and the fields that differ there."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import golden_cases as gc
import make_golden_f32 as mg
from raytrace_cpu_amd import api, capi

method = sys.argv[1] if len(sys.argv) > 1 else "euler"
L = mg.lib()
case = gc.cases()["ps_h10"]
first = {}
prev_bad = None
for k in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 128):
    p = capi.copy_params(case["runs"][method], steplim=k)
    init, want, _, _ = mg.run(L, case, p)
    got, _ = api.trace(p, init)
    live = want["steps"] != -1
    diff = {f: live & (got[f].view(np.int32) != want[f].view(np.int32)) & ~(np.isnan(got[f]) & np.isnan(want[f])) for f in ("t", "r", "theta", "phi", "pt", "pr", "ptheta", "pphi")}
    anyd = np.zeros(len(got), dtype=bool)
    for f, m in diff.items():
        anyd |= m
    print(f"steplim {k}: rays differing {int(anyd.sum())} of {int(live.sum())}; by field " + ", ".join(f"{f} {int(m.sum())}" for f, m in diff.items()))
    for i in np.flatnonzero(anyd)[:3]:
        if i not in first:
            first[i] = k
            print(f"   ray {i} first differs at steplim {k}: " + "; ".join(f"{f} {got[f][i]!r}/{want[f][i]!r}" for f in diff if diff[f][i]) + f" | theta0 {init['theta'][i]!r} r0 {init['r'][i]!r} k {init['k'][i]!r} h {init['h'][i]!r} Q {init['Q'][i]!r}")
