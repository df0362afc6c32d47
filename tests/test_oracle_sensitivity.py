"""CPU: pins the reference algorithm's OWN sensitivity to 1 ulp of input noise, which is what justifies the
per-ray tolerances of tests/parity.py (a GPU libm differs from glibc by about that much)."""
import numpy as np

import golden_cases as gc
import oracle_lib as ol
import parity


def _envelope(run):
    g = np.load(gc.golden_path("ps_h10"))
    init = g["init"]
    p = gc.cases()["ps_h10"]["runs"][run]
    base, _ = ol.oracle_trace(p, init)
    pert = init.copy()
    pert["Q"] = np.nextafter(pert["Q"], np.inf)
    out, _ = ol.oracle_trace(p, pert)
    ok = (base["steps"] > 0) & (base["status"] == 1) & (out["status"] == 1)
    return np.abs(out["r"][ok] - base["r"][ok]) / base["r"][ok]


def test_rk4_noise_envelope():
    rel = _envelope("rk4")
    assert np.median(rel) < 1e-12
    assert (rel > parity.RAY_RTOL).mean() <= 2 * parity.CHAOTIC_FRAC      # photon-sphere rays: O(1) changes
    assert rel.max() > 1e-6                                               # ... they do exist on the CPU too


def test_rk45_noise_envelope():
    rel = _envelope("rk45")
    assert 1e-13 < np.median(rel) < 1e-10          # orders of magnitude above RK4's: the step controller amplifies
    assert (rel > 1e-9).mean() > 0.01              # a 1e-9 band would reject the reference against itself
    assert (rel > parity.RAY_RTOL_RK45).mean() <= parity.CHAOTIC_FRAC
