"""GPU: the hybrid launch away from the BASELINE geometry -- a reduced form of tests/tool_gpu_hybrid_sweep.py that runs with the suite.
Five sources (lamp posts at three spins / heights, an orbiting disc source, an image plane at low inclination) x ~2.5e5 rays x {Euler, RK4},
theta-limit overload, against the oracle (bit-identical to the compiled reference, tests/test_oracle_vs_ref.py): the integer outcome of EVERY
ray -- status, step count, turning points, equatorial crossings -- must be the reference's, but for at most 2 rays per run (a ray that ends an
ulp from the equatorial plane is decided by the fast arithmetic's last bit: 1 in 1e6 on one geometry, profiles/r03_hybrid_sweep_euler.jsonl).
Round 3's first sweep found 5-23 rays per 1e6 with one step too many on four of these geometries (the clipped last step):
none of the fixtures -- all on the BASELINE geometry then -- could see it."""
import math

import numpy as np
import pytest

import oracle_lib as ol
from raytrace_cpu_amd import api, capi

pytestmark = pytest.mark.gpu
RAYS = 2.5e5


def _sources():
    d = 1.99 / (math.sqrt(RAYS) - 1.0)
    out = []
    for spin, pos, V, tag in [(0.0, [0, 10, 1e-3, 0.0], 0.0, "lamp_h10_a0"), (0.5, [0, 5, 1e-3, 0.0], 0.0, "lamp_h5_a05"), (0.998, [0, 3, 1e-3, 0.0], 0.0, "lamp_h3"),
                              (0.998, [0, 6, math.pi / 2 - 1e-6, 1.5707], None, "orbiting_r6")]:
        if V is None:
            V = 1.0 / (spin + pos[1] ** 1.5)          # disc_velocity(r, a, +1), src/include/kerr.h:35-38 (nothing of libkrtrace is touched at collection time)
        out.append(("ps", tag, ol.pointsource_spec(pos, V, spin, d, d * math.pi / 0.995, cosalpha0=-0.995, cosalphamax=0.995, beta0=-math.pi, betamax=math.pi), spin, V))
    n = int(math.sqrt(RAYS)) | 1
    out.append(("ip", "image_incl5_a09", ol.imageplane_spec(10000.0, 5.0, -30.0, 30.0, 60.0 / n, -30.0, 30.0, 60.0 / n, 0.9), 0.9, 0.0))
    return out


@pytest.mark.parametrize("method", [capi.EULER, capi.RK4], ids=["euler", "rk4"])
@pytest.mark.parametrize("source", _sources(), ids=lambda s: s[1])
def test_hybrid_launch_reproduces_integer_outcomes_away_from_baseline(krlib, source, method):
    kind, tag, spec, spin, V = source
    if kind == "ps":
        init = ol.oracle_pointsource(spec)
        ol.oracle().kro_redshift_start_f64(spin, V, 0, 0, ol.ptr(init), len(init))
        p = capi.default_params(spin)
        p.integrator, p.r_max = method, 1000.0
    else:
        init = ol.oracle_imageplane(spec)
        ol.oracle().kro_redshift_start_f64(-spin, 0.0, 1, 0, ol.ptr(init), len(init))
        p = capi.default_params(-spin)
        p.integrator, p.r_max = method, 11000.0
    want, _ = ol.oracle_trace(p, init)
    valid = want["steps"] != -1
    assert valid.sum() > 2e5
    for flags, mode, allowed in ((capi.FLAG_HYBRID, "hybrid", 2), (0, "strict", 0)):
        got, st = api.trace(capi.copy_params(p, flags=flags), init)
        differ = np.zeros(len(init), dtype=bool)
        for k in ("status", "steps", "rdot_flips", "equatorial_crossings"):
            differ |= got[k] != want[k]
        n_bad = int((valid & differ).sum())
        assert n_bad <= allowed, (tag, mode, n_bad, np.flatnonzero(valid & differ)[:10].tolist())
