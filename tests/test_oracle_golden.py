"""CPU: the oracle restatement (oracle/kr_oracle.c) reproduces the golden fixtures captured from the compiled
reference BIT FOR BIT -- every field of every ray, for every integrator / stop surface / source kind."""
import ctypes as C

import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as ol
from raytrace_cpu_amd import capi

CASES = gc.cases()
RUNS = [(c, r) for c in CASES for r in CASES[c]["runs"]]


def oracle_pipeline(case, params):
    """ctor -> redshift_start -> run_raytrace -> range_phi -> redshift, all through the oracle."""
    o = ol.oracle()
    src = case["source"]
    rays = ol.oracle_imageplane(src) if gc.is_imageplane(case) else ol.oracle_pointsource(src)
    V, rev, proj = case["start"]
    # Raytracer::redshift_start physical spin argument: the tracer's stored spin (negated for ImagePlane)
    o.kro_redshift_start_f64(params.spin, V, rev, proj, ol.ptr(rays), len(rays))
    init = rays.copy()
    out, st = ol.oracle_trace(params, rays)
    o.kro_range_phi_f64(-np.pi, np.pi, ol.ptr(out), len(out))
    V, rev, proj = case["post"]
    if params.stop_kind == capi.STOP_THETA:
        o.kro_redshift_f64(params.spin, V, rev, proj, 0, ol.ptr(out), len(out))
    else:
        o.kro_redshift_dest_f64(params.spin, rev, ol.ptr(out), len(out))
    return init, out, st


@pytest.mark.parametrize("case_name", list(CASES))
def test_init_matches_reference(case_name):
    case = CASES[case_name]
    g = np.load(gc.golden_path(case_name))
    params = next(iter(case["runs"].values()))
    init, _, _ = oracle_pipeline(case, capi.copy_params(params, steplim=1))
    assert ol.rays_equal_bitwise(g["init"], init) == []


@pytest.mark.parametrize("case_name,run", RUNS)
def test_trace_matches_reference(case_name, run):
    case = CASES[case_name]
    g = np.load(gc.golden_path(case_name))
    _, out, st = oracle_pipeline(case, case["runs"][run])
    assert ol.rays_equal_bitwise(g[f"final__{run}"], out) == []
    assert st["steps_total"] == int(g[f"steps__{run}"])


def test_constants_match_survey_spot_values():
    # SURVEY.md section 8c spot values (reference compiled with -ffp-contract=off)
    o = ol.oracle()
    assert o.kro_kerr_horizon(0.998) == 1.0632139225171164
    assert o.kro_kerr_isco(0.998, 1) == 1.2369706630706787
    spec = ol.pointsource_spec([0, 10, 1e-3, 1.5707], 0.0, 0.998, 0.05, 0.05, cosalpha0=-0.995, cosalphamax=0.995,
                               beta0=-np.pi, betamax=np.pi)
    rays = ol.oracle_pointsource(spec)
    assert len(rays) == 5167
    assert rays[777]["k"] == 0.89552909569197359
    assert rays[777]["h"] == -0.003595576665977989
    assert rays[777]["Q"] == 51.415215947851898
    expect = {capi.EULER: (7.2041480467226258, 345), capi.RK4: (7.1770693848828495, 347), capi.RK45: (7.1967674856265065, 89)}
    for method, (r, steps) in expect.items():
        p = capi.default_params(0.998)
        p.integrator = method
        out, _ = ol.oracle_trace(p, rays)
        assert out[777]["r"] == r and out[777]["steps"] == steps and out[777]["status"] == 1
        if method == capi.RK45:   # rays 0,1 head straight for the horizon and burn RK45_STEPLIM
            assert out[0]["steps"] == -100000 and out[0]["status"] & capi.STATUS_STEPLIM
    assert out[777]["rdot_flips"] == 1 and out[777]["equatorial_crossings"] == 1


def test_rerun_skips_finished_and_steplimited_rays():
    """run_raytrace re-entrancy (reference raytracer.cpp:116-117, :335-337): step-limited rays get negative steps
    and are skipped by a second call; rays that finished keep integrating only if their loop condition still holds."""
    case = CASES["ps_h5"]
    rays = ol.oracle_pointsource(case["source"])
    p = capi.copy_params(case["runs"]["rk4"], steplim=300)
    out1, st1 = ol.oracle_trace(p, rays)
    lim = (out1["status"] & capi.STATUS_STEPLIM) != 0
    assert lim.any() and (out1["steps"][lim] == -300).all()
    out2, st2 = ol.oracle_trace(p, out1)
    assert (out2["steps"][lim] == -300).all()
    done = (out1["steps"] > 0)
    # finished rays sit on their stop surface: zero further steps, state unchanged
    assert ol.rays_equal_bitwise(out1[done], out2[done]) == []


def test_invalid_arguments():
    o = ol.oracle()
    rays = np.zeros(4, dtype=capi.RAY_F64)
    p = capi.default_params(0.5)
    p.integrator, p.stop_kind = capi.EULER, capi.STOP_FLATDISC     # assert in raytracer.cpp:983
    assert o.kro_trace_f64(C.byref(p), ol.ptr(rays), 4, 1, None) == capi.KR_EINVAL
    p.integrator, p.stop_kind = 7, capi.STOP_THETA
    assert o.kro_trace_f64(C.byref(p), ol.ptr(rays), 4, 1, None) == capi.KR_EINVAL
