"""GPU: the HIP path, called through the C ABI, against (a) the committed golden fixtures captured from the
compiled reference and (b) the oracle on the same seeded inputs.  Tolerances: tests/parity.py."""
import ctypes as C

import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as ol
import parity
from raytrace_cpu_amd import api, capi

pytestmark = pytest.mark.gpu

CASES = gc.cases()
RUNS = [(c, r) for c in CASES for r in CASES[c]["runs"]]


# hybrid = fast arithmetic for well-conditioned rays + strict for the rest: held to the STRICT bar (no ray excluded)
MODES = [pytest.param(0, id="strict"), pytest.param(capi.FLAG_FAST_MATH, id="fastmath"), pytest.param(capi.FLAG_HYBRID, id="hybrid")]


def hip_pipeline(case, params, init, flags=0):
    """run_raytrace -> range_phi -> redshift through libkrtrace, starting from the reference's own init rays."""
    params = capi.copy_params(params, flags=flags)
    out, st = api.trace(params, init)
    api.range_phi(out)
    V, rev, proj = case["post"]
    if params.stop_kind == capi.STOP_THETA:
        api.redshift(params.spin, V, rev, proj, out)
    else:
        api.redshift_dest(params.spin, rev, out)
    return out, st


@pytest.mark.parametrize("flags", MODES)
@pytest.mark.parametrize("case_name,run", RUNS)
def test_trace_vs_golden(krlib, case_name, run, flags):
    case = CASES[case_name]
    g = np.load(gc.golden_path(case_name))
    params = case["runs"][run]
    out, st = hip_pipeline(case, params, g["init"], flags)
    want = g[f"final__{run}"]
    # kr_stats.longest_ray_steps: the step count of the call's longest ray (every ray of a fixture starts at steps = 0)
    traced = out["steps"] != -1
    assert st["longest_ray_steps"] == (int(np.abs(out["steps"][traced]).max()) if traced.any() else 0)
    assert 0 <= st["longest_ray_steps_strict_side"] <= st["longest_ray_steps"]
    rtol = parity.rtol_for(params)
    slack = parity.steps_slack_for(params, flags)
    mode = {0: "strict", capi.FLAG_FAST_MATH: "fastmath", capi.FLAG_HYBRID: "hybrid"}[flags]
    envelope = parity.noise_envelope_frac(params, g["init"], rtol)
    allowed = parity.allowed_bad_frac(params, g["init"], rtol, envelope=envelope)
    if flags & capi.FLAG_FAST_MATH:
        # same tolerances, knife-edge column excluded (parity.knife_edge_mask); step totals are then not comparable
        ke = parity.knife_edge_mask(g["init"], gc.is_imageplane(case))
        res = parity.compare_rays(parity.drop_rays(out, ke), parity.drop_rays(want, ke), rtol=rtol, check_redshift=True, steps_slack=slack)
        if case_name in parity.HYBRID_CAP:
            allowed = min(allowed, parity.HYBRID_CAP[case_name])
        parity.record_margin("test_trace_vs_golden", f"{case_name}-{run}-{mode}", res, allowed, envelope)
        assert res["n_traced"] > 0
        assert res["frac_bad"] <= allowed, res
        if case_name == "ps_h10_a0_landing":
            # the clipped last step onto the disc is the reference's correctly rounded quotient in every arithmetic mode: no ray takes an extra step
            assert res["n_steps_differ"] == 0 and res["n_int_fields_differ"] == 0, res
        return
    res = parity.compare_rays(out, want, rtol=rtol, check_redshift=True, steps_slack=slack)
    if flags == 0 and params.integrator != capi.RK45:
        allowed = parity.allowed_bad_frac_strict(params, res["n_traced"])          # strict arithmetic, fixed step: a fixed, tight bar
    elif case_name in parity.HYBRID_CAP:
        allowed = min(allowed, parity.HYBRID_CAP[case_name])
    parity.record_margin("test_trace_vs_golden", f"{case_name}-{run}-{mode}", res, allowed, envelope)
    assert res["n_traced"] > 0
    assert res["frac_bad"] <= allowed, res
    if case_name == "ps_h10_a0_landing":
        assert res["n_steps_differ"] == 0 and res["n_int_fields_differ"] == 0, res
    if flags == 0 and params.integrator != capi.RK45 and params.stop_kind in (capi.STOP_FLATDISC, capi.STOP_DISC_ISCO) and not gc.is_imageplane(case):
        # run_raytrace(RayDestination*) on the arithmetic the class mirror uses for it by default: the integer outcome of EVERY ray is the
        # reference's -- at a = 0.998 and at a = 0 / 0.5, where 0.1-0.35 % of the rays are decided at the 1-ulp level (tests/tool_oracle_isco_noise.py)
        assert res["n_int_fields_differ"] == 0 and res["n_steps_differ"] == 0, res
    if flags == 0 and params.integrator != capi.RK45:
        # strict arithmetic, fixed step: sin / cos are correctly rounded, everything else is IEEE -- nearly every ray carries the reference's bits in
        # every output (measured: PointSource 99.7-100 %, image plane 96.5-98.4 %: the rest is where glibc's own sin / cos is not correctly rounded)
        assert res["frac_bit_identical"] >= (0.95 if gc.is_imageplane(case) else 0.99), res["frac_bit_identical"]
    if flags == 0 and params.integrator == capi.RK45 and not gc.is_imageplane(case):
        # strict RK45 from a PointSource: sin / cos and the controller's root are correctly rounded -> measured 96-99 % bit-identical, the rays
        # that are not "bad" within 1e-10 (the bad ones are the NaN-ending polar rays, whose step of death is rounding-decided in the reference)
        # (the 1e-9 ceiling on the rays that are not bit-identical is a lamp post's at a = 0.998; at a <= 0.5 the rays that whirl inside the ISCO
        # amplify a last-bit difference of the controller's root to a few 1e-9 -- measured 1.8e-9 -- inside RK45's 1e-7 band)
        assert res["frac_bit_identical"] >= 0.93 and res["worst_ok"] <= (1e-9 if abs(params.spin) == gc.SPIN else 1e-8), (res["frac_bit_identical"], res["worst_ok"])
    if parity.is_unconverged_endpoint(params):
        # end positions are ill-conditioned in the reference itself; what a ray DID (hit the plane / escaped / fell in) is not
        assert res["frac_terminal_status_differs"] <= 0.01, res
    # the kernel's own step counter agrees with the per-ray records it wrote
    live = out["steps"] != -1
    assert st["steps_total"] == int((np.abs(out["steps"][live].astype(np.int64)) - np.abs(g["init"]["steps"][live].astype(np.int64))).sum())
    assert abs(st["steps_total"] - int(g[f"steps__{run}"])) <= 0.01 * int(g[f"steps__{run}"]) + 2 * res["n_traced"]
    assert st["rays_traced"] == int((g["init"]["steps"] >= 0).sum())


@pytest.mark.parametrize("flags", MODES)
@pytest.mark.parametrize("case_name", ["ps_h5", "ps_h10", "ps_kep", "ps_kep5k"])
@pytest.mark.parametrize("run", ["euler", "rk4", "rk45"])
def test_emissivity_bins_vs_golden(krlib, case_name, run, flags):
    case = CASES[case_name]
    if run not in case["runs"]:
        pytest.skip("run not defined for this case")
    g = np.load(gc.golden_path(case_name))
    out, _ = hip_pipeline(case, case["runs"][run], g["init"], flags)
    ref_final = g[f"final__{run}"]
    if flags & capi.FLAG_FAST_MATH:
        ke = parity.knife_edge_mask(g["init"], False)
        out, ref_final = parity.drop_rays(out, ke), parity.drop_rays(ref_final, ke)
    bins = gc.emis_bins(case["source"])
    got = api.reduce_emissivity(bins, out)
    # the oracle's reducer on the REFERENCE's final rays
    want = oracle_reduce_emissivity(bins, ref_final)
    assert want["disc_count"] > 0
    assert parity.compare_bins(got, want, label=f"test_emissivity_bins_vs_golden[{case_name}-{run}-flags{flags}]") == []
    assert abs(got["disc_count"] - want["disc_count"]) <= parity.BIN_COUNT_SLACK


def oracle_reduce_emissivity(bins, rays):
    nr = bins.nr
    count = np.zeros(nr, dtype=np.int64)
    flux, emis, sg, stt = (np.zeros(nr) for _ in range(4))
    dc = C.c_int64()
    ol.oracle().kro_reduce_emissivity_f64(C.byref(bins), ol.ptr(rays), len(rays), ol.ptr(count), ol.ptr(flux), ol.ptr(emis),
                                          ol.ptr(sg), ol.ptr(stt), C.byref(dc))
    return {"count": count, "flux": flux, "emis": emis, "sum_redshift": sg, "sum_time": stt, "disc_count": dc.value}


def oracle_reduce_image(bins, rays):
    npix = bins.img_nx * bins.img_ny
    nrays = np.zeros(npix, dtype=np.int32)
    keys = ("flux", "r", "phi", "enshift", "time", "emis")
    planes = {k: np.zeros(npix) for k in keys}
    dc = C.c_int64()
    ol.oracle().kro_reduce_image_f64(C.byref(bins), ol.ptr(rays), len(rays), ol.ptr(nrays), *[ol.ptr(planes[k]) for k in keys], C.byref(dc))
    out = {"nrays": nrays, "disc_count": dc.value}
    out.update(planes)
    return out


@pytest.mark.parametrize("flags", MODES)
@pytest.mark.parametrize("case_name,run", [("ip15", "rk4"), ("ip15", "rk45"), ("ip16", "rk4")])
def test_image_planes_vs_golden(krlib, case_name, run, flags):
    case = CASES[case_name]
    g = np.load(gc.golden_path(case_name))
    out, _ = hip_pipeline(case, case["runs"][run], g["init"], flags)
    ref_final = g[f"final__{run}"]
    if flags & capi.FLAG_FAST_MATH:
        ke = parity.knife_edge_mask(g["init"], True)
        out, ref_final = parity.drop_rays(out, ke), parity.drop_rays(ref_final, ke)
    bins = gc.image_bins(case["source"])
    got = api.reduce_image(bins, out)
    want = oracle_reduce_image(bins, ref_final)
    assert want["disc_count"] > 0
    assert np.abs(got["nrays"].astype(int) - want["nrays"].astype(int)).max() <= parity.BIN_COUNT_SLACK
    same = got["nrays"] == want["nrays"]
    for k in ("flux", "r", "phi", "enshift", "time", "emis"):
        np.testing.assert_allclose(got[k][same], want[k][same], rtol=parity.BIN_RTOL, atol=1e-12, err_msg=k)


# ---- sources and O(N) passes against the oracle ---------------------------------------------------------------
@pytest.mark.parametrize("case_name", list(CASES))
def test_source_init_and_redshift_start(krlib, case_name):
    case = CASES[case_name]
    spec = case["source"]
    g = np.load(gc.golden_path(case_name))
    rays = api.imageplane_init(spec) if gc.is_imageplane(case) else api.pointsource_init(spec)
    assert len(rays) == len(g["init"])
    params = next(iter(case["runs"].values()))
    V, rev, proj = case["start"]
    api.redshift_start(params.spin, V, rev, proj, rays)
    want = g["init"]
    assert (rays["steps"] == want["steps"]).all()
    live = want["steps"] == 0
    for f in ("rdot_sign", "thetadot_sign", "rdot_flips", "equatorial_crossings", "status"):
        assert (rays[f][live] == want[f][live]).all(), f
    for f in ("t", "r", "theta", "phi", "pt", "pr", "ptheta", "pphi", "k", "h", "Q", "emit", "alpha", "beta"):
        ok = live & ~(np.isnan(want[f]))
        np.testing.assert_allclose(rays[f][ok], want[f][ok], rtol=1e-11, atol=1e-13, err_msg=f)
        assert np.isnan(rays[f][live & np.isnan(want[f])]).all(), f
    # how many records carry the reference's bits (what is left: acos / atan2 of the device library, glibc's last-bit choices in sin / cos)
    same = {f: float(((rays[f][live].view(np.int64) == want[f][live].view(np.int64)) | (np.isnan(rays[f][live]) & np.isnan(want[f][live]))).mean()) for f in ("k", "h", "Q", "emit", "theta", "phi")}
    parity.record_margin("test_source_init_and_redshift_start", case_name, {"n_traced": int(live.sum()), "n_bad": 0, "frac_bad": 0.0, "worst_ok": None},
                         **{f"frac_bit_identical_{f}": v for f, v in same.items()})
    if gc.is_imageplane(case):
        # ImagePlane: N^2 distinct arguments of acos / atan2 / asin / tan, evaluated on the device by correctly rounded routines (kr_crmath.hpp): a record
        # differs from the reference constructor's only where glibc itself is not correctly rounded (per call: 0.06-0.2 % of arguments)
        for f, floor in (("theta", 0.99), ("phi", 0.99), ("Q", 0.98), ("h", 0.99), ("k", 1.0), ("emit", 0.99)):
            assert same[f] >= floor, (f, same[f])
    if not gc.is_imageplane(case):
        # PointSource: the constructor's sin / cos / acos / tan values come from host-built tables (glibc, as in the reference) and the rest is
        # IEEE + - x / sqrt on both sides -- every field of every device-built record carries the reference's bits (pointsource.cpp:30-64,
        # raytracer.cpp:625-676); `emit` goes through the device's correctly rounded sin / cos of the ONE source position
        for f in ("t", "r", "theta", "phi", "k", "h", "Q", "alpha", "beta"):
            assert (rays[f][live].view(np.int64) == want[f][live].view(np.int64)).all(), f
        assert same["emit"] == 1.0, same


def test_redshift_variants_vs_oracle(krlib):
    g = np.load(gc.golden_path("ps_h5"))
    fin = g["final__rk4"]
    o = ol.oracle()
    for (V, rev, proj, motion) in [(-1.0, 0, 0, 0), (-1.0, 0, 1, 0), (0.05, 0, 0, 0), (-0.3, 0, 0, 1), (-1.0, 1, 0, 0)]:
        a, b = fin.copy(), fin.copy()
        api.redshift(gc.SPIN, V, rev, proj, a, motion=motion)
        o.kro_redshift_f64(gc.SPIN, V, rev, proj, motion, ol.ptr(b), len(b))
        live = (fin["steps"] > 0) & np.isfinite(b["redshift"])
        np.testing.assert_allclose(a["redshift"][live], b["redshift"][live], rtol=1e-10)
        same = float((a["redshift"][live].view(np.int64) == b["redshift"][live].view(np.int64)).mean())
        parity.record_margin("test_redshift_variants_vs_oracle", f"V{V}-rev{rev}-proj{proj}-motion{motion}", {"n_traced": int(live.sum()), "n_bad": 0, "frac_bad": 0.0, "worst_ok": None},
                             frac_bit_identical_redshift=same)
        assert same >= 0.995, same       # measured 1.0 on all five variants: IEEE + - x / sqrt and correctly rounded sin / cos (glibc may pick another last bit)
    a, b = fin.copy(), fin.copy()
    api.calculate_momentum(gc.SPIN, a)
    o.kro_calculate_momentum_f64(gc.SPIN, ol.ptr(b), len(b))
    live = fin["steps"] > 0
    for f in ("pt", "pr", "ptheta", "pphi"):
        np.testing.assert_allclose(a[f][live], b[f][live], rtol=1e-10, atol=1e-14)
    # redshift_start with V = -1: the first record's orbital velocity is reused for every ray (raytracer.cpp:391-394)
    a, b = g["init"].copy(), g["init"].copy()
    api.redshift_start(gc.SPIN, -1.0, 0, 0, a)
    o.kro_redshift_start_f64(gc.SPIN, -1.0, 0, 0, ol.ptr(b), len(b))
    live = g["init"]["steps"] == 0
    np.testing.assert_allclose(a["emit"][live], b["emit"][live], rtol=1e-10)


def test_range_phi_bitwise(krlib):
    rays = np.zeros(4096, dtype=capi.RAY_F64)
    rng = np.random.default_rng(7)
    rays["phi"] = rng.uniform(-1200, 1200, len(rays))
    rays["phi"][:8] = [np.nan, np.pi, -np.pi, 1000.0, -1000.0, 999.9999, 0.0, 3 * np.pi]
    rays["steps"] = rng.integers(-2, 3, len(rays))
    a, b = rays.copy(), rays.copy()
    api.range_phi(a)
    ol.oracle().kro_range_phi_f64(-np.pi, np.pi, ol.ptr(b), len(b))
    assert ol.rays_equal_bitwise(a, b) == []


# ---- edge cases ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flags", MODES)
def test_empty_and_skipped_inputs(krlib, flags):
    p = capi.default_params(gc.SPIN)
    p.integrator, p.flags = capi.RK4, flags
    empty = np.zeros(0, dtype=capi.RAY_F64)
    out, st = api.trace(p, empty)
    assert len(out) == 0 and st["rays_traced"] == 0
    # all rays invalid (steps = -1) or already at the step limit: nothing may change
    rays = np.zeros(1000, dtype=capi.RAY_F64)
    rays["steps"] = -1
    rays["steps"][::3] = 10_000_000
    rays["r"] = 5.0
    out, st = api.trace(p, rays)
    assert st["rays_traced"] == 0 and st["steps_total"] == 0
    assert ol.rays_equal_bitwise(out, rays) == []


@pytest.mark.parametrize("flags", [pytest.param(0, id="strict"), pytest.param(capi.FLAG_HYBRID, id="hybrid")])
def test_ragged_sizes_match_oracle(krlib, flags):
    """n not a multiple of the wave / workgroup size, down to a single ray."""
    g = np.load(gc.golden_path("ps_h10"))
    init = g["init"]
    p = capi.copy_params(CASES["ps_h10"]["runs"]["rk4"], flags=flags)
    for n in (1, 63, 64, 65, 257, 1001):
        sub = init[200:200 + n].copy()
        out, _ = api.trace(p, sub)
        want, _ = ol.oracle_trace(p, sub)
        res = parity.compare_rays(out, want)
        assert res["n_bad"] <= max(1, int(0.02 * n)), (n, res)


def test_rerun_is_idempotent_and_resumes(krlib):
    """run_raytrace re-entrancy (raytracer.cpp:116-117, :335-337)."""
    g = np.load(gc.golden_path("ps_h5"))
    p = capi.copy_params(CASES["ps_h5"]["runs"]["rk4"], steplim=300)
    out1, st1 = api.trace(p, g["init"])
    lim = (out1["status"] & capi.STATUS_STEPLIM) != 0
    assert lim.any() and (out1["steps"][lim] == -300).all()
    out2, st2 = api.trace(p, out1)
    assert st2["steps_total"] == 0
    assert ol.rays_equal_bitwise(out1, out2) == []


def test_invalid_arguments_fail_loudly(krlib):
    p = capi.default_params(0.5)
    rays = np.zeros(4, dtype=capi.RAY_F64)
    p.integrator, p.stop_kind = capi.EULER, capi.STOP_FLATDISC          # assert in raytracer.cpp:983
    assert krlib.kr_trace_f64(C.byref(p), ol.ptr(rays), 4, None) == capi.KR_EINVAL
    assert b"Euler" in krlib.kr_last_error()
    p.integrator, p.stop_kind = 9, capi.STOP_THETA
    assert krlib.kr_trace_f64(C.byref(p), ol.ptr(rays), 4, None) == capi.KR_EINVAL
    p.integrator = capi.RK4
    assert krlib.kr_trace_f64(C.byref(p), None, 4, None) == capi.KR_EINVAL


def test_nan_ray_terminates(krlib):
    """ImagePlane pixel (0,0) has NaN constants; the reference's RK45 never returns for it.
    The device path must end the ray (KR_STATUS_NAN) and leave the others untouched by it."""
    g = np.load(gc.golden_path("ip16"))
    init = g["init"]
    nan_rays = np.flatnonzero(np.isnan(init["h"]) & (init["steps"] == 0))
    assert len(nan_rays) == 1
    p = capi.copy_params(CASES["ip16"]["runs"]["rk4"], integrator=capi.RK45)
    out, _ = api.trace(p, init)
    i = nan_rays[0]
    assert out["status"][i] & capi.STATUS_NAN
    want, _ = ol.oracle_trace(p, init)        # the oracle mirrors the same documented deviation
    res = parity.compare_rays(out, want, rtol=parity.rtol_for(p))
    assert res["frac_bad"] <= parity.allowed_bad_frac(p, init, parity.rtol_for(p)), res


# ---- larger grids against the oracle run on the spot --------------------------------------------------------------
@pytest.mark.parametrize("flags", MODES)
@pytest.mark.parametrize("method", [capi.EULER, capi.RK4, capi.RK45])
def test_perf_test_grid_vs_oracle(krlib, method, flags):
    """integrator_perf_test.cpp:35-45 grid (5167 rays) at BASELINE's h = 10: per-ray and per-bin parity."""
    spec = ol.pointsource_spec([0.0, 10.0, 1e-3, 1.5707], 0.0, gc.SPIN, 0.05, 0.05, cosalpha0=-0.995, cosalphamax=0.995,
                               beta0=-np.pi, betamax=np.pi)
    init = ol.oracle_pointsource(spec)
    ol.oracle().kro_redshift_start_f64(gc.SPIN, 0.0, 0, 0, ol.ptr(init), len(init))
    p = capi.default_params(gc.SPIN)
    p.integrator, p.flags = method, flags
    want, wst = ol.oracle_trace(p, init)
    ol.oracle().kro_range_phi_f64(-np.pi, np.pi, ol.ptr(want), len(want))
    ol.oracle().kro_redshift_f64(gc.SPIN, -1.0, 0, 0, 0, ol.ptr(want), len(want))
    out, st = api.trace(p, init)
    api.range_phi(out)
    api.redshift(gc.SPIN, -1.0, 0, 0, out)
    if flags & capi.FLAG_FAST_MATH:
        ke = parity.knife_edge_mask(init, False)
        out, want = parity.drop_rays(out, ke), parity.drop_rays(want, ke)
    res = parity.compare_rays(out, want, rtol=parity.rtol_for(p), check_redshift=True, steps_slack=parity.steps_slack_for(p, flags))
    parity.record_margin("test_perf_test_grid_vs_oracle", f"ps_h10_5167-{['euler', 'rk4', 'rk45'][method]}-flags{flags}", res, parity.CHAOTIC_FRAC)
    assert res["frac_bad"] <= parity.CHAOTIC_FRAC, res
    bins = gc.emis_bins(spec, nr=30)
    assert parity.compare_bins(api.reduce_emissivity(bins, out), oracle_reduce_emissivity(bins, want), label=f"test_perf_test_grid_vs_oracle[{method}-flags{flags}]") == []
    if not flags & capi.FLAG_FAST_MATH:
        assert abs(st["steps_total"] - wst["steps_total"]) <= 0.01 * wst["steps_total"]


# ---- float instantiation (Raytracer<float>, reference raytracer.cpp:1897) -----------------------------------------
# (share of rays within 1e-5 of the reference's float result in t, r, theta, phi; share carrying its bits in every output) -- measured with the
# float kernels' sinf / cosf / powf evaluated in double and rounded once (kr_device.hpp, KR_F32_VIA_F64; with the device library's float
# routines the image-plane shares were 0.71 / 0.51 and 0.74 / 0.14):
# ps_h10 euler 0.999 / 0.848, rk4 1.000 / 0.862, rk45 0.813 / 0.294; ip15 (rays start at r = 1e4) rk4 0.918 / 0.824, rk4_isco 0.949 / 0.695
F32_TRACE_FLOORS = {("ps_h10", "euler"): (0.99, 0.81), ("ps_h10", "rk4"): (0.99, 0.82), ("ps_h10", "rk45"): (0.77, 0.25),
                    ("ip15", "rk4"): (0.88, 0.78), ("ip15", "rk4_isco"): (0.91, 0.64)}


@pytest.mark.parametrize("case_name,run", [("ps_h10", "euler"), ("ps_h10", "rk4"), ("ps_h10", "rk45"), ("ip15", "rk4"), ("ip15", "rk4_isco")])
def test_f32_trace_vs_reference_float(krlib, case_name, run):
    """kr_trace_f32 against fixtures captured from the compiled reference's float instantiation
    (tests/golden/make_golden_f32.py).  Single precision carries ~7 digits and the reference's fixed-step
    heuristics take hundreds of steps, so per-ray agreement is asked at 2e-3 relative for >= 90 % of the rays
    (device sinf/cosf vs glibc's differ in the last bit; every later step amplifies it), integer outputs included."""
    from tests_f32 import F32_STEPLIM
    g = np.load(gc.golden_path(f"f32_{case_name}"))
    p = capi.copy_params(CASES[case_name]["runs"][run], steplim=F32_STEPLIM)
    out, st = api.trace(p, g["init"])
    want = g[f"final__{run}"]
    assert out.dtype == capi.RAY_F32 and st["rays_traced"] == int((g["init"]["steps"] >= 0).sum())
    dead = want["steps"] == -1
    assert (out["steps"][dead] == -1).all()
    live = ~dead
    ok = live.copy()
    for f in ("status", "rdot_flips", "equatorial_crossings"):
        ok &= out[f] == want[f]
    sunk = (want["status"] & (capi.STATUS_HORIZON | capi.STATUS_STEPLIM)) != 0
    for f in ("r", "theta", "t", "phi"):
        a, b = out[f].astype(np.float64), want[f].astype(np.float64)
        if f in ("t", "phi"):
            a, b = np.where(sunk, 0, a), np.where(sunk, 0, b)
        with np.errstate(invalid="ignore"):
            err = np.abs(a - b) / np.maximum(np.abs(b), 1.0)
        ok &= (err <= 2e-3) | (np.isnan(a) & np.isnan(b))
    frac = ok[live].mean()
    # how close it really runs (written to parity_margins.json): the share of rays carrying the reference's bits in every output, and within 1e-5
    bits = live.copy()
    tight = live.copy()
    for f in ("t", "r", "theta", "phi"):
        a, b = out[f], want[f]
        bits &= (a.view(np.int32) == b.view(np.int32)) | (f in ("t", "phi")) & sunk
        with np.errstate(invalid="ignore"):
            e = np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(np.abs(b.astype(np.float64)), 1.0)
        tight &= (e <= 1e-5) | (np.isnan(a) & np.isnan(b)) | (f in ("t", "phi")) & sunk
    bits &= (out["steps"] == want["steps"]) & (out["status"] == want["status"])
    parity.record_margin("test_f32_trace_vs_reference_float", f"{case_name}-{run}", {"n_traced": int(live.sum()), "n_bad": int((~ok[live]).sum()), "frac_bad": float(1 - frac), "worst_ok": None},
                         frac_bit_identical=float(bits[live].mean()), frac_within_1e_5=float(tight[live].mean()), allowed=0.10)
    assert frac >= 0.90, frac
    # per-case floors a little under what gfx950 measures (profiles/r02_parity_margins.json), so that a regression of the float kernels shows
    floor_1e5, floor_bits = F32_TRACE_FLOORS[(case_name, run)]
    assert tight[live].mean() >= floor_1e5 and bits[live].mean() >= floor_bits, (tight[live].mean(), bits[live].mean())
    # step totals agree to 2 %
    s_out = np.abs(out["steps"][live].astype(np.int64)).sum()
    s_want = np.abs(want["steps"][live].astype(np.int64)).sum()
    assert abs(s_out - s_want) <= 0.02 * s_want


def _f32_rel_err(got, want):
    a, b = got.astype(np.float64), want.astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        err = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    return np.where(same, 0.0, err)


@pytest.mark.parametrize("case_name,run", [("ps_h10", "euler"), ("ps_h10", "rk4"), ("ps_h10", "rk45"), ("ip15", "rk4"), ("ip15", "rk4_isco")])
def test_f32_passes_vs_reference_float(krlib, case_name, run):
    """The O(N) passes of the float instantiation on the device (kr_redshift_start_f32, kr_range_phi_f32, kr_redshift_f32,
    kr_redshift_dest_f32, kr_calculate_momentum_f32) against the compiled reference's Raytracer<float>, pass by pass ON THE REFERENCE'S
    OWN INPUTS (fixtures: tests/golden/make_golden_f32.py), so that each comparison is one evaluation deep: range_phi bit for bit
    (additions of a double 2 pi rounded to float, no library call); the others bit for bit on >= 90 % of the rays and within 1e-4 on all (the device's sinf /
    cosf may differ from glibc's in the last bit)."""
    g = np.load(gc.golden_path(f"f32_{case_name}"))
    case = CASES[case_name]
    spin = case["runs"][run].spin                 # the Raytracer member (already negated for an ImagePlane)
    # redshift_start on the source's records
    init = g["init"].copy()
    init["emit"] = 0
    api.redshift_start(spin, *case["start"], init)
    err = _f32_rel_err(init["emit"], g["init"]["emit"])
    margins = {"emit_max_rel": float(err.max()), "emit_frac_bit_identical": float((err == 0).mean())}
    assert err.max() <= 1e-5 and (err == 0).mean() >= 0.99, margins
    # after the trace: the reference's final records in, the reference's post-pass records as the expectation
    want = g[f"post__{run}"]
    rays = g[f"final__{run}"].copy()
    lo, hi = float(np.float32(-np.pi)), float(np.float32(np.pi))
    api.range_phi(rays, lo, hi)
    assert (rays["phi"].view(np.int32) == want["phi"].view(np.int32)).all()
    V, reverse, projradius = case["post"]
    api.redshift(spin, V, reverse, projradius, rays)
    api.calculate_momentum(spin, rays)
    for f in ("t", "r", "theta", "k", "h", "Q", "emit", "alpha", "beta"):
        assert (rays[f].view(np.int32) == want[f].view(np.int32)).all(), f            # untouched
    for f in ("steps", "status", "rdot_sign", "thetadot_sign", "rdot_flips", "equatorial_crossings"):
        assert (rays[f] == want[f]).all(), f
    live = want["steps"] > 0
    # measured on gfx950 (profiles/r02_parity_margins.json): emit carries the reference's bits on every ray, the other fields on 95-100 % of
    # them and are never further than 2e-6 away (the device's sinf / cosf differ from glibc's in the last bit on some arguments).
    for f in ("redshift", "pt", "pr", "ptheta", "pphi"):
        err = _f32_rel_err(rays[f], want[f])[live]
        margins[f"{f}_frac_bit_identical"] = float((err == 0).mean())
        margins[f"{f}_max_rel"] = float(err.max())
        assert (err == 0).mean() >= 0.90 and np.median(err) == 0 and err.max() <= 1e-4, (f, margins)
    if f"postdest__{run}" in g.files:
        api.redshift_dest(spin, reverse, rays)
        err = _f32_rel_err(rays["redshift"], g[f"postdest__{run}"])[live]
        margins["redshift_dest_frac_bit_identical"] = float((err == 0).mean())
        margins["redshift_dest_max_rel"] = float(err.max())
        assert (err == 0).mean() >= 0.90 and err.max() <= 1e-4, margins
    parity.record_margin("test_f32_passes_vs_reference_float", f"{case_name}-{run}", {"n_traced": int(live.sum()), "n_bad": 0, "frac_bad": 0.0, "worst_ok": None}, **margins)


def test_f32_device_pointer_passes_equal_the_host_pointer_ones(krlib):
    """kr_*_dev_f32 (device pointers + stream) against kr_*_f32 (host pointers): the same kernels, so the same bits in every record."""
    lib, vp = krlib, C.c_void_p
    g = np.load(gc.golden_path("f32_ps_h10"))
    spin = CASES["ps_h10"]["runs"]["rk4"].spin
    fin = g["final__rk4"]
    n = len(fin)
    want = fin.copy()
    api.range_phi(want, float(np.float32(-np.pi)), float(np.float32(np.pi)))
    api.redshift(spin, -1.0, 0, 0, want)
    api.calculate_momentum(spin, want)
    want_dest = want.copy()
    api.redshift_dest(spin, 0, want_dest)
    want_start = g["init"].copy()
    want_start["emit"] = 0
    api.redshift_start(spin, 0.0, 0, 0, want_start)
    d = vp()
    capi.check(lib, lib.kr_malloc(C.byref(d), n * 84), "malloc")
    try:
        got = fin.copy()
        capi.check(lib, lib.kr_memcpy_h2d(d, got.ctypes.data_as(vp), n * 84), "h2d")
        capi.check(lib, lib.kr_range_phi_dev_f32(float(np.float32(-np.pi)), float(np.float32(np.pi)), d, n, None), "range_phi")
        capi.check(lib, lib.kr_redshift_dev_f32(spin, -1.0, 0, 0, 0, d, n, None), "redshift")
        capi.check(lib, lib.kr_calculate_momentum_dev_f32(spin, d, n, None), "momentum")
        capi.check(lib, lib.kr_memcpy_d2h(got.ctypes.data_as(vp), d, n * 84), "d2h")
        assert got.tobytes() == want.tobytes()
        capi.check(lib, lib.kr_redshift_dest_dev_f32(spin, 0, d, n, None), "redshift_dest")
        capi.check(lib, lib.kr_memcpy_d2h(got.ctypes.data_as(vp), d, n * 84), "d2h")
        assert got.tobytes() == want_dest.tobytes()
        start = g["init"].copy()
        start["emit"] = 0
        capi.check(lib, lib.kr_memcpy_h2d(d, start.ctypes.data_as(vp), n * 84), "h2d")
        capi.check(lib, lib.kr_redshift_start_dev_f32(spin, 0.0, 0, 0, d, n, None), "redshift_start")
        capi.check(lib, lib.kr_memcpy_d2h(start.ctypes.data_as(vp), d, n * 84), "d2h")
        assert start.tobytes() == want_start.tobytes()
    finally:
        lib.kr_free(d)


def test_degenerate_denominators_match_oracle(krlib):
    """Schwarzschild (a = 0) with a source on the axis region: h is ~0 for every ray, so phidot ~ 0 and the step
    heuristic divides by it; and a ray record placed exactly ON the pole (sin(theta) = 0).  Where IEEE division gives
    inf / NaN the strict path must behave like the CPU: this exercises the evaluation-level guard of the lean
    division chains (kr_device.hpp::momentum_impl)."""
    spec = ol.pointsource_spec([0.0, 8.0, 1e-3, 0.0], 0.0, 0.0, 0.2, 0.2, cosalpha0=-0.995, cosalphamax=0.995, beta0=-np.pi, betamax=np.pi)
    init = ol.oracle_pointsource(spec)
    init["h"][::3] = 0.0                       # exactly zero axial angular momentum -> phidot exactly 0
    extra = init[:1].copy()
    extra["theta"] = [0.0]                       # exactly on the pole: sin(theta) = 0
    init = np.concatenate([init, extra])
    for method in (capi.EULER, capi.RK4, capi.RK45):
        p = capi.default_params(0.0)
        p.integrator, p.steplim = method, 20000
        want, _ = ol.oracle_trace(p, init)
        out, _ = api.trace(p, init)
        res = parity.compare_rays(out, want, rtol=parity.rtol_for(p))
        assert res["frac_bad"] <= parity.allowed_bad_frac(p, init, parity.rtol_for(p)), (method, res)
        # records poisoned by a zero denominator are non-finite on both sides (inf vs NaN may differ) and end at once
        bad = ~np.isfinite(want["r"])
        assert (~np.isfinite(out["r"][bad])).all()


def test_fast_arithmetic_keeps_the_time_cap_when_phidot_is_zero(krlib):
    """Pure KR_FLAG_FAST_MATH with phidot == 0 exactly (a = 0, h = 0): the fixed-step integrators' time and azimuth caps share one reciprocal,
    min(dt |phidot|, dphi |tdot|) / (|tdot| |phidot|), which would be 0 x inf = NaN there -- v_min would drop it and with it the TIME cap, which the
    reference applies inside maxtstep_rlim (raytracer.cpp:862-865: step > |max_tstep / tdot|).  |phidot| is floored at 1e-300 (kr_device.hpp::step_fixed):
    these rays must then take the reference's steps.  (Under the hybrid launch they go to the strict kernel anyway: |h| < 1e-13.)"""
    spec = ol.pointsource_spec([0.0, 8.0, 0.7, 0.0], 0.0, 0.0, 0.2, 0.2, cosalpha0=-0.995, cosalphamax=0.995, beta0=-np.pi, betamax=np.pi)
    init = ol.oracle_pointsource(spec)
    live = init["steps"] == 0
    init["h"][live] = 0.0                        # exactly zero axial angular momentum at a = 0 -> phidot exactly 0 along the whole ray
    for method in (capi.EULER, capi.RK4):
        p = capi.default_params(0.0)
        p.integrator, p.r_max, p.steplim = method, 1000.0, 200000
        assert p.max_tstep > 0 and p.maxtstep_rlim > 8.0            # the cap is active where these rays start
        want, wst = ol.oracle_trace(p, init)
        out, st = api.trace(capi.copy_params(p, flags=capi.FLAG_FAST_MATH), init)
        res = parity.compare_rays(out, want, rtol=parity.rtol_for(p), steps_slack=2)
        parity.record_margin("test_fast_arithmetic_keeps_the_time_cap_when_phidot_is_zero", ["euler", "rk4"][method], res, parity.CHAOTIC_FRAC)
        assert res["n_traced"] > 100 and res["frac_bad"] <= parity.CHAOTIC_FRAC, res
        assert abs(st["steps_total"] - wst["steps_total"]) <= 2 * res["n_traced"], (st["steps_total"], wst["steps_total"])


def test_return_radiation_vs_oracle(krlib):
    """BASELINE configs[4] semantics (disc -> disc returning radiation, per-radius relaunch): source on the disc at r_s,
    Keplerian, beta in [0, pi), Euler to 1.1 r_esc, then the escape / return / lost classification
    (src/return_radiation/disc_source_photonfrac_r.cpp:74-135).  That app is stale in the reference (does not compile),
    so the classification itself is pinned only by the oracle's restatement of its loop body; the trace is pinned as usual."""
    o, lib = ol.oracle(), api.lib()
    r_isco = o.kro_kerr_isco(gc.SPIN, 1)
    for r_s in (2.0, 6.0, 30.0):
        V = o.kro_disc_velocity(r_s, gc.SPIN, 1)
        spec = ol.pointsource_spec([0.0, r_s, np.pi / 2 - 1e-6, 1.5707], V, gc.SPIN, 0.05, 0.05, cosalpha0=-0.995, cosalphamax=0.995, beta0=0.0, betamax=np.pi)
        init = ol.oracle_pointsource(spec)
        o.kro_redshift_start_f64(gc.SPIN, V, 0, 0, ol.ptr(init), len(init))
        p = capi.default_params(gc.SPIN)
        p.integrator, p.r_max = capi.EULER, 1100.0
        want, _ = ol.oracle_trace(p, init)
        o.kro_range_phi_f64(-np.pi, np.pi, ol.ptr(want), len(want))
        out, _ = api.trace(p, init)
        api.range_phi(out)
        b = capi.ReturnBins()
        b.r_isco, b.r_disc, b.r_esc, b.source_r, b.source_phi = r_isco, 500.0, 1000.0, r_s, 1.5707
        b.plane_iso, b.limb, b.weight_norm, b.pad = 1, 0, 1, 0
        w4, g4 = (C.c_double * 4)(), (C.c_double * 4)()
        o.kro_reduce_return_f64(C.byref(b), ol.ptr(want), len(want), C.byref(w4))
        capi.check(lib, lib.kr_reduce_return_f64(C.byref(b), ol.ptr(out), len(out), C.byref(g4)), "reduce_return")
        w, g = np.array(w4), np.array(g4)
        assert w[0] > 0 and (w[1:] > 0).any()
        np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-9 * w[0])
        # the GPU reducer on the CPU's rays agrees to summation order
        capi.check(lib, lib.kr_reduce_return_f64(C.byref(b), ol.ptr(want), len(want), C.byref(g4)), "reduce_return")
        np.testing.assert_allclose(np.array(g4), w, rtol=1e-12)


# ---- hybrid mode is exactly "strict for the flagged rays, fast for the others" ------------------------------------
_same_bits = parity.same_records      # (bit for bit; a NaN equals a NaN)


@pytest.mark.parametrize("method", [capi.EULER, capi.RK4, capi.RK45])
def test_hybrid_is_the_union_of_strict_and_fast(krlib, method):
    g = np.load(gc.golden_path("ps_h10"))
    init = g["init"]
    p = capi.default_params(gc.SPIN)
    p.integrator = method
    strict, _ = api.trace(capi.copy_params(p, flags=0), init)
    fast, _ = api.trace(capi.copy_params(p, flags=capi.FLAG_FAST_MATH), init)
    hyb, st = api.trace(capi.copy_params(p, flags=capi.FLAG_HYBRID), init)
    ke = parity.knife_edge_mask(init, False) & (init["steps"] >= 0)
    assert st["rays_strict_side"] == int(ke.sum()) > 0              # the classifier finds exactly the beta = -pi / 0 columns
    assert _same_bits(hyb[ke], strict[ke])
    assert _same_bits(hyb[~ke], fast[~ke])
    # no ill-conditioned ray: one fast launch, nothing on the side
    clean = init[~ke]
    h2, st2 = api.trace(capi.copy_params(p, flags=capi.FLAG_HYBRID), clean)
    assert st2["rays_strict_side"] == 0 and _same_bits(h2, fast[~ke])
    # mostly ill-conditioned rays (all in the meridional plane h = 0): everything takes the strict kernel
    merid = init.copy()
    merid["h"] = 0.0
    s3, _ = api.trace(capi.copy_params(p, flags=0), merid)
    h3, st3 = api.trace(capi.copy_params(p, flags=capi.FLAG_HYBRID), merid)
    assert st3["rays_strict_side"] == int((merid["steps"] >= 0).sum()) and _same_bits(h3, s3)


def test_strict_launch_split_is_bitwise_neutral(krlib, monkeypatch):
    """Large all-strict launches put their ill-conditioned rays on the side launch as well (kr_trace.hip: kIsolateMinRays);
    that changes where rays run, never what is computed: same bits as the single launch (KR_NO_ISOLATE=1)."""
    import bench
    spec = bench.make_spec(capi, bench.grid_spacing_for(3.2e5))
    init = api.pointsource_init(spec)
    api.redshift_start(gc.SPIN, 0.0, 0, 0, init)
    assert len(init) >= (1 << 18)
    p = capi.default_params(gc.SPIN)
    p.integrator, p.r_max = capi.RK4, 1000.0
    split, st_split = api.trace(p, init)
    monkeypatch.setenv("KR_NO_ISOLATE", "1")
    single, st_single = api.trace(p, init)
    assert st_split["rays_strict_side"] > 0 and st_single["rays_strict_side"] == 0
    assert st_split["steps_total"] == st_single["steps_total"] and _same_bits(split, single)


def test_rk45_creep_mode_reproduces_iteration(krlib):
    """Captured RK45 rays whose theta advances by a whole number of ulps per outer step while r stands still are carried to the
    step limit from k1 alone (kr_device.hpp::creep_step).  Against iterating every step (KR_FLAG_RK45_ITERATE_ALL), on the strict
    path: every other ray bit-identical; on the creeping rays r, theta and every integer output bit-identical, t and phi to 1e-10."""
    g = np.load(gc.golden_path("ps_h5"))
    p = capi.copy_params(CASES["ps_h5"]["runs"]["rk45"], flags=0)
    fast_way, st = api.trace(p, g["init"])
    slow_way, st0 = api.trace(capi.copy_params(p, flags=capi.FLAG_RK45_ITERATE_ALL), g["init"])
    assert st0["rk45_extrapolated_steps"] == 0 and st["rk45_extrapolated_steps"] > 50 * 90000      # ps_h5: 62 creeping rays
    assert st["steps_total"] == st0["steps_total"] and st["rk45_attempts"] == st0["rk45_attempts"]
    approx = ("t", "phi", "pt", "pr", "ptheta", "pphi")
    moved = np.zeros(len(fast_way), dtype=bool)
    for f in fast_way.dtype.names:
        a, b = fast_way[f], slow_way[f]
        same = (a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))
        if f in approx:
            moved |= ~same
        else:
            assert same.all(), f
    lim = (slow_way["status"] & capi.STATUS_STEPLIM) != 0
    assert moved.sum() > 0 and not (moved & ~lim).any()                       # only step-limit rays were touched
    for f in ("t", "phi"):
        np.testing.assert_allclose(fast_way[f][moved], slow_way[f][moved], rtol=1e-10)


def test_fused_pipeline_ends_equal_the_separate_passes(krlib):
    """kr_pointsource_init_emit_dev_f64 == init + redshift_start, kr_post_emissivity_dev_f64 == range_phi + redshift + reduce:
    rays bit-identical, histogram counts identical, sums equal up to the order of the atomic additions."""
    import bench
    lib, vp = krlib, C.c_void_p
    spec = bench.make_spec(capi, bench.grid_spacing_for(2e5))
    n = lib.kr_pointsource_count(C.byref(spec), None, None)
    bins = gc.emis_bins(spec, nr=100)
    words = 5 * bins.nr + 1
    p = capi.default_params(gc.SPIN)
    p.integrator, p.r_max = capi.RK4, 1000.0
    bufs = []
    for fused in (False, True):
        d_rays, d_hist = vp(), vp()
        capi.check(lib, lib.kr_malloc(C.byref(d_rays), n * 144), "malloc")
        capi.check(lib, lib.kr_malloc(C.byref(d_hist), words * 8), "malloc")
        capi.check(lib, lib.kr_memset(d_hist, 0, words * 8), "memset")
        if fused:
            capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d_rays, n, None), "init_emit")
        else:
            capi.check(lib, lib.kr_pointsource_init_dev_f64(C.byref(spec), d_rays, n, None), "init")
            capi.check(lib, lib.kr_redshift_start_dev_f64(gc.SPIN, 0.0, 0, 0, d_rays, n, None), "redshift_start")
        start = np.zeros(n, dtype=capi.RAY_F64)
        capi.check(lib, lib.kr_memcpy_d2h(start.ctypes.data_as(vp), d_rays, n * 144), "d2h")
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, n, None, None), "trace")
        if fused:
            capi.check(lib, lib.kr_post_emissivity_dev_f64(gc.SPIN, -1.0, 0, 0, 0, -np.pi, np.pi, C.byref(bins), d_rays, n, d_hist, None), "post")
        else:
            capi.check(lib, lib.kr_range_phi_dev_f64(-np.pi, np.pi, d_rays, n, None), "range_phi")
            capi.check(lib, lib.kr_redshift_dev_f64(gc.SPIN, -1.0, 0, 0, 0, d_rays, n, None), "redshift")
            capi.check(lib, lib.kr_reduce_emissivity_dev_f64(C.byref(bins), d_rays, n, d_hist, None), "reduce")
        end = np.zeros(n, dtype=capi.RAY_F64)
        hist = np.zeros(words)
        capi.check(lib, lib.kr_memcpy_d2h(end.ctypes.data_as(vp), d_rays, n * 144), "d2h")
        capi.check(lib, lib.kr_memcpy_d2h(hist.ctypes.data_as(vp), d_hist, words * 8), "d2h")
        lib.kr_free(d_rays)
        lib.kr_free(d_hist)
        bufs.append((start, end, hist))
    (s0, e0, h0), (s1, e1, h1) = bufs
    assert ol.rays_equal_bitwise(s0, s1) == [] and ol.rays_equal_bitwise(e0, e1) == []
    assert h0[5 * bins.nr] > 1e5
    np.testing.assert_array_equal(h0[:bins.nr], h1[:bins.nr])
    np.testing.assert_allclose(h0, h1, rtol=1e-12)


def test_fused_image_pipeline_ends_equal_the_separate_passes(krlib):
    """kr_imageplane_init_emit_dev_f64 / kr_post_image_dev_f64 against the five separate image-pipeline passes."""
    lib, vp = krlib, C.c_void_p
    N = 301
    spec = ol.imageplane_spec(10000.0, 80.0, -30.0, 30.0, 60.0 / N, -30.0, 30.0, 60.0 / N, gc.SPIN)
    n = lib.kr_imageplane_count(C.byref(spec), None, None)
    b = capi.ImageBins()
    b.x0, b.y0, b.img_dx, b.img_dy = -30.0, -30.0, 60.0 / 64, 60.0 / 64
    b.r_isco, b.r_disc = lib.kr_kerr_isco(gc.SPIN, 1), 30.0
    b.q1, b.rb1, b.q2, b.rb2, b.q3 = 3.0, 4.0, 3.0, 10.0, 3.0
    b.img_nx, b.img_ny, b.flip_image, b.pad = 64, 64, 1, 0
    words = 7 * 64 * 64 + 1
    p = capi.default_params(-gc.SPIN)
    p.integrator, p.r_max = capi.RK4, 11000.0
    res = []
    for fused in (False, True):
        d_rays, d_pl = vp(), vp()
        capi.check(lib, lib.kr_malloc(C.byref(d_rays), n * 144), "malloc")
        capi.check(lib, lib.kr_malloc(C.byref(d_pl), words * 8), "malloc")
        capi.check(lib, lib.kr_memset(d_pl, 0, words * 8), "memset")
        if fused:
            capi.check(lib, lib.kr_imageplane_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 1, 0, d_rays, n, None), "init_emit")
        else:
            capi.check(lib, lib.kr_imageplane_init_dev_f64(C.byref(spec), d_rays, n, None), "init")
            capi.check(lib, lib.kr_redshift_start_dev_f64(-gc.SPIN, 0.0, 1, 0, d_rays, n, None), "redshift_start")
        start = np.zeros(n, dtype=capi.RAY_F64)
        capi.check(lib, lib.kr_memcpy_d2h(start.ctypes.data_as(vp), d_rays, n * 144), "d2h")
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, n, None, None), "trace")
        if fused:
            capi.check(lib, lib.kr_post_image_dev_f64(-gc.SPIN, -1.0, 1, 0, 0, -np.pi, np.pi, C.byref(b), d_rays, n, d_pl, None), "post")
        else:
            capi.check(lib, lib.kr_redshift_dev_f64(-gc.SPIN, -1.0, 1, 0, 0, d_rays, n, None), "redshift")
            capi.check(lib, lib.kr_range_phi_dev_f64(-np.pi, np.pi, d_rays, n, None), "range_phi")
            capi.check(lib, lib.kr_reduce_image_dev_f64(C.byref(b), d_rays, n, d_pl, None), "reduce")
        end = np.zeros(n, dtype=capi.RAY_F64)
        planes = np.zeros(words)
        capi.check(lib, lib.kr_memcpy_d2h(end.ctypes.data_as(vp), d_rays, n * 144), "d2h")
        capi.check(lib, lib.kr_memcpy_d2h(planes.ctypes.data_as(vp), d_pl, words * 8), "d2h")
        lib.kr_free(d_rays)
        lib.kr_free(d_pl)
        res.append((start, end, planes))
    (s0, e0, p0), (s1, e1, p1) = res
    assert ol.rays_equal_bitwise(s0, s1) == [] and ol.rays_equal_bitwise(e0, e1) == []
    assert p0[-1] > 1000 and p0[-1] == p1[-1]
    np.testing.assert_array_equal(p0[:64 * 64], p1[:64 * 64])
    np.testing.assert_allclose(p0, p1, rtol=1e-12)


def test_fused_init_with_keplerian_V_on_a_shard_equals_the_unsharded_source(krlib):
    """V = -1 makes redshift_start use the orbital velocity at rays[0] of the WHOLE source for every ray (raytracer.cpp:389-393).
    A strided shard (first = 1, stride = 2: what rank 1 of 2 generates) must use that same velocity, not its own first ray's."""
    lib, vp = krlib, C.c_void_p
    spec = ol.pointsource_spec([0.0, 6.0, np.pi / 2 - 1e-3, 0.3], gc.kep_velocity(6.0), gc.SPIN, 0.05, 0.05, cosalpha0=-0.995, cosalphamax=0.995, beta0=-np.pi, betamax=np.pi)
    n = lib.kr_pointsource_count(C.byref(spec), None, None)
    full = np.zeros(n, dtype=capi.RAY_F64)
    d = vp()
    capi.check(lib, lib.kr_malloc(C.byref(d), n * 144), "malloc")
    try:
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, -1.0, 0, 0, d, n, None), "init_emit")
        capi.check(lib, lib.kr_memcpy_d2h(full.ctypes.data_as(vp), d, n * 144), "d2h")
        # the separate passes on the whole source are the reference semantics
        sep = api.pointsource_init(spec)
        api.redshift_start(gc.SPIN, -1.0, 0, 0, sep)
        assert ol.rays_equal_bitwise(full, sep) == []
        m = (n - 1 + 1) // 2
        shard = np.zeros(m, dtype=capi.RAY_F64)
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 1, 2, -1.0, 0, 0, d, m, None), "init_emit shard")
        capi.check(lib, lib.kr_memcpy_d2h(shard.ctypes.data_as(vp), d, m * 144), "d2h")
        assert ol.rays_equal_bitwise(shard, full[1::2][:m]) == []
        live = shard["steps"] == 0
        assert live.sum() > 100 and np.isfinite(shard["emit"][live]).all() and np.ptp(shard["emit"][live]) > 0
    finally:
        lib.kr_free(d)


def test_fused_return_post_pass_equals_the_separate_passes(krlib):
    """kr_post_return_dev_f64 == kr_range_phi_dev_f64 + kr_reduce_return_dev_f64: records bit-identical, the four sums equal to the order of the atomics."""
    lib, vp = krlib, C.c_void_p
    g = np.load(gc.golden_path("ps_h5"))
    fin = g["final__rk4"].copy()
    fin["phi"] += 40.0                          # so that range_phi has something to wrap
    n = len(fin)
    b = capi.ReturnBins()
    b.r_isco, b.r_disc, b.r_esc, b.source_r, b.source_phi = lib.kr_kerr_isco(gc.SPIN, 1), 400.0, 1000.0, 5.0, 0.0
    b.plane_iso, b.limb, b.weight_norm, b.pad = 1, 1, 1, 0
    res = []
    for fused in (False, True):
        d, d_out = vp(), vp()
        capi.check(lib, lib.kr_malloc(C.byref(d), n * 144), "malloc")
        capi.check(lib, lib.kr_malloc(C.byref(d_out), 32), "malloc")
        capi.check(lib, lib.kr_memset(d_out, 0, 32), "memset")
        capi.check(lib, lib.kr_memcpy_h2d(d, fin.ctypes.data_as(vp), n * 144), "h2d")
        if fused:
            capi.check(lib, lib.kr_post_return_dev_f64(-np.pi, np.pi, C.byref(b), d, n, d_out, None), "post")
        else:
            capi.check(lib, lib.kr_range_phi_dev_f64(-np.pi, np.pi, d, n, None), "range_phi")
            capi.check(lib, lib.kr_reduce_return_dev_f64(C.byref(b), d, n, d_out, None), "reduce")
        rays, out = np.zeros(n, dtype=capi.RAY_F64), np.zeros(4)
        capi.check(lib, lib.kr_memcpy_d2h(rays.ctypes.data_as(vp), d, n * 144), "d2h")
        capi.check(lib, lib.kr_memcpy_d2h(out.ctypes.data_as(vp), d_out, 32), "d2h")
        lib.kr_free(d)
        lib.kr_free(d_out)
        res.append((rays, out))
    (r0, o0), (r1, o1) = res
    assert ol.rays_equal_bitwise(r0, r1) == []
    moved = r1["steps"] > 0
    assert (r1["phi"][moved] != fin["phi"][moved]).mean() > 0.9          # range_phi did wrap (rays beyond |phi| = 1000 and NaN stay as they are)
    assert o0[0] > 100 and o0[1] > 0 and o0[2] > 0
    np.testing.assert_allclose(o1, o0, rtol=1e-12)
