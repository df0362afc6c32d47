"""GPU, BASELINE full size (configs[1]: 1e7 rays, RK4, fp64): size-independent properties of the device-resident pipeline.
  * a random sample of the 1e7 traced rays equals the oracle's trace of the very same input records (rays are independent)
  * re-running the trace on the finished array changes nothing but the horizon-captured rays (run_raytrace re-entrancy)
  * the emissivity histogram is additive over ray-cyclic shards (what the multi-GPU reduce relies on)
  * every traced ray ends in exactly one way, inside [horizon, r_max], with a positive step count"""
import ctypes as C
import math

import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as ol
import parity
from raytrace_cpu_amd import api, capi

pytestmark = pytest.mark.gpu
vp = C.c_void_p


def _spec(d):
    s = capi.PointSourceSpec()
    for i, v in enumerate([0.0, 10.0, 1e-3, 1.5707]):
        s.pos[i] = v
    s.V, s.spin, s.tol, s.E = 0.0, gc.SPIN, 100.0, 1.0
    s.cosalpha0, s.cosalphamax, s.dcosalpha = -0.995, 0.995, d
    s.beta0, s.betamax, s.dbeta = -math.pi, math.pi, d * math.pi / 0.995
    return s


@pytest.mark.parametrize("flags", [pytest.param(capi.FLAG_HYBRID, id="hybrid"), pytest.param(0, id="strict")])
def test_full_size_properties(krlib, flags):
    lib = krlib
    spec = _spec(1.99 / (math.sqrt(1e7) - 1.0))
    n_beta = C.c_int32()
    n = lib.kr_pointsource_count(C.byref(spec), None, C.byref(n_beta))
    assert n >= 9_990_000
    d_rays = vp()
    capi.check(lib, lib.kr_malloc(C.byref(d_rays), n * 144), "malloc")
    try:
        capi.check(lib, lib.kr_pointsource_init_dev_f64(C.byref(spec), d_rays, n, None), "init")
        capi.check(lib, lib.kr_redshift_start_dev_f64(gc.SPIN, 0.0, 0, 0, d_rays, n, None), "redshift_start")
        rng = np.random.default_rng(20261004)
        idx = np.sort(rng.choice(n, 3000, replace=False))
        # plus rays of the beta = -pi column, which holds the longest rays of this grid (2e4 .. 3.5e4 steps)
        idx = np.unique(np.concatenate([idx, np.arange(150, 800, 25) * n_beta.value]))
        before = np.zeros(len(idx), dtype=capi.RAY_F64)
        for k, i in enumerate(idx):
            capi.check(lib, lib.kr_memcpy_d2h(before[k:k + 1].ctypes.data_as(vp), vp(d_rays.value + int(i) * 144), 144), "d2h")
        p = capi.default_params(gc.SPIN)
        p.integrator, p.r_max, p.flags = capi.RK4, 1000.0, flags
        st = capi.Stats()
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, n, None, C.byref(st)), "trace")
        assert st.rays_traced > 9_990_000 and 4e9 < st.steps_total < 7e9
        assert st.rays_strict_side == 3162      # either way exactly the beta = -pi column (one ray per row) goes to the side launch
        after = np.zeros(len(idx), dtype=capi.RAY_F64)
        for k, i in enumerate(idx):
            capi.check(lib, lib.kr_memcpy_d2h(after[k:k + 1].ctypes.data_as(vp), vp(d_rays.value + int(i) * 144), 144), "d2h")
        # (1) sample vs oracle on identical inputs
        want, _ = ol.oracle_trace(p, before)
        res = parity.compare_rays(after, want, rtol=parity.RAY_RTOL)
        assert res["frac_bad"] <= parity.CHAOTIC_FRAC, res
        longest = int(np.argmax(want["steps"]))
        assert want["steps"][longest] > 20000 and after["steps"][longest] == want["steps"][longest]
        # (4) every sampled ray ended in exactly one way
        live = after["steps"] != -1
        term = parity.terminal_bits(after["status"][live])
        assert (np.isin(term, [capi.STATUS_DEST, capi.STATUS_HORIZON, capi.STATUS_RLIM])).all()
        assert (after["steps"][live] > 0).all() and (after["r"][live] <= 1000.0 * (1 + 1e-6)).all() and (after["r"][live] > 1.0).all()
        # (2) idempotence
        st2 = capi.Stats()
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, n, None, C.byref(st2)), "trace again")
        # rays that ended on their stop surface or at r_max take no further step; a ray that ended inside the horizon still
        # satisfies the loop condition and takes exactly one more step per call -- in the reference as well (raytracer.cpp:799, :917)
        again = np.zeros(len(idx), dtype=capi.RAY_F64)
        for k, i in enumerate(idx):
            capi.check(lib, lib.kr_memcpy_d2h(again[k:k + 1].ctypes.data_as(vp), vp(d_rays.value + int(i) * 144), 144), "d2h")
        sunk = (after["status"] & capi.STATUS_HORIZON) != 0
        assert ol.rays_equal_bitwise(after[~sunk], again[~sunk]) == []
        assert (again["steps"][sunk] == after["steps"][sunk] + 1).all()
        assert st2.rays_traced == st.rays_traced and st2.steps_total < 1e-4 * st.steps_total
        # (3) histogram additivity over ray-cyclic shards (strided init -> trace -> redshift -> reduce, per shard)
        capi.check(lib, lib.kr_range_phi_dev_f64(-math.pi, math.pi, d_rays, n, None), "range_phi")
        capi.check(lib, lib.kr_redshift_dev_f64(gc.SPIN, -1.0, 0, 0, 0, d_rays, n, None), "redshift")
        bins = gc.emis_bins(spec, nr=100)
        words = 5 * bins.nr + 1
        d_hist = vp()
        capi.check(lib, lib.kr_malloc(C.byref(d_hist), words * 8), "malloc")
        whole = np.zeros(words)
        capi.check(lib, lib.kr_memset(d_hist, 0, words * 8), "memset")
        capi.check(lib, lib.kr_reduce_emissivity_dev_f64(C.byref(bins), d_rays, n, d_hist, None), "reduce")
        capi.check(lib, lib.kr_memcpy_d2h(whole.ctypes.data_as(vp), d_hist, words * 8), "d2h")
        parts = np.zeros(words)
        shards = 4
        for r in range(shards):
            cnt = (n - r + shards - 1) // shards
            capi.check(lib, lib.kr_pointsource_init_strided_dev_f64(C.byref(spec), r, shards, d_rays, cnt, None), "init shard")
            capi.check(lib, lib.kr_redshift_start_dev_f64(gc.SPIN, 0.0, 0, 0, d_rays, cnt, None), "redshift_start")
            capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, cnt, None, None), "trace shard")
            capi.check(lib, lib.kr_range_phi_dev_f64(-math.pi, math.pi, d_rays, cnt, None), "range_phi")
            capi.check(lib, lib.kr_redshift_dev_f64(gc.SPIN, -1.0, 0, 0, 0, d_rays, cnt, None), "redshift")
            capi.check(lib, lib.kr_memset(d_hist, 0, words * 8), "memset")
            capi.check(lib, lib.kr_reduce_emissivity_dev_f64(C.byref(bins), d_rays, cnt, d_hist, None), "reduce")
            one = np.zeros(words)
            capi.check(lib, lib.kr_memcpy_d2h(one.ctypes.data_as(vp), d_hist, words * 8), "d2h")
            parts += one
        capi.check(lib, lib.kr_free(d_hist), "free")
        assert whole[5 * bins.nr] > 5e6
        np.testing.assert_array_equal(parts[:bins.nr], whole[:bins.nr])           # counts: exact
        np.testing.assert_allclose(parts, whole, rtol=1e-11)                      # sums: order of addition only
    finally:
        lib.kr_free(d_rays)


def test_rk45_creep_mode_at_scale(krlib):
    """1e6 rays of the BASELINE source under RK45, strict arithmetic: carrying the ~2000 creeping captured rays to the step limit
    from k1 alone (default) against iterating all of their 1e5 steps (KR_FLAG_RK45_ITERATE_ALL) -- every ray's r, theta, status,
    step count and counters bit-identical; t and phi of the creeping rays within 1e-10; everything else bit-identical."""
    spec = _spec(1.99 / (math.sqrt(1e6) - 1.0))
    init = api.pointsource_init(spec)
    api.redshift_start(gc.SPIN, 0.0, 0, 0, init)
    p = capi.default_params(gc.SPIN)
    p.integrator, p.r_max = capi.RK45, 1000.0
    quick, st = api.trace(p, init)
    full, st0 = api.trace(capi.copy_params(p, flags=capi.FLAG_RK45_ITERATE_ALL), init)
    assert st0["rk45_extrapolated_steps"] == 0 and st["rk45_extrapolated_steps"] > 1e8
    assert st["steps_total"] == st0["steps_total"] and st["rk45_attempts"] == st0["rk45_attempts"] and st["rk45_rejects"] == st0["rk45_rejects"]
    touched = np.zeros(len(init), dtype=bool)
    for f in quick.dtype.names:
        a, b = quick[f], full[f]
        same = (a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))
        if f in ("t", "phi", "pt", "pr", "ptheta", "pphi"):
            touched |= ~same
        else:
            assert same.all(), f
    lim = (full["status"] & capi.STATUS_STEPLIM) != 0
    assert 500 < touched.sum() <= lim.sum() and not (touched & ~lim).any()
    for f in ("t", "phi"):
        np.testing.assert_allclose(quick[f][touched], full[f][touched], rtol=1e-10)
