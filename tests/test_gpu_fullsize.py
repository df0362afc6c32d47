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
        res = parity.compare_rays(after, want, rtol=parity.RAY_RTOL, steps_slack=0)
        parity.record_margin("test_full_size_properties", f"ps1e7-rk4-flags{flags}-sample", res, parity.CHAOTIC_FRAC)
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


def test_timed_pipeline_equals_the_cpu_reference_at_full_size(krlib):
    """BASELINE configs[1] end to end, the way bench.py TIMES it -- device PointSource constructor + redshift_start (fused), hybrid trace, fused
    range_phi + redshift + histogram, nothing leaving HBM but the 501 histogram words -- against the CPU on the same 3162^2 grid: the reference's
    own PointSource constructor and run_raytrace (oracle/_ref; the oracle port where it was not built) + the oracle's O(N) passes and reducer.
      * every device-built record carries the reference constructor's bits (host-tabulated acos / sin / cos / tan, kr_post_device.hpp::SourceTables);
      * the launch takes exactly the CPU's number of RK4 steps;
      * every bin's count is the CPU's; sums within the north-star 1e-6 (measured ~1e-11)."""
    import os
    lib = krlib
    spec = _spec(1.99 / (math.sqrt(1e7) - 1.0))
    n = lib.kr_pointsource_count(C.byref(spec), None, None)
    p = capi.default_params(gc.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, 1000.0, capi.FLAG_HYBRID
    # CPU leg (reference where built): constructor -> redshift_start -> run_raytrace -> range_phi -> redshift -> reducer
    threads = min(64, len(os.sched_getaffinity(0)))
    if ol.ref() is not None:
        src = ol.RefSource(spec)
        src.lib.ref_redshift_start(src.h, 0.0, 0, 0)
        init = src.snapshot()
        try:
            C.CDLL("libgomp.so.1").omp_set_num_threads(threads)
        except OSError:
            pass
        src.run(p)
        cpu = src.snapshot()
        src.close()
    else:
        init = ol.oracle_pointsource(spec)
        ol.oracle().kro_redshift_start_f64(gc.SPIN, 0.0, 0, 0, ol.ptr(init), len(init))
        cpu, _ = ol.oracle_trace(p, init, nthreads=threads)
    assert len(init) == n
    live = cpu["steps"] != -1
    cpu_steps = int(np.abs(cpu["steps"][live].astype(np.int64)).sum())
    o = ol.oracle()
    o.kro_range_phi_f64(-math.pi, math.pi, ol.ptr(cpu), len(cpu))
    o.kro_redshift_f64(gc.SPIN, -1.0, 0, 0, 0, ol.ptr(cpu), len(cpu))
    bins = gc.emis_bins(spec, nr=100)
    nr = bins.nr
    cnt = np.zeros(nr, dtype=np.int64)
    flux, emis, sg, stt = (np.zeros(nr) for _ in range(4))
    dc = C.c_int64()
    o.kro_reduce_emissivity_f64(C.byref(bins), ol.ptr(cpu), len(cpu), ol.ptr(cnt), ol.ptr(flux), ol.ptr(emis), ol.ptr(sg), ol.ptr(stt), C.byref(dc))
    # device leg: the fused pipeline
    d_rays, d_hist = vp(), vp()
    words = 5 * nr + 1
    capi.check(lib, lib.kr_malloc(C.byref(d_rays), n * 144), "malloc")
    capi.check(lib, lib.kr_malloc(C.byref(d_hist), words * 8), "malloc")
    try:
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d_rays, n, None), "init_emit")
        dev_init = np.zeros(n, dtype=capi.RAY_F64)
        capi.check(lib, lib.kr_memcpy_d2h(dev_init.ctypes.data_as(vp), d_rays, n * 144), "d2h")
        assert np.array_equal(dev_init["steps"], init["steps"])
        fresh = init["steps"] == 0
        for f in ("t", "r", "theta", "phi", "k", "h", "Q", "emit", "alpha", "beta"):
            assert np.array_equal(dev_init[f][fresh].view(np.int64), init[f][fresh].view(np.int64)), f
        for f in ("rdot_sign", "thetadot_sign", "status"):
            assert np.array_equal(dev_init[f][fresh], init[f][fresh]), f
        del dev_init
        st = capi.Stats()
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, n, None, C.byref(st)), "trace")
        capi.check(lib, lib.kr_memset(d_hist, 0, words * 8), "memset")
        capi.check(lib, lib.kr_post_emissivity_dev_f64(gc.SPIN, -1.0, 0, 0, 0, -math.pi, math.pi, C.byref(bins), d_rays, n, d_hist, None), "post")
        h = np.zeros(words)
        capi.check(lib, lib.kr_memcpy_d2h(h.ctypes.data_as(vp), d_hist, words * 8), "d2h")
    finally:
        lib.kr_free(d_rays)
        lib.kr_free(d_hist)
    assert st.rays_traced == int(live.sum())
    assert st.steps_total == cpu_steps, (st.steps_total, cpu_steps)
    assert np.array_equal(np.rint(h[:nr]).astype(np.int64), cnt)
    assert int(round(h[5 * nr])) == dc.value > 5e6
    worst = 0.0
    for k, w in enumerate((flux, emis, sg, stt)):
        m = cnt > 0
        worst = max(worst, float(np.max(np.abs(h[(k + 1) * nr:(k + 2) * nr][m] - w[m]) / np.abs(w[m]))))
    parity.record_margin("test_timed_pipeline_equals_the_cpu_reference_at_full_size", "ps1e7-rk4-hybrid",
                         {"n_traced": int(live.sum()), "n_bad": 0, "frac_bad": 0.0, "worst_ok": worst}, bins_count_mismatch=0, rk_steps=int(st.steps_total))
    assert worst <= 1e-6, worst


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


def _fetch(lib, d_rays, idx):
    out = np.zeros(len(idx), dtype=capi.RAY_F64)
    for k, i in enumerate(idx):
        capi.check(lib, lib.kr_memcpy_d2h(out[k:k + 1].ctypes.data_as(vp), vp(d_rays.value + int(i) * 144), 144), "d2h")
    return out


def test_imageplane_4097_full_size_properties(krlib):
    """BASELINE configs[3]: the 4097 x 4097-ray observer plane (par_example/imageplane_disc_image.par_example geometry, img 4096^2),
    RK4, hybrid launch, whole pipeline in HBM.  The CPU cannot trace 1.7e7 rays inside a test, so:
      * a sample (random + the x = 0 column + the y = 0 row, where the ill-conditioned rays live) equals the oracle's trace of the
        same input records, integer outputs and step counts included;
      * the seven image planes are additive over 4 ray-cyclic shards (what the multi-GPU image reduce relies on): ray counts per
        pixel exactly, sums to the order of the atomic additions; same set of empty pixels (the NaN pattern of the app's output);
      * totals are consistent: rays counted into pixels == disc_count, every traced ray took at least one step."""
    lib = krlib
    N = 4096
    s = capi.ImagePlaneSpec()
    s.dist, s.inc_deg, s.x0, s.xmax, s.y0, s.ymax = 10000.0, 80.0, -30.0, 30.0, -30.0, 30.0
    s.dx = s.dy = 60.0 / N
    s.spin, s.phi0, s.precision = gc.SPIN, 0.0, 100.0
    total, nx, ny = api.imageplane_count(s)
    assert (nx, ny) == (N + 1, N + 1) and total == (N + 1) ** 2
    b = capi.ImageBins()
    b.x0, b.y0, b.img_dx, b.img_dy = s.x0, s.y0, 60.0 / N, 60.0 / N
    b.r_isco, b.r_disc = lib.kr_kerr_isco(gc.SPIN, 1), 30.0
    b.q1, b.rb1, b.q2, b.rb2, b.q3 = 3.0, 4.0, 3.0, 10.0, 3.0
    b.img_nx, b.img_ny, b.flip_image, b.pad = N, N, 1, 0
    words = 7 * N * N + 1
    p = capi.default_params(-gc.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, 1.1 * s.dist, capi.FLAG_HYBRID
    d_rays, d_pl = vp(), vp()
    capi.check(lib, lib.kr_malloc(C.byref(d_rays), total * 144), "malloc")
    capi.check(lib, lib.kr_malloc(C.byref(d_pl), words * 8), "malloc")
    try:
        capi.check(lib, lib.kr_imageplane_init_emit_dev_f64(C.byref(s), 0, 1, 0.0, 1, 0, d_rays, total, None), "init_emit")
        rng = np.random.default_rng(4097)
        mid = N // 2                                      # x = 0 is grid column 2048, y = 0 grid row 2048
        idx = np.unique(np.concatenate([rng.choice(total, 2500, replace=False), mid * ny + np.arange(0, ny, 16), np.arange(0, nx, 16) * ny + mid]))
        before = _fetch(lib, d_rays, idx)
        st = capi.Stats()
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, total, None, C.byref(st)), "trace")
        assert st.rays_traced == total
        assert 8000 <= st.rays_strict_side <= 8400                           # the x = 0 column and the y = 0 row
        after = _fetch(lib, d_rays, idx)
        want, _ = ol.oracle_trace(p, before)
        res = parity.compare_rays(after, want, rtol=parity.RAY_RTOL, steps_slack=0)
        parity.record_margin("test_imageplane_4097_full_size_properties", "ip4097-rk4-hybrid-sample", res, parity.CHAOTIC_FRAC)
        assert res["frac_bad"] <= parity.CHAOTIC_FRAC, res
        live = after["steps"] != -1
        assert (np.abs(after["steps"][live]) > 0).all()
        # planes of the whole grid
        capi.check(lib, lib.kr_memset(d_pl, 0, words * 8), "memset")
        capi.check(lib, lib.kr_post_image_dev_f64(-gc.SPIN, -1.0, 1, 0, 0, -math.pi, math.pi, C.byref(b), d_rays, total, d_pl, None), "post")
        whole = np.zeros(words)
        capi.check(lib, lib.kr_memcpy_d2h(whole.ctypes.data_as(vp), d_pl, words * 8), "d2h")
        # ... and accumulated over 4 ray-cyclic shards (rank r of 4 generates rays r, r + 4, ...; kr_post_image ADDS into d_planes)
        capi.check(lib, lib.kr_memset(d_pl, 0, words * 8), "memset")
        shards = 4
        for r in range(shards):
            cnt = (total - r + shards - 1) // shards
            capi.check(lib, lib.kr_imageplane_init_emit_dev_f64(C.byref(s), r, shards, 0.0, 1, 0, d_rays, cnt, None), "init shard")
            capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, cnt, None, None), "trace shard")
            capi.check(lib, lib.kr_post_image_dev_f64(-gc.SPIN, -1.0, 1, 0, 0, -math.pi, math.pi, C.byref(b), d_rays, cnt, d_pl, None), "post shard")
        parts = np.zeros(words)
        capi.check(lib, lib.kr_memcpy_d2h(parts.ctypes.data_as(vp), d_pl, words * 8), "d2h")
        npix = N * N
        assert whole[-1] > 1e6 and whole[-1] == parts[-1] == whole[:npix].sum()        # disc_count == rays binned
        np.testing.assert_array_equal(parts[:npix], whole[:npix])                        # ray count per pixel (hence the empty-pixel / NaN pattern)
        assert 0 < (whole[:npix] == 0).sum() < npix
        lit = whole[:npix] > 0
        for k in range(1, 7):
            np.testing.assert_allclose(parts[k * npix:(k + 1) * npix][lit], whole[k * npix:(k + 1) * npix][lit], rtol=1e-11, atol=1e-300)
            assert (whole[k * npix:(k + 1) * npix][~lit] == 0).all()
    finally:
        lib.kr_free(d_rays)
        lib.kr_free(d_pl)


def test_return_radiation_full_size_properties(krlib):
    """BASELINE configs[4]: 100 source radii x ~1e6 rays, Euler, one launch per radius spread over 4 streams (bench.py's
    ReturnRadiationWorkload, the driver that mirrors src/return_radiation/disc_source_photonfrac_r.cpp:74-135).
      * per radius: return + escape + lost <= ray_count (the classes are disjoint), and within 5 % of it;
      * the overlapped launches (4 streams, async tickets) and the merged batch (all radii resident, one side + one main launch)
        give the table of the serial relaunch loop;
      * one radius split into 3 ray-cyclic shards adds up to that radius's row;
      * a sample of one radius's rays equals the oracle's trace of the same records."""
    import types

    import torch

    import bench
    lib = krlib
    args = types.SimpleNamespace(integrator="euler", radii=100, rays=1e6, streams=4, scaling="weak")
    wl = bench.ReturnRadiationWorkload(args, lib, capi, api, 0, 1)
    wl.p.flags = capi.FLAG_HYBRID
    n = wl.n
    rays = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
    res = torch.zeros(wl.result_words, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    st = wl.step(rays.data_ptr(), res.data_ptr(), stream)
    torch.cuda.synchronize()
    table = res.cpu().numpy().reshape(100, 4).copy()
    assert st["rays_traced"] > 9.9e7 and st["steps_total"] > 1e10
    tot, ret, esc, lost = table.T
    assert (tot > 1e5).all()
    # the three classes do not cover every ray (disc_source_photonfrac_r.cpp:104-126: hits next to the source itself and rays
    # between r_disc and r_esc are counted in ray_count only), but nearly all, and never more than all
    frac = (ret + esc + lost) / tot
    assert (frac <= 1 + 1e-12).all() and (frac > 0.95).all(), (frac.min(), frac.max())
    assert np.ptp(tot) < 1e-6                                                 # same angular grid at every radius (sum order only)
    assert ((ret >= 0) & (esc >= 0) & (lost >= 0)).all() and (ret[0] / tot[0] > esc[0] / tot[0]) and (esc[-1] / tot[-1] > 0.4)
    # serial relaunch loop on one stream: same table
    args1 = types.SimpleNamespace(integrator="euler", radii=100, rays=1e6, streams=1, scaling="weak")
    wl1 = bench.ReturnRadiationWorkload(args1, lib, capi, api, 0, 1)
    wl1.p.flags = capi.FLAG_HYBRID
    res.zero_()
    st1 = wl1.step(rays.data_ptr(), res.data_ptr(), stream)
    torch.cuda.synchronize()
    table1 = res.cpu().numpy().reshape(100, 4)
    assert st1["steps_total"] == st["steps_total"] and st1["rays_traced"] == st["rays_traced"]
    np.testing.assert_allclose(table1, table, rtol=1e-11)
    # ... and all 100 radii resident (14.4 GB), traced by ONE merged batch (bench.py --streams 0, the default): same table again
    args0 = types.SimpleNamespace(integrator="euler", radii=100, rays=1e6, streams=0, scaling="weak")
    wl0 = bench.ReturnRadiationWorkload(args0, lib, capi, api, 0, 1)
    wl0.p.flags = capi.FLAG_HYBRID
    res.zero_()
    st0 = wl0.step(rays.data_ptr(), res.data_ptr(), stream)
    torch.cuda.synchronize()
    table0 = res.cpu().numpy().reshape(100, 4)
    assert st0["steps_total"] == st["steps_total"] and st0["rays_traced"] == st["rays_traced"]
    np.testing.assert_allclose(table0, table, rtol=1e-11)
    assert wl0.groups == 4                                   # (the default: four merged batches on streams of their own, pipelined)
    # ... and as ONE merged batch
    wl0.groups = 1
    res.zero_()
    st0 = wl0.step(rays.data_ptr(), res.data_ptr(), stream)
    torch.cuda.synchronize()
    assert st0["steps_total"] == st["steps_total"] and st0["rays_traced"] == st["rays_traced"]
    np.testing.assert_allclose(res.cpu().numpy().reshape(100, 4), table, rtol=1e-11)
    del wl0
    # one radius: shard additivity + sample vs oracle
    j = 37
    spec, r_s = wl.specs[j], wl.radii[j][1]
    cnt_all = wl.counts[j]
    bins = capi.ReturnBins()
    bins.r_isco, bins.r_disc, bins.r_esc, bins.source_r, bins.source_phi = wl.r_isco, bench.R_DISC, bench.R_MAX, r_s, 1.5707
    bins.plane_iso, bins.limb, bins.weight_norm, bins.pad = 1, 0, 1, 0
    d = vp(rays.data_ptr())
    acc = torch.zeros(4, dtype=torch.float64, device="cuda")
    for r in range(3):
        cnt = (cnt_all - r + 2) // 3
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), r, 3, spec.V, 0, 0, d, cnt, None), "init shard")
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(wl.p), d, cnt, None, None), "trace shard")
        capi.check(lib, lib.kr_range_phi_dev_f64(-math.pi, math.pi, d, cnt, None), "range_phi")
        capi.check(lib, lib.kr_reduce_return_dev_f64(C.byref(bins), d, cnt, vp(acc.data_ptr()), None), "reduce")
    torch.cuda.synchronize()
    np.testing.assert_allclose(acc.cpu().numpy(), table[j], rtol=1e-11)
    capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, spec.V, 0, 0, d, cnt_all, None), "init")
    idx = np.sort(np.random.default_rng(37).choice(cnt_all, 1500, replace=False))
    before = _fetch(lib, d, idx)
    capi.check(lib, lib.kr_trace_dev_f64(C.byref(wl.p), d, cnt_all, None, None), "trace")
    after = _fetch(lib, d, idx)
    want, _ = ol.oracle_trace(wl.p, before)
    resc = parity.compare_rays(after, want, rtol=parity.RAY_RTOL, steps_slack=0)
    parity.record_margin("test_return_radiation_full_size_properties", "rr-radius37-euler-hybrid-sample", resc, parity.CHAOTIC_FRAC)
    assert resc["frac_bad"] <= parity.CHAOTIC_FRAC, resc


def test_ray_buffer_beyond_4_gib(krlib):
    """One ray buffer of 3.3e7 records = 4.75 GB (BASELINE sizes stay under 2^32 bytes: 1e7 x 144 = 1.44e9, 4097^2 x 144 = 2.4e9), so that
    every kernel's record addressing is exercised past the 32-bit byte offset: source + redshift_start fused and separate, the hybrid trace,
    the fused post pass and the separate ones.  Checks: a sample of rays from BEYOND the 4 GiB mark equals the oracle's trace of the same
    input records; every valid ray was traced; fused and separate passes agree (bits / counts) over the whole buffer."""
    lib = krlib
    spec = _spec(1.99 / (math.sqrt(3.3e7) - 1.0))
    n_rows, n_beta = C.c_int32(), C.c_int32()
    n = lib.kr_pointsource_count(C.byref(spec), C.byref(n_rows), C.byref(n_beta))
    assert n * 144 > (1 << 32) + (1 << 28)
    first_beyond = (1 << 32) // 144 + 1
    d_rays, d_hist = vp(), vp()
    capi.check(lib, lib.kr_malloc(C.byref(d_rays), n * 144), "malloc")
    bins = gc.emis_bins(spec, nr=100)
    words = 5 * bins.nr + 1
    capi.check(lib, lib.kr_malloc(C.byref(d_hist), words * 8), "malloc")

    def fetch(idx):
        out = np.zeros(len(idx), dtype=capi.RAY_F64)
        for k, i in enumerate(idx):
            capi.check(lib, lib.kr_memcpy_d2h(out[k:k + 1].ctypes.data_as(vp), vp(d_rays.value + int(i) * 144), 144), "d2h")
        return out

    def histogram():
        h = np.zeros(words)
        capi.check(lib, lib.kr_memcpy_d2h(h.ctypes.data_as(vp), d_hist, words * 8), "d2h")
        return h

    try:
        rng = np.random.default_rng(4097)
        idx = np.sort(rng.choice(np.arange(first_beyond, n), 1500, replace=False))
        idx = np.unique(np.concatenate([idx, [first_beyond, n - 1, (n_rows.value - 2) * n_beta.value]]))      # + a beta = -pi ray (strict side launch)
        p = capi.default_params(gc.SPIN)
        p.integrator, p.r_max, p.flags = capi.RK4, 1000.0, capi.FLAG_HYBRID
        # separate passes
        capi.check(lib, lib.kr_pointsource_init_dev_f64(C.byref(spec), d_rays, n, None), "init")
        capi.check(lib, lib.kr_redshift_start_dev_f64(gc.SPIN, 0.0, 0, 0, d_rays, n, None), "redshift_start")
        before = fetch(idx)
        st = capi.Stats()
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, n, None, C.byref(st)), "trace")
        assert st.rays_traced == n_rows.value * n_beta.value <= n          # (nRays is the truncated product of doubles: a few surplus slots stay unused)
        assert st.rays_strict_side == n_rows.value                         # the beta = -pi column, one ray per row
        after = fetch(idx)
        want, _ = ol.oracle_trace(p, before)
        res = parity.compare_rays(after, want, rtol=parity.RAY_RTOL, steps_slack=0)
        parity.record_margin("test_ray_buffer_beyond_4_gib", "ps3.3e7-rk4-hybrid-sample-beyond-4GiB", res, parity.CHAOTIC_FRAC)
        assert res["frac_bad"] <= parity.CHAOTIC_FRAC, res
        assert (after["steps"][before["steps"] == 0] > 0).all()
        capi.check(lib, lib.kr_range_phi_dev_f64(-math.pi, math.pi, d_rays, n, None), "range_phi")
        capi.check(lib, lib.kr_redshift_dev_f64(gc.SPIN, -1.0, 0, 0, 0, d_rays, n, None), "redshift")
        capi.check(lib, lib.kr_memset(d_hist, 0, words * 8), "memset")
        capi.check(lib, lib.kr_reduce_emissivity_dev_f64(C.byref(bins), d_rays, n, d_hist, None), "reduce")
        sep_rays, sep_hist = fetch(idx), histogram()
        # the two passes vs the oracle on the sampled (device-traced) records: one evaluation deep
        want_post = after.copy()
        ol.oracle().kro_range_phi_f64(-math.pi, math.pi, ol.ptr(want_post), len(want_post))
        ol.oracle().kro_redshift_f64(gc.SPIN, -1.0, 0, 0, 0, ol.ptr(want_post), len(want_post))
        assert ol.rays_equal_bitwise(sep_rays, want_post, fields=["phi"]) == []
        live = (after["steps"] > 0) & np.isfinite(want_post["redshift"])
        np.testing.assert_allclose(sep_rays["redshift"][live], want_post["redshift"][live], rtol=1e-10)
        # fused passes over the same buffer
        capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, d_rays, n, None), "init_emit")
        assert ol.rays_equal_bitwise(fetch(idx), before) == []
        st2 = capi.Stats()
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, n, None, C.byref(st2)), "trace")
        assert (st2.rays_traced, st2.steps_total, st2.rays_strict_side) == (st.rays_traced, st.steps_total, st.rays_strict_side)
        capi.check(lib, lib.kr_memset(d_hist, 0, words * 8), "memset")
        capi.check(lib, lib.kr_post_emissivity_dev_f64(gc.SPIN, -1.0, 0, 0, 0, -math.pi, math.pi, C.byref(bins), d_rays, n, d_hist, None), "post")
        fused_rays, fused_hist = fetch(idx), histogram()
        assert ol.rays_equal_bitwise(fused_rays, sep_rays) == []
        assert sep_hist[5 * bins.nr] > 0.5 * n
        np.testing.assert_array_equal(fused_hist[:bins.nr], sep_hist[:bins.nr])
        np.testing.assert_allclose(fused_hist, sep_hist, rtol=1e-11)
    finally:
        lib.kr_free(d_rays)
        lib.kr_free(d_hist)


def test_imageplane_buffer_beyond_4_gib(krlib):
    """The same for the observer plane: 5801 x 5801 rays = 4.85 GB in one buffer, img 1024^2.  A sample of rays from beyond the 4 GiB mark
    equals the oracle's trace of the same records; the fused ends (init + redshift_start, redshift + range_phi + planes) equal the
    separate passes over the whole buffer (sampled records bit for bit, pixel counts exactly, plane sums to the order of the atomics)."""
    lib = krlib
    N, IMG = 5800, 1024
    s = capi.ImagePlaneSpec()
    s.dist, s.inc_deg, s.x0, s.xmax, s.y0, s.ymax = 10000.0, 80.0, -30.0, 30.0, -30.0, 30.0
    s.dx = s.dy = 60.0 / N
    s.spin, s.phi0, s.precision = gc.SPIN, 0.0, 100.0
    total, nx, ny = api.imageplane_count(s)
    assert total * 144 > (1 << 32) + (1 << 28)
    first_beyond = (1 << 32) // 144 + 1
    b = capi.ImageBins()
    b.x0, b.y0, b.img_dx, b.img_dy = s.x0, s.y0, 60.0 / IMG, 60.0 / IMG
    b.r_isco, b.r_disc = lib.kr_kerr_isco(gc.SPIN, 1), 30.0
    b.q1, b.rb1, b.q2, b.rb2, b.q3 = 3.0, 4.0, 3.0, 10.0, 3.0
    b.img_nx, b.img_ny, b.flip_image, b.pad = IMG, IMG, 1, 0
    words = 7 * IMG * IMG + 1
    npix = IMG * IMG
    p = capi.default_params(-gc.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, 1.1 * s.dist, capi.FLAG_HYBRID
    d_rays, d_pl = vp(), vp()
    capi.check(lib, lib.kr_malloc(C.byref(d_rays), total * 144), "malloc")
    capi.check(lib, lib.kr_malloc(C.byref(d_pl), words * 8), "malloc")

    def planes():
        out = np.zeros(words)
        capi.check(lib, lib.kr_memcpy_d2h(out.ctypes.data_as(vp), d_pl, words * 8), "d2h")
        return out

    try:
        rng = np.random.default_rng(5801)
        idx = np.unique(np.concatenate([rng.choice(np.arange(first_beyond, total), 1200, replace=False), [first_beyond, total - 1]]))
        # separate passes
        capi.check(lib, lib.kr_imageplane_init_dev_f64(C.byref(s), d_rays, total, None), "init")
        capi.check(lib, lib.kr_redshift_start_dev_f64(-gc.SPIN, 0.0, 1, 0, d_rays, total, None), "redshift_start")
        before = _fetch(lib, d_rays, idx)
        st = capi.Stats()
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, total, None, C.byref(st)), "trace")
        assert st.rays_traced == total and 0 < st.rays_strict_side < 3 * (N + 1)
        after = _fetch(lib, d_rays, idx)
        want, _ = ol.oracle_trace(p, before)
        res = parity.compare_rays(after, want, rtol=parity.RAY_RTOL, steps_slack=0)
        parity.record_margin("test_imageplane_buffer_beyond_4_gib", "ip5801-rk4-hybrid-sample-beyond-4GiB", res, parity.CHAOTIC_FRAC)
        assert res["frac_bad"] <= parity.CHAOTIC_FRAC, res
        capi.check(lib, lib.kr_redshift_dev_f64(-gc.SPIN, -1.0, 1, 0, 0, d_rays, total, None), "redshift")
        capi.check(lib, lib.kr_range_phi_dev_f64(-math.pi, math.pi, d_rays, total, None), "range_phi")
        capi.check(lib, lib.kr_memset(d_pl, 0, words * 8), "memset")
        capi.check(lib, lib.kr_reduce_image_dev_f64(C.byref(b), d_rays, total, d_pl, None), "reduce")
        sep_rays, sep = _fetch(lib, d_rays, idx), planes()
        # fused ends
        capi.check(lib, lib.kr_imageplane_init_emit_dev_f64(C.byref(s), 0, 1, 0.0, 1, 0, d_rays, total, None), "init_emit")
        assert ol.rays_equal_bitwise(_fetch(lib, d_rays, idx), before) == []
        st2 = capi.Stats()
        capi.check(lib, lib.kr_trace_dev_f64(C.byref(p), d_rays, total, None, C.byref(st2)), "trace")
        assert (st2.rays_traced, st2.steps_total, st2.rays_strict_side) == (st.rays_traced, st.steps_total, st.rays_strict_side)
        capi.check(lib, lib.kr_memset(d_pl, 0, words * 8), "memset")
        capi.check(lib, lib.kr_post_image_dev_f64(-gc.SPIN, -1.0, 1, 0, 0, -math.pi, math.pi, C.byref(b), d_rays, total, d_pl, None), "post")
        fused_rays, fused = _fetch(lib, d_rays, idx), planes()
        assert ol.rays_equal_bitwise(fused_rays, sep_rays) == []
        assert sep[-1] > 1e6 and sep[-1] == fused[-1] == sep[:npix].sum()
        np.testing.assert_array_equal(fused[:npix], sep[:npix])
        np.testing.assert_allclose(fused, sep, rtol=1e-9, atol=1e-300)       # order of the atomic additions only (the phi plane's sums cancel: 1e-11 seen)
    finally:
        lib.kr_free(d_rays)
        lib.kr_free(d_pl)
