"""GPU: the emissivity pipeline as ONE trace kernel (kr_emissivity_pipeline_dev_f64: rays built in the kernel's load path, redshifted and binned in
its store path, no ray records in memory -- SURVEY section 7 step 6) against the same pipeline over 144-byte records in HBM
(kr_pointsource_init_emit_dev_f64 + kr_trace_dev_f64 + kr_post_emissivity_dev_f64) and against the oracle.  The per-ray device functions are shared,
so bin counts and the trace's own counters are identical; the sums differ only in the order of addition."""
import ctypes as C
import math

import numpy as np
import pytest

import bench
import golden_cases as gc
import oracle_lib as ol
from raytrace_cpu_amd import api, capi

pytestmark = pytest.mark.gpu
vp = C.c_void_p


def _hist(lib, nr):
    d = C.c_void_p()
    capi.check(lib, lib.kr_malloc(C.byref(d), (5 * nr + 1) * 8), "malloc")
    capi.check(lib, lib.kr_memset(d, 0, (5 * nr + 1) * 8), "memset")
    return d


def _fetch(lib, d, nr):
    h = np.zeros(5 * nr + 1)
    capi.check(lib, lib.kr_memcpy_d2h(h.ctypes.data_as(vp), d, h.nbytes), "d2h")
    return h


def records_pipeline(lib, spec, p, bins, first, stride, n):
    d_rays = C.c_void_p()
    capi.check(lib, lib.kr_malloc(C.byref(d_rays), n * capi.RAY_F64.itemsize), "malloc")
    d_h = _hist(lib, bins.nr)
    capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), first, stride, 0.0, 0, 0, d_rays, n, None), "init")
    st = api.trace_dev(p, d_rays.value, n)
    capi.check(lib, lib.kr_post_emissivity_dev_f64(spec.spin, -1.0, 0, 0, 0, -math.pi, math.pi, C.byref(bins), d_rays, n, d_h, None), "post")
    capi.check(lib, lib.kr_synchronize(None), "sync")
    h = _fetch(lib, d_h, bins.nr)
    lib.kr_free(d_rays); lib.kr_free(d_h)
    return h, st


def kernel_pipeline(lib, spec, p, bins, first, stride, n):
    d_h = _hist(lib, bins.nr)
    st = capi.Stats()
    capi.check(lib, lib.kr_emissivity_pipeline_dev_f64(C.byref(spec), first, stride, n, 0.0, 0, 0, C.byref(p), spec.spin, -1.0, 0, 0, 0, C.byref(bins), d_h, None,
                                                       C.byref(st)), "pipeline")
    capi.check(lib, lib.kr_synchronize(None), "sync")
    h = _fetch(lib, d_h, bins.nr)
    lib.kr_free(d_h)
    return h, st.as_dict()


def same_histogram(got, want, nr):
    assert np.array_equal(got[:nr], want[:nr]), "bin counts differ"                 # integer counts: exact
    assert got[5 * nr] == want[5 * nr], "disc count differs"
    for k in range(1, 5):
        np.testing.assert_allclose(got[k * nr:(k + 1) * nr], want[k * nr:(k + 1) * nr], rtol=1e-12, atol=0)


@pytest.mark.parametrize("integrator", ["rk4", "rk45", "euler"])
@pytest.mark.parametrize("mode", ["hybrid", "strict", "fast"])
def test_kernel_pipeline_equals_the_record_pipeline(krlib, integrator, mode):
    lib = krlib
    rays = 4.0e5 if integrator != "rk45" else 6.0e4
    spec = bench.make_spec(capi, bench.grid_spacing_for(rays))
    n, _, _ = api.pointsource_count(spec)
    bins = bench.emis_bins(capi, lib.kr_kerr_isco(bench.SPIN, 1), n)
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max = {"rk4": capi.RK4, "rk45": capi.RK45, "euler": capi.EULER}[integrator], bench.R_MAX
    p.flags = {"hybrid": capi.FLAG_HYBRID, "strict": 0, "fast": capi.FLAG_FAST_MATH}[mode]
    want, st_w = records_pipeline(lib, spec, p, bins, 0, 1, n)
    got, st_g = kernel_pipeline(lib, spec, p, bins, 0, 1, n)
    assert want[5 * bins.nr] > 0
    same_histogram(got, want, bins.nr)
    for k in ("rays_traced", "steps_total", "rays_strict_side", "rk45_attempts", "rk45_rejects", "longest_ray_steps", "longest_ray_steps_strict_side"):
        assert st_g[k] == st_w[k], (k, st_g[k], st_w[k])


def test_kernel_pipeline_shards_add_up_and_match_the_oracle(krlib):
    lib = krlib
    case = gc.cases()["ps_h10"]
    spec = case["source"]
    n, _, _ = api.pointsource_count(spec)
    bins = gc.emis_bins(spec)
    p = capi.copy_params(case["runs"]["rk4"], flags=capi.FLAG_HYBRID)
    whole, _ = kernel_pipeline(lib, spec, p, bins, 0, 1, n)
    parts = np.zeros_like(whole)
    for r in range(3):                                               # ray-cyclic shards: first = r, stride = 3
        cnt = (n - r + 2) // 3
        h, _ = kernel_pipeline(lib, spec, p, bins, r, 3, cnt)
        parts += h
    same_histogram(parts, whole, bins.nr)
    # against the CPU oracle's whole pipeline on the golden fixture's own rays
    g = np.load(gc.golden_path("ps_h10"))
    from test_gpu_parity import oracle_reduce_emissivity
    want = oracle_reduce_emissivity(bins, g["final__rk4"])
    assert np.array_equal(whole[:bins.nr].astype(np.int64), want["count"])
    for k, key in ((1, "flux"), (2, "emis"), (3, "sum_redshift"), (4, "sum_time")):
        m = want["count"] > 0
        np.testing.assert_allclose(whole[k * bins.nr:(k + 1) * bins.nr][m], want[key][m], rtol=1e-6)


def test_kernel_pipeline_refuses_what_it_cannot_do(krlib):
    lib = krlib
    spec = bench.make_spec(capi, bench.grid_spacing_for(1e4))
    n, _, _ = api.pointsource_count(spec)
    bins = bench.emis_bins(capi, lib.kr_kerr_isco(bench.SPIN, 1), n)
    p = capi.copy_params(capi.default_params(bench.SPIN), stop_kind=capi.STOP_FLATDISC, stop_params=(math.pi / 2,))
    p.integrator = capi.RK4
    d_h = _hist(lib, bins.nr)
    st = capi.Stats()
    rc = lib.kr_emissivity_pipeline_dev_f64(C.byref(spec), 0, 1, n, 0.0, 0, 0, C.byref(p), spec.spin, -1.0, 0, 0, 0, C.byref(bins), d_h, None, C.byref(st))
    assert rc == capi.KR_EINVAL and b"theta-limit" in lib.kr_last_error()
    rc = lib.kr_emissivity_pipeline_dev_f64(C.byref(spec), 0, 1, n, 0.0, 0, 0, C.byref(capi.default_params(bench.SPIN)), spec.spin, -1.0, 0, 0, 0, C.byref(bins), None, None, None)
    assert rc == capi.KR_EINVAL
    lib.kr_free(d_h)
