"""CPU, build container only: the oracle restatement against the LIVE compiled reference
(oracle/_ref/libkr_ref.so) on the reference tests' full grids.  Skipped where the
compiled reference is not available."""
import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as ol
from raytrace_cpu_amd import capi

pytestmark = pytest.mark.skipif(ol.ref() is None, reason="compiled reference (oracle/_ref) not available")


def _both(spec, params, start):
    src = ol.RefSource(spec)
    src.lib.ref_redshift_start(src.h, *start)
    src.run(params)
    ref_out = src.snapshot()
    src.close()
    rays = ol.oracle_imageplane(spec) if isinstance(spec, capi.ImagePlaneSpec) else ol.oracle_pointsource(spec)
    ol.oracle().kro_redshift_start_f64(params.spin, *start, ol.ptr(rays), len(rays))
    out, _ = ol.oracle_trace(params, rays)
    return ref_out, out


@pytest.mark.parametrize("method", [capi.EULER, capi.RK4, capi.RK45])
@pytest.mark.parametrize("h", [5.0, 10.0])
def test_perf_test_grid_bitwise(method, h):
    # integrator_perf_test.cpp:35-45 grid (5167 allocated rays), at the reference's h=5 and BASELINE's h=10
    spec = ol.pointsource_spec([0.0, h, 1e-3, 0.0], 0.0, gc.SPIN, 0.05, 0.05, cosalpha0=-0.995, cosalphamax=0.995,
                               beta0=-np.pi, betamax=np.pi)
    p = capi.default_params(gc.SPIN)
    p.integrator = method
    ref_out, out = _both(spec, p, (0.0, 0, 0))
    assert len(out) == 5167
    assert ol.rays_equal_bitwise(ref_out, out) == []


@pytest.mark.parametrize("method", [capi.EULER, capi.RK4])
def test_imageplane_33_bitwise(method):
    spec = ol.imageplane_spec(10000.0, 80.0, -30.0, 30.0, 60.0 / 32, -30.0, 30.0, 60.0 / 32, gc.SPIN)
    p = capi.default_params(-gc.SPIN)
    p.integrator, p.r_max = method, 11000.0
    ref_out, out = _both(spec, p, (0.0, 1, 0))
    assert len(out) == 33 * 33
    assert ol.rays_equal_bitwise(ref_out, out) == []


@pytest.mark.parametrize("tol", [1e-6, 1e-10])
def test_rk45_tolerance_bitwise(tol):
    # emissivity_rk45_tol_sweep.py:38 end points
    spec = ol.pointsource_spec([0.0, 5.0, 1e-3, 0.0], 0.0, gc.SPIN, 0.1, 0.1, cosalpha0=-0.995, cosalphamax=0.995,
                               beta0=-np.pi, betamax=np.pi)
    p = capi.default_params(gc.SPIN)
    p.integrator, p.rk45_tol = capi.RK45, tol
    ref_out, out = _both(spec, p, (0.0, 0, 0))
    assert ol.rays_equal_bitwise(ref_out, out) == []
