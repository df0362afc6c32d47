"""GPU: attached host arrays (kr_host_attach / kr_host_detach, include/kr_trace.h) -- what the class-API mirror does with
Raytracer<T>::rays: page-locked in place, one device residency, only the modified field copied back per pass.  Attaching must
never change a result: every host-pointer entry point gives, on an attached array, bit for bit what it gives on a plain one."""
import ctypes as C

import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as ol
from raytrace_cpu_amd import api, capi

pytestmark = pytest.mark.gpu


def _pipeline(rays, p):
    api.redshift_start(gc.SPIN, 0.0, 0, 0, rays)
    out, st = api.trace(p, rays, inplace=True)
    api.range_phi(rays)
    api.redshift(gc.SPIN, -1.0, 0, 0, rays)
    return st


def test_attached_array_results_are_bit_identical(krlib):
    lib = krlib
    spec = ol.pointsource_spec([0.0, 10.0, 1e-3, 1.5707], 0.0, gc.SPIN, 0.05, 0.05, cosalpha0=-0.995, cosalphamax=0.995, beta0=-np.pi, betamax=np.pi)
    init = api.pointsource_init(spec)
    assert len(init) >= 4096                      # large enough for the one-field write-back path
    p = capi.default_params(gc.SPIN)
    p.integrator, p.flags = capi.RK4, capi.FLAG_HYBRID
    plain, att = init.copy(), init.copy()
    sentinel = 12345.678
    for a in (plain, att):
        a["alpha"][:] = a["alpha"]                # untouched fields must survive a partial write-back ...
        a["pt"][7] = sentinel                     # ... including a host-side edit made between passes (the array is re-uploaded per call)
    capi.check(lib, lib.kr_host_attach(att.ctypes.data_as(C.c_void_p), len(att), capi.RAY_F64.itemsize), "attach")
    try:
        assert lib.kr_host_attach(att.ctypes.data_as(C.c_void_p), len(att), capi.RAY_F64.itemsize) == capi.KR_EINVAL      # twice: refused
        st_p = _pipeline(plain, p)
        st_a = _pipeline(att, p)
        assert st_p["steps_total"] == st_a["steps_total"] and st_a["rays_traced"] == st_p["rays_traced"] > 4000
        assert ol.rays_equal_bitwise(att, plain) == []
        # an edit on the host between two passes is seen by the next pass (the host array is the input of every call)
        for a in (plain, att):
            a["phi"][11] = 100.0
            api.range_phi(a)
        assert att["phi"][11] == plain["phi"][11] != 100.0
        api.calculate_momentum(gc.SPIN, plain)
        api.calculate_momentum(gc.SPIN, att)
        assert ol.rays_equal_bitwise(att, plain) == []
        # a sub-range of an attached array (what the single-ray propagate() forms pass)
        sub_p, sub_a = init[200:264].copy(), None
        api.trace(p, sub_p, inplace=True)
        att[200:264] = init[200:264]
        view = att[200:264]
        api.trace(p, view, inplace=True)
        assert ol.rays_equal_bitwise(view, sub_p) == []
    finally:
        capi.check(lib, lib.kr_host_detach(att.ctypes.data_as(C.c_void_p)), "detach")
    assert lib.kr_host_detach(att.ctypes.data_as(C.c_void_p)) == capi.KR_OK          # detaching an unknown array is a no-op
    # after detaching the array is an ordinary one again
    again = init.copy()
    _pipeline(again, p)
    fresh = init.copy()
    _pipeline(fresh, p)
    assert ol.rays_equal_bitwise(again, fresh) == []
