"""GPU: the batched O(N) passes of the multi-radius drivers (kr_pointsource_init_emit_batch_dev_f64, kr_post_return_batch_dev_f64: all radii of
disc_source_photonfrac_r.cpp:74-126 in a handful of launches) against the same number of single calls: ray records bit for bit, weighted sums
up to the order of the additions."""
import ctypes as C
import math

import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as ol
from raytrace_cpu_amd import api, capi

pytestmark = pytest.mark.gpu
vp = C.c_void_p


def _specs():
    out = []
    for k, r in enumerate([1.3, 2.0, 3.7, 6.0, 11.0, 40.0, 150.0] * 6):              # 42 sources: more than one chunk of either kernel
        d = 0.02 + 0.002 * (k % 5)
        V = 1.0 / (gc.SPIN + r ** 1.5)
        out.append(ol.pointsource_spec([0.0, r, math.pi / 2 - 1e-6, 1.5707], V, gc.SPIN, d, d * math.pi, cosalpha0=-0.995, cosalphamax=0.995, beta0=0.0, betamax=math.pi))
    return out


def test_batched_source_and_return_passes_equal_the_single_calls(krlib):
    lib = krlib
    specs = _specs()
    k = len(specs)
    counts = [api.pointsource_count(s)[0] for s in specs]
    counts[5] = 0                                                   # an empty launch in the middle of a chunk
    bufs = [[C.c_void_p() for _ in range(k)] for _ in range(2)]
    outs = [C.c_void_p() for _ in range(2)]
    try:
        for side in range(2):
            for j in range(k):
                capi.check(lib, lib.kr_malloc(C.byref(bufs[side][j]), max(counts[j], 1) * 144), "malloc")
            capi.check(lib, lib.kr_malloc(C.byref(outs[side]), k * 32), "malloc")
            capi.check(lib, lib.kr_memset(outs[side], 0, k * 32), "memset")
        bins = (capi.ReturnBins * k)()
        for j, s in enumerate(specs):
            b = bins[j]
            b.r_isco, b.r_disc, b.r_esc, b.source_r, b.source_phi = gc.r_isco(), 500.0, 550.0, s.pos[1], 1.5707
            b.plane_iso, b.limb, b.weight_norm, b.pad = 1, j % 2, 1, 0
        p = capi.default_params(gc.SPIN)
        p.integrator, p.r_max, p.flags = capi.EULER, 550.0, capi.FLAG_HYBRID
        # single calls
        for j, s in enumerate(specs):
            capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(s), 0, 1, s.V, 0, 0, bufs[0][j], counts[j], None), "init")
        # one batch
        arr_specs = (capi.PointSourceSpec * k)(*specs)
        V = (C.c_double * k)(*[s.V for s in specs])
        ptrs = (C.c_void_p * k)(*[b.value for b in bufs[1]])
        ns = (C.c_int64 * k)(*counts)
        capi.check(lib, lib.kr_pointsource_init_emit_batch_dev_f64(k, arr_specs, V, 0, 0, ptrs, ns, None), "init batch")
        capi.check(lib, lib.kr_synchronize(None), "sync")

        def fetch(d, n):
            h = np.zeros(n, dtype=capi.RAY_F64)
            if n:
                capi.check(lib, lib.kr_memcpy_d2h(h.ctypes.data_as(vp), d, h.nbytes), "d2h")
            return h
        for j in range(k):
            assert not ol.rays_equal_bitwise(fetch(bufs[0][j], counts[j]), fetch(bufs[1][j], counts[j])), j
        # V = NULL takes the specs' own velocities
        capi.check(lib, lib.kr_memset(bufs[1][3], 0, counts[3] * 144), "memset")
        capi.check(lib, lib.kr_pointsource_init_emit_batch_dev_f64(k, arr_specs, None, 0, 0, ptrs, ns, None), "init batch, V from the specs")
        capi.check(lib, lib.kr_synchronize(None), "sync")
        assert not ol.rays_equal_bitwise(fetch(bufs[0][3], counts[3]), fetch(bufs[1][3], counts[3]))
        # trace both sets the same way, then the return pass: singles against one batch
        for side in range(2):
            for j in range(k):
                if counts[j]:
                    api.trace_dev(p, bufs[side][j].value, counts[j])
        for j in range(k):
            capi.check(lib, lib.kr_post_return_dev_f64(-math.pi, math.pi, C.byref(bins[j]), bufs[0][j], counts[j], vp(outs[0].value + 32 * j), None), "post")
        out_ptrs = (C.c_void_p * k)(*[outs[1].value + 32 * j for j in range(k)])
        capi.check(lib, lib.kr_post_return_batch_dev_f64(k, -math.pi, math.pi, bins, ptrs, ns, out_ptrs, None), "post batch")
        capi.check(lib, lib.kr_synchronize(None), "sync")
        t = [np.zeros(4 * k) for _ in range(2)]
        for side in range(2):
            capi.check(lib, lib.kr_memcpy_d2h(t[side].ctypes.data_as(vp), outs[side], t[side].nbytes), "d2h")
        assert t[0].reshape(k, 4)[:, 0].min() == 0 and (t[0].reshape(k, 4)[:, 0] > 0).sum() == k - 1          # the empty launch adds nothing
        np.testing.assert_allclose(t[1], t[0], rtol=1e-12, atol=0)
        for j in range(k):                                           # range_phi ran inside both: the records agree bit for bit afterwards too
            assert not ol.rays_equal_bitwise(fetch(bufs[0][j], counts[j]), fetch(bufs[1][j], counts[j])), j
        # arguments
        assert lib.kr_post_return_batch_dev_f64(k, -math.pi, math.pi, bins, ptrs, ns, None, None) == capi.KR_EINVAL
        assert lib.kr_pointsource_init_emit_batch_dev_f64(k, None, V, 0, 0, ptrs, ns, None) == capi.KR_EINVAL
        assert lib.kr_pointsource_init_emit_batch_dev_f64(0, None, None, 0, 0, None, None, None) == 0
    finally:
        for side in range(2):
            for j in range(k):
                if bufs[side][j]:
                    lib.kr_free(bufs[side][j])
            if outs[side]:
                lib.kr_free(outs[side])
