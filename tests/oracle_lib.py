"""TEST INFRASTRUCTURE: ctypes loaders for the two CPU checkers under oracle/.

  oracle()  -> oracle/libkr_oracle.so   (C restatement; built on demand with oracle/Makefile)
  ref()     -> oracle/_ref/libkr_ref.so (the reference's own sources compiled with a shim; None when
               neither the prebuilt .so nor /root/reference is available)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytrace_cpu_amd import capi  # noqa: E402  (interface structs only)

ORACLE_SO = os.path.join(ROOT, "oracle", "libkr_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libkr_ref.so")

P = C.POINTER
_vp, _i64, _i32, _dbl, _int = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int


def build_oracle(force=False):
    src = os.path.join(ROOT, "oracle", "kr_oracle.c")
    stale = (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), ORACLE_SO])
    return ORACLE_SO


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        protos = {
            "kro_kerr_horizon": (_dbl, [_dbl]),
            "kro_kerr_isco": (_dbl, [_dbl, _int]),
            "kro_disc_velocity": (_dbl, [_dbl, _dbl, _int]),
            "kro_trace_f64": (_int, [P(capi.Params), _vp, _i64, _int, P(capi.Stats)]),
            "kro_redshift_start_f64": (None, [_dbl, _dbl, _int, _int, _vp, _i64]),
            "kro_redshift_f64": (None, [_dbl, _dbl, _int, _int, _int, _vp, _i64]),
            "kro_redshift_dest_f64": (None, [_dbl, _int, _vp, _i64]),
            "kro_range_phi_f64": (None, [_dbl, _dbl, _vp, _i64]),
            "kro_calculate_momentum_f64": (None, [_dbl, _vp, _i64]),
            "kro_pointsource_count": (_i64, [P(capi.PointSourceSpec), P(_i32), P(_i32)]),
            "kro_pointsource_init_f64": (_int, [P(capi.PointSourceSpec), _vp, _i64]),
            "kro_imageplane_count": (_i64, [P(capi.ImagePlaneSpec), P(_i32), P(_i32)]),
            "kro_imageplane_init_f64": (_int, [P(capi.ImagePlaneSpec), _vp, _i64]),
            "kro_reduce_emissivity_f64": (None, [P(capi.EmisBins), _vp, _i64, _vp, _vp, _vp, _vp, _vp, P(_i64)]),
            "kro_reduce_image_f64": (None, [P(capi.ImageBins), _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, P(_i64)]),
            "kro_reduce_return_f64": (None, [P(capi.ReturnBins), _vp, _i64, P(_dbl * 4)]),
            "kro_max_threads": (_int, []),
        }
        for name, (res, args) in protos.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _oracle = lib
    return _oracle


_ref = False


def ref():
    """The compiled reference, or None.  Never built on the GPU box (no /root/reference there)."""
    global _ref
    if _ref is False:
        if not os.path.exists(REF_SO) and os.path.exists("/root/reference/src/raytracer/raytracer.cpp"):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
        if not os.path.exists(REF_SO):
            _ref = None
            return None
        lib = C.CDLL(REF_SO)
        protos = {
            "ref_sizeof_ray": (_int, []),
            "ref_pointsource_new": (_vp, [P(_dbl), _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl]),
            "ref_imageplane_new": (_vp, [_dbl] * 11),
            "ref_free": (None, [_vp]),
            "ref_count": (_int, [_vp]),
            "ref_rays": (_vp, [_vp]),
            "ref_set_rk45_tol": (None, [_vp, _dbl]),
            "ref_set_max_tstep": (None, [_vp, _dbl, _dbl]),
            "ref_set_max_phistep": (None, [_vp, _dbl]),
            "ref_set_boundary": (None, [_vp, _dbl]),
            "ref_redshift_start": (None, [_vp, _dbl, _int, _int]),
            "ref_redshift_start_source": (None, [_vp]),
            "ref_run_thetalim": (None, [_vp, _int, _dbl, _dbl, _int]),
            "ref_run_dest": (_int, [_vp, _int, _int, P(_dbl), _dbl, _int]),
            "ref_redshift": (None, [_vp, _dbl, _int, _int, _int]),
            "ref_redshift_dest": (_int, [_vp, _int, P(_dbl), _int]),
            "ref_range_phi": (None, [_vp, _dbl, _dbl]),
            "ref_calculate_momentum": (None, [_vp]),
            "ref_kerr_horizon": (_dbl, [_dbl]),
            "ref_kerr_isco": (_dbl, [_dbl, _int]),
            "ref_disc_velocity": (_dbl, [_dbl, _dbl, _int]),
        }
        for name, (res, args) in protos.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        assert lib.ref_sizeof_ray() == capi.RAY_F64.itemsize
        _ref = lib
    return _ref


# ---------------------------------------------------------------------------------------------------
# convenience wrappers shared by the tests


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pointsource_spec(pos, V, spin, dcosalpha, dbeta, cosalpha0=-0.999999, cosalphamax=0.995, beta0=-0.995 * np.pi,
                     betamax=np.pi, E=1.0, tol=100.0):
    s = capi.PointSourceSpec()
    for i in range(4):
        s.pos[i] = pos[i]
    s.V, s.spin, s.tol, s.dcosalpha, s.dbeta = V, spin, tol, dcosalpha, dbeta
    s.cosalpha0, s.cosalphamax, s.beta0, s.betamax, s.E = cosalpha0, cosalphamax, beta0, betamax, E
    return s


def imageplane_spec(dist, inc_deg, x0, xmax, dx, y0, ymax, dy, spin, phi0=0.0, precision=100.0):
    s = capi.ImagePlaneSpec()
    s.dist, s.inc_deg, s.x0, s.xmax, s.dx, s.y0, s.ymax, s.dy = dist, inc_deg, x0, xmax, dx, y0, ymax, dy
    s.spin, s.phi0, s.precision = spin, phi0, precision
    return s


def oracle_pointsource(spec):
    lib = oracle()
    n = lib.kro_pointsource_count(C.byref(spec), None, None)
    rays = np.zeros(n, dtype=capi.RAY_F64)
    rc = lib.kro_pointsource_init_f64(C.byref(spec), ptr(rays), n)
    assert rc == 0
    return rays


def oracle_imageplane(spec):
    lib = oracle()
    n = lib.kro_imageplane_count(C.byref(spec), None, None)
    rays = np.zeros(n, dtype=capi.RAY_F64)
    rc = lib.kro_imageplane_init_f64(C.byref(spec), ptr(rays), n)
    assert rc == 0
    return rays


def oracle_trace(params, rays, nthreads=0):
    """Traces a COPY of rays through the oracle; returns (rays_out, stats dict)."""
    out = rays.copy()
    st = capi.Stats()
    rc = oracle().kro_trace_f64(C.byref(params), ptr(out), len(out), nthreads, C.byref(st))
    assert rc == 0, rc
    return out, st.as_dict()


class RefSource:
    """A reference PointSource<double> / ImagePlane<double> object behind the shim."""

    def __init__(self, spec):
        self.lib = ref()
        assert self.lib is not None, "compiled reference not available"
        if isinstance(spec, capi.PointSourceSpec):
            pos = (C.c_double * 4)(*spec.pos)
            self.h = self.lib.ref_pointsource_new(pos, spec.V, spec.spin, spec.tol, spec.dcosalpha, spec.dbeta,
                                                  spec.cosalpha0, spec.cosalphamax, spec.beta0, spec.betamax, spec.E)
            self.is_imageplane = False
        else:
            self.h = self.lib.ref_imageplane_new(spec.dist, spec.inc_deg, spec.x0, spec.xmax, spec.dx, spec.y0,
                                                 spec.ymax, spec.dy, spec.spin, spec.phi0, spec.precision)
            self.is_imageplane = True
        self.n = self.lib.ref_count(self.h)
        buf = (C.c_char * (self.n * capi.RAY_F64.itemsize)).from_address(self.lib.ref_rays(self.h))
        self.rays = np.frombuffer(buf, dtype=capi.RAY_F64)   # a VIEW of the reference's rays[]
        self.sanitize()

    def sanitize(self):
        """Give the fields the reference leaves indeterminate (`new Ray<T>[n]`) a defined value: records of
        never-initialised rays are zeroed (steps=-1 kept); ImagePlane never sets rdot_flips/equatorial_crossings."""
        dead = self.rays["steps"] == -1
        z = np.zeros(1, dtype=capi.RAY_F64)
        z["steps"] = -1
        self.rays[dead] = z
        if self.is_imageplane:
            self.rays["rdot_flips"] = 0
            self.rays["equatorial_crossings"] = 0
        live = ~dead
        for f in ("emit", "redshift"):
            self.rays[f][live] = 0.0

    def snapshot(self):
        return self.rays.copy()

    def run(self, params):
        lib = self.lib
        lib.ref_set_rk45_tol(self.h, params.rk45_tol)
        lib.ref_set_max_tstep(self.h, params.max_tstep, params.maxtstep_rlim)
        lib.ref_set_max_phistep(self.h, params.max_phistep)
        if params.stop_kind == capi.STOP_THETA:
            lib.ref_run_thetalim(self.h, params.integrator, params.theta_max, params.r_max, params.steplim)
        else:
            sp = (C.c_double * 4)(*params.stop_params)
            rc = lib.ref_run_dest(self.h, params.integrator, params.stop_kind, sp, params.r_max, params.steplim)
            assert rc == 0

    def close(self):
        if self.h:
            self.rays = None
            self.lib.ref_free(self.h)
            self.h = None


def rays_equal_bitwise(a, b, fields=None):
    """Field-by-field bitwise comparison; a NaN matches any NaN (x86 propagates the sign/payload of whichever
    operand the compiler happened to place first, which is not part of the algorithm).  Returns the bad fields."""
    bad = []
    for f in (fields or a.dtype.names):
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            same = (x.view(np.int64) == y.view(np.int64)) | (np.isnan(x) & np.isnan(y))
        else:
            same = x == y
        if not same.all():
            bad.append((f, int((~same).sum()), int(np.flatnonzero(~same)[0])))
    return bad
