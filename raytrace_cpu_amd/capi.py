"""ctypes mirror of include/kr_trace.h (struct layouts, constants, prototypes).

Pure interface definitions: nothing here computes.  `load()` opens the HIP shared library
(raytrace_cpu_amd/csrc/libkrtrace.so) and fails loudly when it is missing -- there is no CPU
fallback in the product path.
"""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 15

KR_OK, KR_EINVAL, KR_ENODEVICE, KR_EHIP, KR_ENOMEM = 0, -1, -2, -3, -4
EULER, RK4, RK45 = 0, 1, 2
STOP_THETA, STOP_FLATDISC, STOP_DISC_ISCO, STOP_FLATPLANE = 0, 1, 2, 3
STATUS_DEST, STATUS_HORIZON, STATUS_RLIM, STATUS_STEPLIM = 1, 2, 4, 8
STATUS_ERGO, STATUS_NEG_ENERGY, STATUS_NAN = 16, 32, 64
STEPLIM, RK45_STEPLIM, MIN_STEP = 10_000_000, 100_000, 1e-3
FLAG_FAST_MATH = 1
FLAG_HYBRID = 2
FLAG_RK45_ITERATE_ALL = 4

# Ray<double> / Ray<float>  (reference src/raytracer/raytracer.h:65-78)
_F64 = [(n, "<f8") for n in ("t", "r", "theta", "phi", "pt", "pr", "ptheta", "pphi", "k", "h", "Q", "emit", "redshift")]
_I32 = [(n, "<i4") for n in ("steps", "status", "rdot_sign", "thetadot_sign", "rdot_flips", "equatorial_crossings")]
RAY_F64 = np.dtype(_F64 + _I32 + [("alpha", "<f8"), ("beta", "<f8")], align=True)
RAY_F32 = np.dtype([(n, "<f4") for n, _ in _F64] + _I32 + [("alpha", "<f4"), ("beta", "<f4")], align=True)
assert RAY_F64.itemsize == 144 and RAY_F32.itemsize == 84


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("spin", "horizon", "precision", "theta_precision", "max_tstep",
                                          "maxtstep_rlim", "max_phistep", "rk45_tol", "r_max", "theta_max")] + \
               [("stop_params", C.c_double * 4)] + \
               [(n, C.c_int32) for n in ("integrator", "stop_kind", "steplim", "flags")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("rays_total", "rays_traced", "steps_total", "rk45_attempts", "rk45_rejects")] + \
               [(n, C.c_double) for n in ("kernel_ms", "h2d_ms", "d2h_ms")] + [("rays_strict_side", C.c_int64), ("rk45_stationary_steps", C.c_int64), ("rk45_extrapolated_steps", C.c_int64)] + \
               [("strict_side_ms", C.c_double), ("main_ms", C.c_double), ("longest_ray_steps", C.c_int64), ("longest_ray_steps_strict_side", C.c_int64),
                ("steps_strict_side", C.c_int64), ("rk45_evaluated_strict_side", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PointSourceSpec(C.Structure):
    _fields_ = [("pos", C.c_double * 4)] + \
               [(n, C.c_double) for n in ("V", "spin", "tol", "dcosalpha", "dbeta", "cosalpha0", "cosalphamax",
                                          "beta0", "betamax", "E")]


class ImagePlaneSpec(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("dist", "inc_deg", "x0", "xmax", "dx", "y0", "ymax", "dy", "spin", "phi0",
                                          "precision")]


class EmisBins(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("r_min", "dr", "r_isco", "gamma", "spin", "num_primary_rays")] + \
               [("nr", C.c_int32), ("logbin", C.c_int32)]


class ImageBins(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("x0", "y0", "img_dx", "img_dy", "r_isco", "r_disc", "q1", "rb1", "q2", "rb2", "q3")] + \
               [(n, C.c_int32) for n in ("img_nx", "img_ny", "flip_image", "pad")]


class ReturnBins(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("r_isco", "r_disc", "r_esc", "source_r", "source_phi")] + \
               [(n, C.c_int32) for n in ("plane_iso", "limb", "weight_norm", "pad")]


def default_params(spin, horizon=None):
    """Raytracer<T> ctor defaults (reference src/raytracer/raytracer.cpp:12-22, raytracer.h:19-44)."""
    p = Params()
    p.spin = spin
    p.horizon = horizon if horizon is not None else 1.0 + np.sqrt((1.0 - spin) * (1.0 + spin))
    p.precision, p.theta_precision = 100.0, 50.0
    p.max_tstep, p.maxtstep_rlim, p.max_phistep = 1.0, 100.0, 0.1
    p.rk45_tol = 1e-8
    p.r_max, p.theta_max = 1000.0, np.pi / 2
    p.integrator, p.stop_kind, p.steplim, p.flags = EULER, STOP_THETA, -1, 0
    return p


def copy_params(p, **kw):
    q = Params()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(Params))
    for k, v in kw.items():
        if k == "stop_params":
            for i, x in enumerate(v):
                q.stop_params[i] = x
        else:
            setattr(q, k, v)
    return q


P = C.POINTER
_vp, _i64, _i32, _dbl, _int = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int
PROGRESS_FN = C.CFUNCTYPE(None, C.c_int64, C.c_int64, C.c_void_p)      # kr_progress_fn

# name -> (restype, argtypes); exactly the entry points include/kr_trace.h declares
PROTOTYPES = {
    "kr_abi_version": (_int, []),
    "kr_last_error": (C.c_char_p, []),
    "kr_device_count": (_int, []),
    "kr_set_device": (_int, [_int]),
    "kr_device_info": (_int, [P(_int), P(_int), P(_i64), C.c_char_p, _int]),
    "kr_params_default": (None, [P(Params), _dbl]),
    "kr_kerr_horizon": (_dbl, [_dbl]),
    "kr_kerr_isco": (_dbl, [_dbl, _int]),
    "kr_disc_velocity": (_dbl, [_dbl, _dbl, _int]),
    "kr_pointsource_count": (_i64, [P(PointSourceSpec), P(_i32), P(_i32)]),
    "kr_imageplane_count": (_i64, [P(ImagePlaneSpec), P(_i32), P(_i32)]),
    "kr_trace_f64": (_int, [P(Params), _vp, _i64, P(Stats)]),
    "kr_trace_f32": (_int, [P(Params), _vp, _i64, P(Stats)]),
    "kr_trace_dev_f64": (_int, [P(Params), _vp, _i64, _vp, P(Stats)]),
    "kr_trace_dev_f32": (_int, [P(Params), _vp, _i64, _vp, P(Stats)]),
    "kr_trace_async_f64": (_int, [P(Params), _vp, _i64, _vp, P(_vp)]),
    "kr_trace_async_f32": (_int, [P(Params), _vp, _i64, _vp, P(_vp)]),
    "kr_trace_batch_async_f64": (_int, [_i32, P(P(Params)), P(_vp), P(_i64), P(_vp), P(_vp)]),
    "kr_trace_wait": (_int, [_vp, P(Stats)]),
    "kr_trace_wait_many": (_int, [_i32, P(_vp), P(Stats), P(Stats)]),
    "kr_trace_release": (_int, [_vp]),
    "kr_redshift_start_f64": (_int, [_dbl, _dbl, _int, _int, _vp, _i64]),
    "kr_redshift_start_dev_f64": (_int, [_dbl, _dbl, _int, _int, _vp, _i64, _vp]),
    "kr_redshift_f64": (_int, [_dbl, _dbl, _int, _int, _int, _vp, _i64]),
    "kr_redshift_dev_f64": (_int, [_dbl, _dbl, _int, _int, _int, _vp, _i64, _vp]),
    "kr_redshift_dest_f64": (_int, [_dbl, _int, _vp, _i64]),
    "kr_redshift_dest_dev_f64": (_int, [_dbl, _int, _vp, _i64, _vp]),
    "kr_range_phi_f64": (_int, [_dbl, _dbl, _vp, _i64]),
    "kr_range_phi_dev_f64": (_int, [_dbl, _dbl, _vp, _i64, _vp]),
    "kr_calculate_momentum_f64": (_int, [_dbl, _vp, _i64]),
    "kr_calculate_momentum_dev_f64": (_int, [_dbl, _vp, _i64, _vp]),
    "kr_redshift_start_f32": (_int, [_dbl, _dbl, _int, _int, _vp, _i64]),
    "kr_redshift_start_dev_f32": (_int, [_dbl, _dbl, _int, _int, _vp, _i64, _vp]),
    "kr_redshift_f32": (_int, [_dbl, _dbl, _int, _int, _int, _vp, _i64]),
    "kr_redshift_dev_f32": (_int, [_dbl, _dbl, _int, _int, _int, _vp, _i64, _vp]),
    "kr_redshift_dest_f32": (_int, [_dbl, _int, _vp, _i64]),
    "kr_redshift_dest_dev_f32": (_int, [_dbl, _int, _vp, _i64, _vp]),
    "kr_range_phi_f32": (_int, [_dbl, _dbl, _vp, _i64]),
    "kr_range_phi_dev_f32": (_int, [_dbl, _dbl, _vp, _i64, _vp]),
    "kr_calculate_momentum_f32": (_int, [_dbl, _vp, _i64]),
    "kr_calculate_momentum_dev_f32": (_int, [_dbl, _vp, _i64, _vp]),
    "kr_pointsource_init_f64": (_int, [P(PointSourceSpec), _vp, _i64]),
    "kr_pointsource_init_dev_f64": (_int, [P(PointSourceSpec), _vp, _i64, _vp]),
    "kr_imageplane_init_f64": (_int, [P(ImagePlaneSpec), _vp, _i64]),
    "kr_imageplane_init_dev_f64": (_int, [P(ImagePlaneSpec), _vp, _i64, _vp]),
    "kr_pointsource_init_strided_dev_f64": (_int, [P(PointSourceSpec), _i64, _i64, _vp, _i64, _vp]),
    "kr_imageplane_init_strided_dev_f64": (_int, [P(ImagePlaneSpec), _i64, _i64, _vp, _i64, _vp]),
    "kr_reduce_emissivity_f64": (_int, [P(EmisBins), _vp, _i64, _vp, _vp, _vp, _vp, _vp, P(_i64)]),
    "kr_reduce_emissivity_dev_f64": (_int, [P(EmisBins), _vp, _i64, _vp, _vp]),
    "kr_pointsource_init_emit_dev_f64": (_int, [P(PointSourceSpec), _i64, _i64, _dbl, _int, _int, _vp, _i64, _vp]),
    "kr_imageplane_init_emit_dev_f64": (_int, [P(ImagePlaneSpec), _i64, _i64, _dbl, _int, _int, _vp, _i64, _vp]),
    "kr_imageplane_init_emit_runs_dev_f64": (_int, [P(ImagePlaneSpec), _i64, _i64, _i64, _dbl, _int, _int, _vp, _i64, _vp]),
    "kr_post_image_dev_f64": (_int, [_dbl, _dbl, _int, _int, _int, _dbl, _dbl, P(ImageBins), _vp, _i64, _vp, _vp]),
    "kr_trace_poll": (_int, [_vp, P(_i64), P(_i32)]),
    "kr_trace_progress_f64": (_int, [P(Params), _vp, _i64, P(Stats), _i64, PROGRESS_FN, _vp]),
    "kr_trace_progress_f32": (_int, [P(Params), _vp, _i64, P(Stats), _i64, PROGRESS_FN, _vp]),
    "kr_pointsource_tables": (_int, [P(PointSourceSpec), _vp, _vp, _vp]),
    "kr_post_emissivity_dev_f64": (_int, [_dbl, _dbl, _int, _int, _int, _dbl, _dbl, P(EmisBins), _vp, _i64, _vp, _vp]),
    "kr_reduce_image_f64": (_int, [P(ImageBins), _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, P(_i64)]),
    "kr_reduce_image_dev_f64": (_int, [P(ImageBins), _vp, _i64, _vp, _vp]),
    "kr_reduce_return_f64": (_int, [P(ReturnBins), _vp, _i64, P(_dbl * 4)]),
    "kr_reduce_return_dev_f64": (_int, [P(ReturnBins), _vp, _i64, _vp, _vp]),
    "kr_post_return_dev_f64": (_int, [_dbl, _dbl, P(ReturnBins), _vp, _i64, _vp, _vp]),
    "kr_post_return_batch_dev_f64": (_int, [_i32, _dbl, _dbl, P(ReturnBins), P(_vp), P(_i64), P(_vp), _vp]),
    "kr_pointsource_init_emit_batch_dev_f64": (_int, [_i32, P(PointSourceSpec), P(_dbl), _int, _int, P(_vp), P(_i64), _vp]),
    "kr_debug_arith_f64": (_int, [_int, _vp, _vp, _vp, _i64]),
    "kr_host_attach": (_int, [_vp, _i64, _i32]),
    "kr_host_detach": (_int, [_vp]),
    "kr_malloc": (_int, [P(_vp), _i64]),
    "kr_free": (_int, [_vp]),
    "kr_host_alloc": (_int, [P(_vp), _i64]),
    "kr_host_free": (_int, [_vp]),
    "kr_memcpy_h2d": (_int, [_vp, _vp, _i64]),
    "kr_memcpy_d2h": (_int, [_vp, _vp, _i64]),
    "kr_memset": (_int, [_vp, _int, _i64]),
    "kr_synchronize": (_int, [_vp]),
    "kr_stream_create": (_int, [P(_vp)]),
    "kr_stream_destroy": (_int, [_vp]),
    "kr_configure_process": (_int, []),
    "kr_shutdown": (_int, []),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libkrtrace.so")


class KrError(RuntimeError):
    pass


def load(path=None):
    """dlopen libkrtrace.so and attach prototypes.  Raises if the library or any declared symbol is missing."""
    path = path or os.environ.get("KRTRACE_LIB") or LIB_PATH      # KRTRACE_LIB: an experiment build (scripts/ab_kernels.py)
    if not os.path.exists(path):
        raise KrError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise KrError(f"{path} does not export {name} (declared in include/kr_trace.h)") from e
        fn.restype, fn.argtypes = res, args
    if lib.kr_abi_version() != ABI_VERSION:
        raise KrError(f"ABI mismatch: library {lib.kr_abi_version()} vs binding {ABI_VERSION}")
    return lib


def check(lib, rc, what):
    if rc != KR_OK:
        msg = lib.kr_last_error()
        raise KrError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
