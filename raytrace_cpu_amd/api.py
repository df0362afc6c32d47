"""Thin Python conveniences over the C ABI (include/kr_trace.h): numpy in, numpy out.

Every function goes through libkrtrace.so (HIP, gfx950).  Nothing here computes on the CPU and nothing
falls back: if the library is missing or no GPU is visible the call raises KrError.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import KrError, Params, Stats  # noqa: F401

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = capi.load()
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _rays_arg(rays, dtype):
    if not isinstance(rays, np.ndarray) or rays.dtype != dtype or not rays.flags["C_CONTIGUOUS"]:
        raise KrError(f"rays must be a C-contiguous numpy array of dtype {dtype.names and 'Ray' or dtype}")
    return rays


def device_count():
    n = lib().kr_device_count()
    if n < 0:
        raise KrError(lib().kr_last_error().decode())
    return n


def device_info():
    cus, khz, mem = C.c_int(), C.c_int(), C.c_int64()
    name = C.create_string_buffer(256)
    capi.check(lib(), lib().kr_device_info(C.byref(cus), C.byref(khz), C.byref(mem), name, 256), "kr_device_info")
    return {"name": name.value.decode(), "cu_count": cus.value, "clock_khz": khz.value, "hbm_bytes": mem.value}


def trace(params, rays, inplace=False):
    """Raytracer<double>::run_raytrace on a host array of Ray<double> records. Returns (rays, stats dict)."""
    f32 = rays.dtype == capi.RAY_F32
    _rays_arg(rays, capi.RAY_F32 if f32 else capi.RAY_F64)
    out = rays if inplace else rays.copy()
    st = Stats()
    fn = lib().kr_trace_f32 if f32 else lib().kr_trace_f64
    capi.check(lib(), fn(C.byref(params), _ptr(out), len(out), C.byref(st)), "kr_trace")
    return out, st.as_dict()


def trace_dev(params, d_ptr, n, stream=None, want_stats=True, f32=False):
    st = Stats() if want_stats else None
    fn = lib().kr_trace_dev_f32 if f32 else lib().kr_trace_dev_f64
    capi.check(lib(), fn(C.byref(params), C.c_void_p(d_ptr), n, C.c_void_p(stream or 0), C.byref(st) if st else None), "kr_trace_dev")
    return st.as_dict() if st else None


def trace_async(params, d_ptr, n, stream=None, f32=False):
    """kr_trace_async_*: enqueues the trace on `stream`, returns a ticket for trace_wait / trace_release."""
    ticket = C.c_void_p()
    fn = lib().kr_trace_async_f32 if f32 else lib().kr_trace_async_f64
    capi.check(lib(), fn(C.byref(params), C.c_void_p(d_ptr), n, C.c_void_p(stream or 0), C.byref(ticket)), "kr_trace_async")
    return ticket


def trace_batch_async(params_list, d_ptrs, ns, streams=None):
    """kr_trace_batch_async_f64: all strict side launches first, then all main launches.  Returns one ticket per trace."""
    count = len(params_list)
    pp = (C.POINTER(Params) * count)(*[C.pointer(p) for p in params_list])
    dd = (C.c_void_p * count)(*[C.c_void_p(d) for d in d_ptrs])
    nn = (C.c_int64 * count)(*ns)
    ss = (C.c_void_p * count)(*[C.c_void_p(s or 0) for s in (streams or [0] * count)])
    tt = (C.c_void_p * count)()
    capi.check(lib(), lib().kr_trace_batch_async_f64(count, pp, dd, nn, ss, tt), "kr_trace_batch_async")
    return [C.c_void_p(t) for t in tt]


def trace_wait(ticket, want_stats=True):
    st = Stats() if want_stats else None
    capi.check(lib(), lib().kr_trace_wait(ticket, C.byref(st) if st else None), "kr_trace_wait")
    return st.as_dict() if st else None


def trace_wait_many(tickets):
    """kr_trace_wait_many: waits for and releases all tickets in one call; returns the summed counters (kernel_ms etc.: the largest)."""
    count = len(tickets)
    tt = (C.c_void_p * count)(*[t.value if isinstance(t, C.c_void_p) else t for t in tickets])
    tot = Stats()
    capi.check(lib(), lib().kr_trace_wait_many(count, tt, None, C.byref(tot)), "kr_trace_wait_many")
    return tot.as_dict()


def trace_release(ticket):
    capi.check(lib(), lib().kr_trace_release(ticket), "kr_trace_release")


def _pass(name, rays):
    """The f64 or f32 entry point of an O(N) pass, by the dtype of `rays`."""
    f32 = isinstance(rays, np.ndarray) and rays.dtype == capi.RAY_F32
    _rays_arg(rays, capi.RAY_F32 if f32 else capi.RAY_F64)
    return getattr(lib(), f"{name}_{'f32' if f32 else 'f64'}")


def redshift_start(spin, V, reverse, projradius, rays):
    capi.check(lib(), _pass("kr_redshift_start", rays)(spin, V, int(reverse), int(projradius), _ptr(rays), len(rays)), "kr_redshift_start")
    return rays


def redshift(spin, V, reverse, projradius, rays, motion=0):
    capi.check(lib(), _pass("kr_redshift", rays)(spin, V, int(reverse), int(projradius), motion, _ptr(rays), len(rays)), "kr_redshift")
    return rays


def redshift_dest(spin, reverse, rays):
    capi.check(lib(), _pass("kr_redshift_dest", rays)(spin, int(reverse), _ptr(rays), len(rays)), "kr_redshift_dest")
    return rays


def range_phi(rays, lo=-np.pi, hi=np.pi):
    capi.check(lib(), _pass("kr_range_phi", rays)(lo, hi, _ptr(rays), len(rays)), "kr_range_phi")
    return rays


def calculate_momentum(spin, rays):
    capi.check(lib(), _pass("kr_calculate_momentum", rays)(spin, _ptr(rays), len(rays)), "kr_calculate_momentum")
    return rays


def pointsource_count(spec):
    nc, nb = C.c_int32(), C.c_int32()
    n = lib().kr_pointsource_count(C.byref(spec), C.byref(nc), C.byref(nb))
    return n, nc.value, nb.value


def imageplane_count(spec):
    nx, ny = C.c_int32(), C.c_int32()
    n = lib().kr_imageplane_count(C.byref(spec), C.byref(nx), C.byref(ny))
    return n, nx.value, ny.value


def pointsource_init(spec):
    n, _, _ = pointsource_count(spec)
    rays = np.zeros(n, dtype=capi.RAY_F64)
    capi.check(lib(), lib().kr_pointsource_init_f64(C.byref(spec), _ptr(rays), n), "kr_pointsource_init")
    return rays


def imageplane_init(spec):
    n, _, _ = imageplane_count(spec)
    rays = np.zeros(n, dtype=capi.RAY_F64)
    capi.check(lib(), lib().kr_imageplane_init_f64(C.byref(spec), _ptr(rays), n), "kr_imageplane_init")
    return rays


def reduce_emissivity(bins, rays):
    _rays_arg(rays, capi.RAY_F64)
    nr = bins.nr
    count = np.zeros(nr, dtype=np.int64)
    flux, emis, sg, stt = (np.zeros(nr) for _ in range(4))
    dc = C.c_int64()
    capi.check(lib(), lib().kr_reduce_emissivity_f64(C.byref(bins), _ptr(rays), len(rays), _ptr(count), _ptr(flux), _ptr(emis),
                                                     _ptr(sg), _ptr(stt), C.byref(dc)), "kr_reduce_emissivity")
    return {"count": count, "flux": flux, "emis": emis, "sum_redshift": sg, "sum_time": stt, "disc_count": dc.value}


def reduce_image(bins, rays):
    _rays_arg(rays, capi.RAY_F64)
    npix = bins.img_nx * bins.img_ny
    nrays = np.zeros(npix, dtype=np.int32)
    planes = {k: np.zeros(npix) for k in ("flux", "r", "phi", "enshift", "time", "emis")}
    dc = C.c_int64()
    capi.check(lib(), lib().kr_reduce_image_f64(C.byref(bins), _ptr(rays), len(rays), _ptr(nrays), *[_ptr(planes[k]) for k in
                                                ("flux", "r", "phi", "enshift", "time", "emis")], C.byref(dc)), "kr_reduce_image")
    out = {"nrays": nrays, "disc_count": dc.value}
    out.update(planes)
    return out


def image_planes_from_words(words, nx, ny):
    """The device-resident reducers' result buffer (kr_reduce_image_dev_f64 / kr_post_image_dev_f64: 7 nx ny + 1 doubles,
    [nrays | flux | r | phi | enshift | time | emis | disc_count]) as the dict reduce_image() returns."""
    npix = nx * ny
    words = np.asarray(words)
    out = {"nrays": np.rint(words[:npix]).astype(np.int32), "disc_count": int(round(float(words[7 * npix])))}
    for q, k in enumerate(("flux", "r", "phi", "enshift", "time", "emis")):
        out[k] = words[(q + 1) * npix:(q + 2) * npix]
    return out
