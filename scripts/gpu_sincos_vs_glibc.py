#!/usr/bin/env python3
"""GPU box: how often does the strict path's sin / cos (kr_sincos_f64 on the device) carry glibc's bits, per call?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from raytrace_cpu_amd import api, capi
libm = C.CDLL("libm.so.6")
for f in (libm.sin, libm.cos):
    f.restype, f.argtypes = C.c_double, [C.c_double]
def probe(op, a):
    a = np.ascontiguousarray(a, dtype=np.float64); out = np.empty_like(a); lib = api.lib()
    capi.check(lib, lib.kr_debug_arith_f64(op, a.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(a)), "probe")
    return out
rng = np.random.default_rng(11)
for lo, hi in ((0.0, 0.0078), (0.0078, 0.1), (0.1, 0.8), (0.8, 1.5), (1.5, 1.64), (1.64, 3.2)):
    x = rng.uniform(lo, hi, 300000)
    gs = np.array([libm.sin(v) for v in x]); gc = np.array([libm.cos(v) for v in x])
    ds, dc = probe(4, x), probe(5, x)
    print(f"[{lo}, {hi}]: device sin == glibc {np.mean(ds == gs):.5f}  cos {np.mean(dc == gc):.5f} | numpy sin == glibc {np.mean(np.sin(x) == gs):.5f}  cos {np.mean(np.cos(x) == gc):.5f}", flush=True)
