"""CPU: the C-ABI shared library loads, exports every symbol include/kr_trace.h declares, agrees with the
ctypes struct layouts, and -- without a GPU -- refuses to compute instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytrace_cpu_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        from raytrace_cpu_amd import _build
        _build.build()
    return capi.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "kr_trace.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kr_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"libkrtrace.so does not export {n}"
        assert n in capi.PROTOTYPES, f"capi.py has no prototype for {n}"
    assert set(capi.PROTOTYPES) == set(names)


def test_struct_layouts_match_header(lib):
    assert capi.RAY_F64.itemsize == 144 and capi.RAY_F32.itemsize == 84      # Ray<double>, Ray<float>
    assert capi.RAY_F64.fields["steps"][1] == 104 and capi.RAY_F64.fields["alpha"][1] == 128
    assert C.sizeof(capi.Params) == 128 and C.sizeof(capi.Stats) == 136
    assert C.sizeof(capi.PointSourceSpec) == 112 and C.sizeof(capi.ImagePlaneSpec) == 88
    assert C.sizeof(capi.EmisBins) == 56 and C.sizeof(capi.ImageBins) == 104 and C.sizeof(capi.ReturnBins) == 56
    p = capi.Params()
    lib.kr_params_default(C.byref(p), 0.998)
    q = capi.default_params(0.998)
    assert bytes(p) == bytes(q)
    assert p.horizon == 1.0632139225171164


def test_ray_grid_counts(lib):
    import oracle_lib as ol
    # 5167 = int-truncated product of doubles, larger than n_cosalpha * n_beta (pointsource.cpp:12)
    spec = ol.pointsource_spec([0, 5, 1e-3, 0], 0.0, 0.998, 0.05, 0.05, cosalpha0=-0.995, cosalphamax=0.995, beta0=-np.pi, betamax=np.pi)
    nc, nb = C.c_int32(), C.c_int32()
    n = lib.kr_pointsource_count(C.byref(spec), C.byref(nc), C.byref(nb))
    assert (n, nc.value, nb.value) == (5167, 40, 126)
    assert n == ol.oracle().kro_pointsource_count(C.byref(spec), None, None)
    ip = ol.imageplane_spec(10000.0, 80.0, -30, 30, 60 / 16, -30, 30, 60 / 16, 0.998)
    assert lib.kr_imageplane_count(C.byref(ip), None, None) == 289 == ol.oracle().kro_imageplane_count(C.byref(ip), None, None)


def test_no_cpu_fallback_without_gpu(lib):
    if lib.kr_device_count() > 0:
        pytest.skip("a GPU is visible")
    rays = np.zeros(8, dtype=capi.RAY_F64)
    p = capi.default_params(0.998)
    p.integrator = capi.RK4
    rc = lib.kr_trace_f64(C.byref(p), rays.ctypes.data_as(C.c_void_p), 8, None)
    assert rc == capi.KR_ENODEVICE
    assert b"no HIP device" in lib.kr_last_error()
    assert lib.kr_redshift_f64(0.998, -1.0, 0, 0, 0, rays.ctypes.data_as(C.c_void_p), 8) == capi.KR_ENODEVICE
    assert (rays["r"] == 0).all()


def test_host_mirror_fails_loudly_without_gpu():
    """The C++ mirror of the reference class API has no CPU integration loop: on a box without a GPU run_raytrace() ends in
    an exception that names the reason (it must never return silently with untraced rays)."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "tests", "cpp", "host_api_test")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no HIP device available" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.parametrize("app,par", [("kr_emissivity", "emissivity.par"), ("kr_imageplane_disc_image", "imageplane_rk4.par")])
def test_device_resident_programs_fail_loudly_without_gpu(app, par, tmp_path):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "raytrace_cpu_amd", "apps")], check=True)
    exe = os.path.join(ROOT, "raytrace_cpu_amd", "apps", "_build", app)
    out = tmp_path / "out"
    r = subprocess.run([exe, f"--parfile={os.path.join(ROOT, 'tests', 'golden', 'apps', par)}", f"--outfile={out}"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and not out.exists()
    assert "no HIP device available" in r.stderr or "no ROCm-capable device" in r.stderr


def test_bench_refuses_to_run_without_gpu():
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout) and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_header_is_valid_c99_and_links_from_c(tmp_path):
    """include/kr_trace.h from a C translation unit (gcc -std=c99 -pedantic-errors), linked against libkrtrace.so."""
    import subprocess
    exe = tmp_path / "abi_c_check"
    csrc = os.path.join(ROOT, "raytrace_cpu_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-pedantic-errors", "-Wall", "-Werror", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "abi_c_check.c"),
                    "-L" + csrc, "-lkrtrace", "-Wl,-rpath," + csrc], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sizeof(ray_f64) 144 sizeof(ray_f32) 84" in r.stdout and "rays 5167" in r.stdout and "horizon 1.0632139225171164" in r.stdout
