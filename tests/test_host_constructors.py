"""CPU: the class mirror's PointSource<double> / ImagePlane<double> constructors (raytrace_cpu_amd/host/raytracer/{pointsource,imageplane}.cpp, the
shared camera ray of image_ray.h) against the oracle's constructors -- which are pinned bit for bit to the compiled reference
(tests/test_oracle_vs_ref.py): every field the reference defines, bit for bit, on several grids (src/raytracer/pointsource.cpp:11-64,
imageplane.cpp:11-121).  tests/cpp/host_ctor_dump builds the objects; nothing is traced, no GPU is needed."""
import math
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from raytrace_cpu_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_ctor_dump")
FIELDS = ("t", "r", "theta", "phi", "pt", "pr", "ptheta", "pphi", "k", "h", "Q", "steps", "status", "rdot_sign", "thetadot_sign", "rdot_flips", "alpha", "beta")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), EXE])
    return EXE


def _dump(exe, tmp_path, kind, args):
    out = tmp_path / f"{kind}.bin"
    env = dict(os.environ, KRTRACE_HOST_THREADS="4")
    r = subprocess.run([exe, kind, str(out)] + [repr(float(a)) for a in args], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    raw = out.read_bytes()
    n = int(np.frombuffer(raw[:4], dtype=np.int32)[0])
    return np.frombuffer(raw[4:], dtype=capi.RAY_F64, count=n)


@pytest.mark.parametrize("pos,V,spin,dc,db,c0,cmax,b0,bmax", [
    ([0.0, 10.0, 1e-3, 1.5707], 0.0, 0.998, 0.05, 0.05, -0.995, 0.995, -math.pi, math.pi),
    ([0.0, 5.0, 1e-3, 0.0], 1.0 / (0.998 + 5.0 ** 1.5), 0.998, 0.2, 0.2, -0.999999, 0.995, -0.995 * math.pi, math.pi),
    ([0.0, 6.0, math.pi / 2 - 1e-6, 1.5707], 1.0 / (0.5 + 6.0 ** 1.5), 0.5, 0.07, 0.03, -0.995, 0.995, 0.0, math.pi)])
def test_pointsource_constructor_carries_the_reference_bits(exe, tmp_path, pos, V, spin, dc, db, c0, cmax, b0, bmax):
    got = _dump(exe, tmp_path, "ps", pos + [V, spin, dc, db, c0, cmax, b0, bmax, 1.0])
    want = ol.oracle_pointsource(ol.pointsource_spec(pos, V, spin, dc, db, cosalpha0=c0, cosalphamax=cmax, beta0=b0, betamax=bmax))
    assert len(got) == len(want) > 300
    live = want["steps"] != -1
    assert np.array_equal(got["steps"] != -1, live)
    assert not ol.rays_equal_bitwise(got[live], want[live], fields=FIELDS)


@pytest.mark.parametrize("incl,spin,x0,xmax,dx,y0,ymax,dy,phi0", [(80.0, 0.998, -30.0, 30.0, 60.0 / 64, -30.0, 30.0, 60.0 / 64, 0.0),      # contains the (0, 0) pixel: NaN constants
                                                                  (30.0, 0.5, -30.0, 30.0, 60.0 / 63, -25.0, 30.0, 55.0 / 41, 0.3)])
def test_imageplane_constructor_carries_the_reference_bits(exe, tmp_path, incl, spin, x0, xmax, dx, y0, ymax, dy, phi0):
    got = _dump(exe, tmp_path, "ip", [10000.0, incl, x0, xmax, dx, y0, ymax, dy, spin, phi0])
    spec = ol.imageplane_spec(10000.0, incl, x0, xmax, dx, y0, ymax, dy, spin, phi0=phi0)
    want = ol.oracle_imageplane(spec)
    assert len(got) == len(want) > 300
    live = want["steps"] != -1
    assert np.array_equal(got["steps"] != -1, live)
    assert not ol.rays_equal_bitwise(got[live], want[live], fields=FIELDS)
