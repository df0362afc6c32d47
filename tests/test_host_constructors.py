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


@pytest.mark.parametrize("pos,V,spin,dc,db,c0,cmax,b0,bmax", [
    ([0.0, 10.0, 1e-3, 1.5707], 0.0, 0.998, 0.05, 0.05, -0.995, 0.995, -math.pi, math.pi),
    ([0.0, 6.0, math.pi / 2 - 1e-6, 1.5707], 1.0 / (0.5 + 6.0 ** 1.5), 0.5, 0.07, 0.03, -0.995, 0.995, 0.0, math.pi),
    # BASELINE configs[1] at full size (3162^2 rays): six of its 6324 table entries are arguments on which glibc's sincos() and sin() / cos() disagree
    ([0.0, 10.0, 1e-3, 1.5707], 0.0, 0.998, 1.99 / (math.sqrt(1e7) - 1.0), (1.99 / (math.sqrt(1e7) - 1.0)) * math.pi / 0.995, -0.995, 0.995, -math.pi, math.pi)])
def test_pointsource_tables_plus_ieee_arithmetic_give_the_reference_bits(pos, V, spin, dc, db, c0, cmax, b0, bmax):
    """kr_pointsource_tables (no GPU needed) holds the host-libm values the DEVICE PointSource constructor reads (kr_post_device.hpp::SourceTables).
    Here the rest of calculate_constants (raytracer.cpp:625-676) is redone in numpy float64 -- IEEE + - x / sqrt in the reference's association, which
    is all the device kernel adds -- and k, h, Q, the two signs must come out as the oracle's constructor (== the compiled reference's) has them."""
    import ctypes as C
    lib = capi.load()
    spec = ol.pointsource_spec(pos, V, spin, dc, db, cosalpha0=c0, cosalphamax=cmax, beta0=b0, betamax=bmax)
    n_ca, n_b = C.c_int32(), C.c_int32()
    n = lib.kr_pointsource_count(C.byref(spec), C.byref(n_ca), C.byref(n_b))
    a_sc, b_sc, p3 = np.zeros(2 * n_ca.value), np.zeros(2 * n_b.value), np.zeros(3)
    assert lib.kr_pointsource_tables(C.byref(spec), a_sc.ctypes.data_as(C.c_void_p), b_sc.ctypes.data_as(C.c_void_p), p3.ctypes.data_as(C.c_void_p)) == 0
    # the tables are the C library's sincos() (what an optimising build of the reference calls for sin(x), cos(x) of one x) at the reference's own grid expressions
    libm = C.CDLL("libm.so.6")
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def sincos(x):
        sn, cs = C.c_double(), C.c_double()
        libm.sincos(x, C.byref(sn), C.byref(cs))
        return sn.value, cs.value
    for i in (0, 1, n_ca.value // 2, n_ca.value - 1):
        assert (a_sc[2 * i], a_sc[2 * i + 1]) == sincos(math.acos(c0 + i * dc))
    for j in (0, 1, n_b.value // 3, n_b.value - 1):
        assert (b_sc[2 * j], b_sc[2 * j + 1]) == sincos(b0 + j * db)
    assert tuple(p3) == sincos(pos[2]) + (math.tan(pos[2]),)
    want = ol.oracle_pointsource(spec)
    assert len(want) == n
    if ol.ref() is not None and n > 1000000:          # (the oracle's constructor is pinned to the reference's on small grids; at full size, here)
        src = ol.RefSource(spec)
        for name in ("k", "h", "Q"):
            assert np.array_equal(src.rays[name].view(np.int64), want[name].view(np.int64)), name
        src.close()
    ix = np.arange(n_ca.value * n_b.value)
    i, j = ix // n_b.value, ix % n_b.value
    sa, ca, sb, cb = a_sc[2 * i], a_sc[2 * i + 1], b_sc[2 * j], b_sc[2 * j + 1]
    f = np.float64
    r, st, ct, tt, E, a = f(pos[1]), f(p3[0]), f(p3[1]), f(p3[2]), f(1.0), f(spin)
    with np.errstate(all="ignore"):
        rhosq = r * r + (a * ct) * (a * ct)
        delta = r * r - 2 * r + a * a
        sigmasq = (r * r + a * a) * (r * r + a * a) - a * a * delta * st * st
        e2nu = rhosq * delta / sigmasq
        e2psi = sigmasq * st * st / rhosq
        omega = 2 * a * r / sigmasq
        Vv = f(V)
        et0 = (1 / np.sqrt(e2nu)) / np.sqrt(1 - (Vv - omega) * (Vv - omega) * e2psi / e2nu)
        et3 = (1 / np.sqrt(e2nu)) * Vv / np.sqrt(1 - (Vv - omega) * (Vv - omega) * e2psi / e2nu)
        e10 = (Vv - omega) * np.sqrt(e2psi / e2nu) / np.sqrt(e2nu - (Vv - omega) * (Vv - omega) * e2psi)
        e13 = (1 / np.sqrt(e2nu * e2psi)) * (e2nu + Vv * omega * e2psi - omega * omega * e2psi) / np.sqrt(e2nu - (Vv - omega) * (Vv - omega) * e2psi)
        e22 = -1 / np.sqrt(rhosq)
        e31 = np.sqrt(delta / rhosq)
        rp0, rp1, rp2, rp3 = E, E * sa * cb, E * sa * sb, E * ca
        tdot = rp0 * et0 + rp1 * e10
        phidot = rp0 * et3 + rp1 * e13
        rdot = rp3 * e31
        thetadot = rp2 * e22
        k = (1 - 2 * r / rhosq) * tdot + (2 * a * r * st * st / rhosq) * phidot
        h = phidot * ((r * r + a * a) * (r * r + a * a * ct * ct - 2 * r) * st * st + 2 * a * a * r * st * st * st * st)
        h = h - 2 * a * r * k * st * st
        h = h / (r * r + a * a * ct * ct - 2 * r)
        Q = rhosq * rhosq * thetadot * thetadot - (a * k * ct + h / tt) * (a * k * ct - h / tt)
    live = want["steps"][:len(ix)] != -1
    assert live.sum() > 300
    for name, got in (("k", k), ("h", h), ("Q", Q)):
        assert np.array_equal(np.asarray(got)[live].view(np.int64), want[name][:len(ix)][live].view(np.int64)), name
    assert np.array_equal(np.where(rdot >= 0, 1, -1)[live], want["rdot_sign"][:len(ix)][live])
    assert np.array_equal(np.where(thetadot > 0, 1, -1)[live], want["thetadot_sign"][:len(ix)][live])
