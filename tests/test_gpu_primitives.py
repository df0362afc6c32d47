"""GPU: the arithmetic primitives of the trace kernel, one at a time, against the host's IEEE results.
The strict path's claim is: + - * are IEEE, division and square root are CORRECTLY ROUNDED (bit-equal to numpy),
sin/cos are correctly rounded in practice, i.e. the device differs from the CPU only where glibc itself is not (<= 0.3 % of arguments)."""
import ctypes as C

import numpy as np
import pytest

from raytrace_cpu_amd import api, capi

pytestmark = pytest.mark.gpu
N = 2_000_000


def probe(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float64)
    out = np.empty_like(a)
    lib = api.lib()
    capi.check(lib, lib.kr_debug_arith_f64(op, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(a)), "probe")
    return out


def ulps(x, ref):
    return np.abs(x - ref) / np.spacing(np.abs(ref))


def operands(rng, n):
    """magnitudes the tracer meets (1e-30 .. 1e12), both signs, plus exact powers of two and near-equal pairs"""
    a = rng.standard_normal(n) * 10.0 ** rng.uniform(-30, 12, n)
    b = rng.standard_normal(n) * 10.0 ** rng.uniform(-30, 12, n)
    a[:1000] = 2.0 ** rng.integers(-60, 60, 1000)
    b[1000:2000] = a[1000:2000] * (1 + rng.integers(-3, 4, 1000) * 2.0 ** -52)
    return a, b


def test_division_is_correctly_rounded_and_lean_chain_is_bit_identical(krlib):
    rng = np.random.default_rng(1)
    a, b = operands(rng, N)
    want = a / b
    assert np.array_equal(probe(0, a, b), want)          # compiler's sequence == IEEE
    assert np.array_equal(probe(1, a, b), want)          # lean chain == IEEE
    # the compiler's sequence is IEEE everywhere; the lean chain is only claimed (and only used) for finite non-zero
    # denominators away from the ends of the exponent range -- a zero numerator is fine, a zero / infinite denominator
    # gives NaN (the evaluation-level guard in kr_device.hpp re-runs those evaluations with the compiler's sequence)
    sa = np.array([0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, 1e-310, 1e300, 3.0, 1e-320, 1.0, 5e-324, 1e308], dtype=np.float64)
    sb = np.array([1.0, 3.0, 0.0, 2.0, np.inf, 1.0, 1e-310, 1e-300, np.inf, 3.0, 1e-310, 5e-324, 1e-308], dtype=np.float64)
    with np.errstate(all="ignore"):
        w = sa / sb
    assert np.array_equal(probe(0, sa, sb), w, equal_nan=True)
    g = probe(1, sa[:2], sb[:2])
    assert np.array_equal(g, w[:2])                       # 0/b = 0 (a -0 numerator comes back as +0)
    assert np.isnan(probe(1, sa[2:3], sb[2:3])).all()    # b = 0 -> NaN where IEEE says inf


def test_sqrt_is_correctly_rounded_and_lean_chain_is_bit_identical(krlib):
    rng = np.random.default_rng(2)
    a = np.abs(operands(rng, N)[0])
    want = np.sqrt(a)
    assert np.array_equal(probe(2, a), want)
    assert np.array_equal(probe(3, a), want)
    s = np.array([0.0, -0.0, np.inf, np.nan, -1.0, 1e-310, 5e-324, 1e-300, 1e300, 4.0], dtype=np.float64)
    with np.errstate(all="ignore"):
        w = np.sqrt(s)
    g = probe(2, s)
    assert np.array_equal(g, w, equal_nan=True) and np.array_equal(np.signbit(g), np.signbit(w)), (g, w)
    g = probe(3, s[[0, 9]])                     # lean chain: zero and ordinary values (its callers pass |x| of mid-range numbers)
    assert np.array_equal(g, w[[0, 9]])


def test_compact_sincos_within_one_ulp_of_glibc(krlib):
    rng = np.random.default_rng(3)
    x = rng.uniform(-1.0, 4.2, N)
    x[:1000] = np.pi / 2 + rng.uniform(-1e-6, 1e-6, 1000)
    x[1000:2000] = rng.uniform(0, 2e-3, 1000)
    for op, f in ((4, np.sin), (5, np.cos)):
        u = ulps(probe(op, x), f(x))
        assert u.max() <= 1.0, (op, u.max())
    # correctly rounded on the device: against glibc itself (what the reference calls; numpy may use a vector library) it differs only where
    # glibc is not (measured 99.8-99.99 % bit-equal)
    libm = C.CDLL("libm.so.6")
    xs = x[:300_000]
    for op, name in ((4, "sin"), (5, "cos")):
        f = getattr(libm, name)
        f.restype, f.argtypes = C.c_double, [C.c_double]
        want = np.array([f(v) for v in xs])
        assert (probe(op, xs) == want).mean() > 0.997, name
    # far outside the polar-angle range and for non-finite input the library path answers
    big = np.array([1e5, -3e7, 1e300, np.inf, np.nan], dtype=np.float64)
    with np.errstate(all="ignore"):
        assert ulps(probe(4, big[:3]), np.sin(big[:3])).max() <= 2
        assert np.isnan(probe(4, big[3:])).all() and np.isnan(probe(5, big[3:])).all()


def test_fast_math_primitives_within_two_ulp(krlib):
    rng = np.random.default_rng(4)
    a, b = operands(rng, N)
    assert ulps(probe(6, a, b), a / b).max() <= 2.0
    a = np.abs(a)
    assert ulps(probe(7, a), np.sqrt(a)).max() <= 1.0
    # no special cases: the argument is floored at 1e-300 (kr_device.hpp::fast_sqrt), so a vanishing velocity component
    # becomes 1e-150 -- below anything an O(1) coordinate can register -- instead of 0
    assert probe(7, np.array([0.0]))[0] == pytest.approx(1e-150, rel=1e-12)
    assert probe(7, np.array([-4.0]))[0] == 2.0          # callers pass |x|; the routine takes it again


def test_device_libm_distance_from_glibc(krlib):
    """documents (does not bound tightly) how far the device libm is from glibc: this is the ONLY source of
    strict-path differences where the library functions are still used (sources, redshift passes, RK45 pow)."""
    rng = np.random.default_rng(5)
    x = rng.uniform(-1.0, 4.2, 500_000)
    assert ulps(probe(8, x), np.sin(x)).max() <= 2 and ulps(probe(9, x), np.cos(x)).max() <= 2
    base = rng.uniform(1e-3, 1e10, 500_000)
    assert ulps(probe(10, base, np.full_like(base, 0.2)), base ** 0.2).max() <= 2


def test_division_by_a_uniform_constant_is_correctly_rounded(krlib):
    """kr_device.hpp::div_by_uniform (step / precision, / theta_precision, / 6 on the strict path): 3 instructions with a host-side
    correctly rounded reciprocal, bit-identical to IEEE division for every numerator."""
    rng = np.random.default_rng(11)
    a, _ = operands(rng, 1_000_000)
    a = np.concatenate([a, np.abs(a) * 1e-3, rng.uniform(0, 10, 200_000), [0.0, 1.0, 100.0, 1e-3, 5e-324, 1e300]])
    for b in (100.0, 50.0, 6.0, 3.0, 7.0, 10.0, 1e-5, 123456.789, 0.1, 1.9999999999999996 / 3):
        got = probe(17, a, np.full_like(a, b))
        want = a / b
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), b


def test_assembled_reciprocals_give_the_ieee_quotients(krlib):
    """kr_device.hpp::StageRecips: 1 / (sin^2 rho^2 Delta) and 1 / rho^4 are put together from the reciprocals of sin, rho^2 Delta and rho^2 and polished
    with one Newton step instead of being taken from the hardware; the quotients formed with them must be the IEEE ones, bit for bit, over the ranges
    a ray visits (sin theta down to 1e-12: polar-axis rays; rho^2 Delta from 1e-17: next to the horizon)."""
    rng = np.random.default_rng(20)
    n = 2_000_000
    sin = 10.0 ** rng.uniform(-12, 0, n) * rng.choice([-1.0, 1.0], n)
    b1 = 10.0 ** rng.uniform(-17, 7, n) * rng.choice([-1.0, 1.0], n)
    got = probe(20, sin, b1)
    want = 1.2345678901234567 / ((sin * sin) * b1)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    rhosq = 10.0 ** rng.uniform(0, 6.1, n)
    num = rng.normal(size=n) * 10.0 ** rng.uniform(-20, 20, n)
    got = probe(21, rhosq, num)
    want = num / (rhosq * rhosq)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_fifth_root_of_the_step_controller(krlib):
    """kr_device.hpp::fifth_root_for_controller: within 1.5 ulp of x**0.2 where the DOPRI5 controller can tell (its factor is clamped to
    [0.1, 5], i.e. x in [1.7e-5, 5.3e3]); outside, whatever saturates the clamp the same way; NaN propagates."""
    rng = np.random.default_rng(13)
    x = np.concatenate([10.0 ** rng.uniform(-6, 6, 1_000_000), rng.uniform(0.5, 2.0, 500_000), [1.0, 32.0, 1e-5, 1e10, 1e-300, 1e300]])
    got = probe(18, x)
    inside = (x >= 1e-6) & (x <= 1e6)
    assert ulps(got[inside], x[inside] ** 0.2).max() <= 1.0      # numpy's pow with the same double exponent 0.2 (not the exact 5th root)
    fac = lambda y: np.clip(0.9 * y, 0.1, 5.0)
    assert np.array_equal(fac(got[~inside]), fac(x[~inside] ** 0.2))          # 0.1 or 5 either way
    assert np.isnan(probe(18, np.array([np.nan]))[0])
    assert probe(18, np.array([1.0]))[0] == 1.0 and abs(probe(18, np.array([32.0]))[0] - 32.0 ** 0.2) <= 4.5e-16
    # correctly rounded in practice: against glibc's pow itself (what the reference calls; numpy may use a vector library), bit for bit on >= 99.8 %
    libm = C.CDLL("libm.so.6")
    libm.pow.restype, libm.pow.argtypes = C.c_double, [C.c_double, C.c_double]
    xs = 10.0 ** rng.uniform(-5, 4, 200_000)
    want = np.array([libm.pow(v, 0.2) for v in xs])
    got = probe(18, xs)
    assert ulps(got, want).max() <= 1.0 and (got == want).mean() >= 0.998, (got == want).mean()


def test_replayed_additions_equal_the_loop(krlib):
    """kr_replay_additions on the device (the RK45 fixed-point replay's t and phi): 20 000 additions in closed form against numpy
    doing them one at a time -- increments from far below an ulp (absorbed) to a few per cent of x, both directions, ties."""
    rng = np.random.default_rng(19)
    n = 8192
    x = rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 6, n)
    dx = x * 2.0 ** -rng.uniform(5, 70, n) * rng.choice([-1.0, 1.0], n)
    u = np.spacing(np.abs(x[:512]))
    dx[:512] = (rng.integers(0, 32, 512) + 0.5) * u * rng.choice([-1.0, 1.0], 512)        # exact ties
    dx[512:600] = 0.0
    got = probe(19, x, dx)
    want = x.copy()
    for _ in range(20000):
        want = want + dx
    assert (got.view(np.int64) == want.view(np.int64)).all()
