"""Definition of the golden cases (shared by tests/golden/make_golden.py, which captures them from the
compiled reference, and by the parity tests, which replay them through the oracle and the HIP path).

Grids follow the reference's own tests, coarsened so that the committed fixtures
stay small:
  ps_h5    src/tests/emissivity_rk45_test.cpp:37-51 / integrator_perf_test.cpp:35-45 source (0,5,1e-3,0), V=0
  ps_h10   BASELINE.json configs[0..1]: par_example/emissivity.par_example source + --source_h=10
  ps_kep   src/tests/raytrace_rk4_test.cpp:26-32 (Keplerian source, default angular limits); ps_kep5k: the same on 5000 rays
  ps_h5_a05, ps_h10_a0   lamp posts at a = 0.5 / a = 0 for the RayDestination overloads (raytracer.cpp:1036-1254, :1600-1894)
  ps_h10_a0_landing      two rows of a 1e6-ray grid of the a = 0 lamp post: the clipped last step onto the disc (raytracer.cpp:870-871, :242-243)
  ip15/16  par_example/imageplane_disc_image.par_example geometry on a 16x16 / 17x17 ray grid
           (the 17x17 grid contains the (0,0) pixel whose constants are NaN)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytrace_cpu_amd import capi  # noqa: E402
import oracle_lib as ol  # noqa: E402

SPIN = 0.998
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def _ps(pos, V, dc, db, full_range=True, spin=SPIN):
    if full_range:
        return ol.pointsource_spec(pos, V, spin, dc, db, cosalpha0=-0.995, cosalphamax=0.995, beta0=-np.pi, betamax=np.pi)
    return ol.pointsource_spec(pos, V, spin, dc, db)


def kep_velocity(r, a=SPIN):
    # disc_velocity(), reference src/include/kerr.h:35-38 (pow(r, 3./2.))
    return 1.0 / (a + r ** 1.5)


def r_isco(spin=SPIN):
    return ol.oracle().kro_kerr_isco(spin, 1)


def _params(integrator, stop_kind=capi.STOP_THETA, stop_params=(), spin=SPIN, r_max=1000.0, rk45_tol=1e-8, steplim=-1):
    p = capi.default_params(spin)
    p.integrator, p.stop_kind, p.r_max, p.rk45_tol, p.steplim = integrator, stop_kind, r_max, rk45_tol, steplim
    for i, x in enumerate(stop_params):
        p.stop_params[i] = x
    return p


# name -> dict(source=spec, start=(V, reverse, projradius) for redshift_start,
#              post=(V, reverse, projradius) for redshift, runs={run_name: Params})
def cases():
    risco = r_isco()
    half_pi = np.pi / 2
    c = {}
    c["ps_h5"] = dict(
        source=_ps([0.0, 5.0, 1e-3, 0.0], 0.0, 0.1, 0.1), start=(0.0, 0, 0), post=(-1.0, 0, 0),
        runs={
            "euler": _params(capi.EULER),
            "rk4": _params(capi.RK4),
            "rk45": _params(capi.RK45),
            "rk45_tol1e-6": _params(capi.RK45, rk45_tol=1e-6),
            "rk4_flatdisc": _params(capi.RK4, capi.STOP_FLATDISC, (half_pi,)),
            "rk4_isco": _params(capi.RK4, capi.STOP_DISC_ISCO, (risco, 400.0, half_pi)),
            "rk45_flatdisc": _params(capi.RK45, capi.STOP_FLATDISC, (half_pi,)),
            "rk45_isco": _params(capi.RK45, capi.STOP_DISC_ISCO, (risco, -1.0, half_pi)),
            "rk4_steplim300": _params(capi.RK4, steplim=300),
        })
    c["ps_h10"] = dict(
        source=_ps([0.0, 10.0, 1e-3, 1.5707], 0.0, 0.1, 0.1), start=(0.0, 0, 0), post=(-1.0, 0, 0),
        runs={"euler": _params(capi.EULER), "rk4": _params(capi.RK4), "rk45": _params(capi.RK45),
              # BASELINE configs[2] at h = 10: both ends of the tolerance sweep of src/tests/emissivity_rk45_tol_sweep.py:38
              "rk45_tol1e-6": _params(capi.RK45, rk45_tol=1e-6), "rk45_tol1e-10": _params(capi.RK45, rk45_tol=1e-10)})
    c["ps_kep"] = dict(
        source=_ps([0.0, 5.0, 1e-3, 0.0], kep_velocity(5.0), 0.2, 0.2, full_range=False),
        start=(kep_velocity(5.0), 0, 0), post=(-1.0, 0, 0),
        runs={"euler": _params(capi.EULER), "rk4": _params(capi.RK4)})
    # the same Keplerian source on a 40 x 125 grid (5000 rays): on the 320-ray grid above the reference's own 1-ulp noise envelope is
    # 3-4 % of the rays -- a coarse-grid artefact a test cannot tell from a regression
    c["ps_kep5k"] = dict(
        source=_ps([0.0, 5.0, 1e-3, 0.0], kep_velocity(5.0), 0.05, 0.05, full_range=False),
        start=(kep_velocity(5.0), 0, 0), post=(-1.0, 0, 0),
        runs={"euler": _params(capi.EULER), "rk4": _params(capi.RK4)})
    # run_raytrace(RayDestination*) AWAY from a = 0.998 (VERDICT r02 missing #2): at a <= 0.5 the ISCO lies well outside the photon
    # sphere, rays cross the equatorial plane inside it, whirl and come back -- the reference's result for those rays is decided at
    # the 1-ulp level (tests/tool_oracle_isco_noise.py), so only an arithmetic that carries its bits reproduces their integer outputs
    for name, spin, pos in (("ps_h5_a05", 0.5, [0.0, 5.0, 1e-3, 0.0]), ("ps_h10_a0", 0.0, [0.0, 10.0, 1e-3, 0.0])):
        ri = r_isco(spin)
        c[name] = dict(
            source=_ps(pos, 0.0, 0.05, 0.1, spin=spin), start=(0.0, 0, 0), post=(-1.0, 0, 0),
            runs={
                "rk4_isco": _params(capi.RK4, capi.STOP_DISC_ISCO, (ri, 400.0, half_pi), spin=spin),
                "rk4_flatdisc": _params(capi.RK4, capi.STOP_FLATDISC, (half_pi,), spin=spin),
                "rk45_isco": _params(capi.RK45, capi.STOP_DISC_ISCO, (ri, 400.0, half_pi), spin=spin),
                "rk45_flatdisc": _params(capi.RK45, capi.STOP_FLATDISC, (half_pi,), spin=spin),
            })
    # Landing on the disc: two rows (cos alpha ~ -0.64) of the 1000 x 1000 grid of tests/tool_gpu_hybrid_sweep.py's "lamp h=10 a=0", theta-limit
    # overload.  The clipped last step |(theta_lim - theta) / thetadot| decides whether theta lands ON the limit or an ulp short of it (one more
    # step): with an approximate reciprocal in that quotient 12 of these 2 000 Euler rays took the extra step (profiles/r03_hybrid_sweep_euler.jsonl,
    # first version) -- held to ZERO step-count differences in every arithmetic mode (test_trace_vs_golden).
    d_landing = 1.99 / 999.0
    c["ps_h10_a0_landing"] = dict(
        source=ol.pointsource_spec([0.0, 10.0, 1e-3, 0.0], 0.0, 0.0, d_landing, d_landing * np.pi / 0.995, cosalpha0=-0.995 + 178 * d_landing,
                                   cosalphamax=-0.995 + 180 * d_landing + 1e-9, beta0=-np.pi, betamax=np.pi),
        start=(0.0, 0, 0), post=(-1.0, 0, 0),
        runs={"euler": _params(capi.EULER, spin=0.0), "rk4": _params(capi.RK4, spin=0.0)})
    ip = dict(dist=10000.0, inc_deg=80.0, x0=-30.0, xmax=30.0, y0=-30.0, ymax=30.0, spin=SPIN)
    incl = 80.0 * np.pi / 180
    c["ip15"] = dict(
        source=ol.imageplane_spec(dx=60.0 / 15, dy=60.0 / 15, **ip), start=(0.0, 1, 0), post=(-1.0, 1, 0),
        runs={
            "euler": _params(capi.EULER, spin=-SPIN, r_max=11000.0),
            "rk4": _params(capi.RK4, spin=-SPIN, r_max=11000.0),
            "rk45": _params(capi.RK45, spin=-SPIN, r_max=11000.0),
            "rk4_isco": _params(capi.RK4, capi.STOP_DISC_ISCO, (risco, 30.0, half_pi), spin=-SPIN, r_max=11000.0),
            "rk4_plane": _params(capi.RK4, capi.STOP_FLATPLANE, (incl, 0.0, 20.0), spin=-SPIN, r_max=11000.0),
            "rk45_plane": _params(capi.RK45, capi.STOP_FLATPLANE, (incl, 0.0, 20.0), spin=-SPIN, r_max=11000.0),
        })
    c["ip16"] = dict(
        source=ol.imageplane_spec(dx=60.0 / 16, dy=60.0 / 16, **ip), start=(0.0, 1, 0), post=(-1.0, 1, 0),
        runs={"rk4": _params(capi.RK4, spin=-SPIN, r_max=11000.0),
              "euler": _params(capi.EULER, spin=-SPIN, r_max=11000.0)})
    return c


def emis_bins(spec, nr=30, r_disc=500.0, gamma=2.0):
    """Bin geometry of src/tests/emissivity_rk45_test.cpp:163-164 / src/emissivity/emissivity.cpp:57-61."""
    b = capi.EmisBins()
    b.r_isco = r_isco()
    b.r_min = b.r_isco
    b.dr = float(np.exp(np.log(r_disc / b.r_min) / nr))
    b.gamma, b.spin = gamma, spec.spin
    b.num_primary_rays = float(int(((spec.cosalphamax - spec.cosalpha0) / spec.dcosalpha) * ((spec.betamax - spec.beta0) / spec.dbeta)))
    b.nr, b.logbin = nr, 1
    return b


def image_bins(spec, img_n=8, r_disc=30.0):
    """imageplane_disc_image.cpp:81-84,122-140 with par_example defaults (q=3, rb1=4, rb2=10, flip_image)."""
    b = capi.ImageBins()
    b.x0, b.y0 = spec.x0, spec.y0
    b.img_dx, b.img_dy = (spec.xmax - spec.x0) / img_n, (spec.ymax - spec.y0) / img_n
    b.r_isco, b.r_disc = r_isco(), r_disc
    b.q1, b.rb1, b.q2, b.rb2, b.q3 = 3.0, 4.0, 3.0, 10.0, 3.0
    b.img_nx, b.img_ny, b.flip_image, b.pad = img_n, img_n, 1, 0
    return b


def is_imageplane(case):
    return isinstance(case["source"], capi.ImagePlaneSpec)


def golden_path(case_name):
    return os.path.join(GOLDEN_DIR, f"{case_name}.npz")
