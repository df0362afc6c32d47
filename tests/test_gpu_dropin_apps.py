"""GPU: the reference's OWN application binaries, compiled unchanged against this repo's host-side API mirror and
libkrtrace.so (dropin/build_apps.sh -> dropin/_build/, built in the build container and shipped as binaries),
produce the same output files as their CPU builds (fixtures in tests/golden/apps/).  Skipped where the binaries
were not built (they need the reference sources at build time)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import fits_lite
import golden_cases as gc
from test_oracle_app_outputs import APPS, load_dat

pytestmark = pytest.mark.gpu

BUILD = os.path.join(gc.ROOT, "dropin", "_build")
ENV = dict(os.environ, LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libstdc++.so.6", LD_LIBRARY_PATH="/opt/conda/lib")


COUNT_KEYS = {"DISC_N", "HIT_N", "ESC_N", "NHIT", "DISCRAYS"}     # header cards whose value may move by a chaotic ray


def close_to_cpu(test, case, g, w, rtol=1e-6, atol=0.0):
    """np.testing.assert_allclose at the north-star tolerance, with the measured worst relative difference left in the round's margin record
    (gpurun_out/parity_margins.json -> profiles/): how close each program's output runs to its bar is on file, not just pass / fail."""
    import parity
    g, w = np.asarray(g, dtype=np.float64), np.asarray(w, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.where(g == w, 0.0, np.abs(g - w) / np.maximum(np.abs(w), 1e-300))
    rel = np.where(np.abs(g - w) <= atol, 0.0, rel)
    worst = float(rel.max()) if rel.size else 0.0
    parity.record_margin(test, case, {"n_traced": int(rel.size), "n_bad": int((rel > rtol).sum()), "frac_bad": float((rel > rtol).mean()) if rel.size else 0.0, "worst_ok": worst}, rtol)
    assert worst <= rtol, (test, case, worst, rtol)


def need(app):
    path = os.path.join(BUILD, app)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (dropin/build_apps.sh needs the reference sources)")
    return path


def run_emissivity_app(app, arithmetic=None):
    exe = need(app)
    env = dict(ENV, KRTRACE_ARITHMETIC=arithmetic) if arithmetic else ENV
    with tempfile.TemporaryDirectory() as w:
        os.makedirs(os.path.join(w, "par"))
        os.makedirs(os.path.join(w, "run"))
        shutil.copy(os.path.join(APPS, f"{app}.par"), os.path.join(w, "par", f"{app}.par"))   # the app's built-in default path
        out = os.path.join(w, "out.dat")
        subprocess.run([exe, f"--outfile={out}"], cwd=os.path.join(w, "run"), check=True, stdout=subprocess.DEVNULL, env=env, timeout=300)
        rows = [l.split() for l in open(out) if l.strip()]
    return np.array([[float(x) for x in r] for r in rows])


@pytest.mark.parametrize("arithmetic", [None, "strict"])          # None: the host mirror's default (hybrid)
@pytest.mark.parametrize("app", ["emissivity", "emissivity_rd"])
def test_emissivity_apps_match_cpu_output(app, arithmetic):
    got, want = run_emissivity_app(app, arithmetic), load_dat(f"{app}.dat")
    assert got.shape == want.shape
    assert (got[:, :2] == want[:, :2]).all()                       # bin radii and areas: host-only code, identical
    dcount = np.abs(got[:, 2] - want[:, 2])
    assert dcount.max() <= 1, dcount.max()                          # a chaotic ray may move between neighbouring bins
    same = dcount == 0
    assert (~same).sum() <= 2, (~same).sum()                        # ... and that is at most one ray (two bins) per table: the sums of every other bin are checked
    for col in (3, 4, 5, 6):
        g, w = got[same, col], want[same, col]
        assert (np.isnan(g) == np.isnan(w)).all()
        ok = ~np.isnan(w)
        np.testing.assert_allclose(g[ok], w[ok], rtol=1e-6, err_msg=f"{app} column {col}")   # BASELINE north star: 1e-6 per bin


@pytest.mark.parametrize("par", ["imageplane_rk4", "imageplane_rk45"])
def test_imageplane_app_matches_cpu_output(par):
    exe = need("imageplane_disc_image")
    with tempfile.TemporaryDirectory() as w:
        out = os.path.join(w, "out.fits")
        subprocess.run([exe, f"--parfile={os.path.join(APPS, par + '.par')}", f"--outfile={out}"], check=True, stdout=subprocess.DEVNULL, env=ENV, timeout=300)
        got = {h["name"]: h for h in fits_lite.read(out)}
        got_cards = fits_lite.header_cards(out)
    want = {h["name"]: h for h in fits_lite.read(os.path.join(APPS, par + ".fits"))}
    assert list(got) == list(want) == ["PRIMARY", "FLUX", "RADIUS", "PHI", "ENSHIFT", "TIME", "EMIS"]
    # the drop-in build writes the file with this repo's fits_output.h (no cfitsio): every header card identical to cfitsio's
    assert got_cards == fits_lite.header_cards(os.path.join(APPS, par + ".fits"))
    for k in ("DIST", "INCL", "SPIN", "ISCO", "RDISC", "NRAYS", "DISCRAYS"):
        assert got["PRIMARY"]["header"][k] == want["PRIMARY"]["header"][k], k
    for name in list(want)[1:]:           # RK4 and RK45 alike: the north-star 1e-6 on every pixel of every plane
        g, w = got[name]["data"], want[name]["data"]
        assert (np.isnan(g) == np.isnan(w)).all(), name             # NaN pattern (empty pixels) identical
        ok = ~np.isnan(w)
        close_to_cpu("test_imageplane_app_matches_cpu_output", f"{par}-{name}", g[ok], w[ok], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("par,app", [("imageplane_isco", "imageplane_disc_image_isco"), ("imageplane_rd", "imageplane_disc_image_rd")])
def test_imageplane_destination_apps_match_cpu_output(par, app):
    """src/imageplane/imageplane_disc_image_isco.cpp (DiscWithISCODestination) and imageplane_disc_image_rd.cpp
    (FlatDiscDestination + redshift through the destination's four-velocity), unmodified, on the HIP path."""
    exe = need(app)
    with tempfile.TemporaryDirectory() as w:
        out = os.path.join(w, "out.fits")
        if app.endswith("_rd"):     # same dangling --parfile pointer as the emissivity programs: use the built-in default path
            os.makedirs(os.path.join(w, "par"))
            os.makedirs(os.path.join(w, "run"))
            shutil.copy(os.path.join(APPS, par + ".par"), os.path.join(w, "par", app + ".par"))
            subprocess.run([exe, f"--outfile={out}"], cwd=os.path.join(w, "run"), check=True, stdout=subprocess.DEVNULL, env=ENV, timeout=300)
        else:
            subprocess.run([exe, f"--parfile={os.path.join(APPS, par + '.par')}", f"--outfile={out}"], check=True, stdout=subprocess.DEVNULL, env=ENV, timeout=300)
        got = {h["name"]: h for h in fits_lite.read(out)}
        got_cards = fits_lite.header_cards(out)
    golden = os.path.join(APPS, par + ".fits")
    want = {h["name"]: h for h in fits_lite.read(golden)}
    assert list(got) == list(want) and got_cards == fits_lite.header_cards(golden)
    for name in list(want)[1:]:
        g, w_ = got[name]["data"], want[name]["data"]
        assert (np.isnan(g) == np.isnan(w_)).all(), name
        ok = ~np.isnan(w_)
        np.testing.assert_allclose(g[ok], w_[ok], rtol=1e-6, atol=1e-12, err_msg=name)


@pytest.mark.parametrize("tol,fixture", [("1e-8", "emissivity_rk45_plot.csv"), ("1e-6", "emissivity_rk45_plot_tol1e-6.csv"), ("1e-10", "emissivity_rk45_plot_tol1e-10.csv")])
def test_rk45_tolerance_sweep_program_matches_cpu_output(tol, fixture):
    """BASELINE configs[2]: src/tests/emissivity_rk45_plot.cpp (one point of the tolerance sweep: RK4 and RK45 emissivity
    profiles of the same 125 863-slot grid, 22 s on the build container's CPUs), unmodified, on the HIP path -- at the sweep's
    reference tolerance and at both of its ends (src/tests/emissivity_rk45_tol_sweep.py:38)."""
    exe = need("emissivity_rk45_plot")
    with tempfile.TemporaryDirectory() as w:
        out = os.path.join(w, "out.csv")
        subprocess.run([exe, out, tol], check=True, stdout=subprocess.DEVNULL, env=ENV, timeout=600)
        got_lines = open(out).read().splitlines()
    want_lines = open(os.path.join(APPS, fixture)).read().splitlines()
    assert got_lines[:2] == want_lines[:2] and len(got_lines) == len(want_lines)
    got = np.array([[float(x) for x in l.split()] for l in got_lines[2:]])
    want = np.array([[float(x) for x in l.split()] for l in want_lines[2:]])
    assert (got[:, 0] == want[:, 0]).all()
    assert np.abs(got[:, 1] - want[:, 1]).max() <= 1 and np.abs(got[:, 2] - want[:, 2]).max() <= 2      # ray counts: RK4, RK45
    same4, same45 = got[:, 1] == want[:, 1], got[:, 2] == want[:, 2]
    assert (~same4).sum() <= 1 and (~same45).sum() <= 2, ((~same4).sum(), (~same45).sum())      # bins left out of the sum check below (a ray moved across an edge)
    for col, same in ((3, same4), (5, same4), (7, same4), (4, same45), (6, same45), (8, same45)):
        close_to_cpu("test_rk45_tolerance_sweep_program_matches_cpu_output", f"tol{tol}-column{col}", got[same, col], want[same, col], rtol=1e-6)


def test_integrator_perf_report_matches_cpu_statistics():
    """src/tests/integrator_perf_test.cpp on the HIP path: the per-ray step statistics of its report (ray counts, min / mean /
    median / percentiles / max, totals for RK4 and RK45) against the CPU build's.  RK4 columns equal; RK45 within the +-1 step per
    ray that its adaptive step control turns libm differences into."""
    exe = need("integrator_perf_test")
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, r.stderr[-1000:]
    def table(text):
        rows = {}
        for l in text.splitlines():
            if l.startswith(("Total rays", "Valid rays", "Invalid rays", "Steps per ray", "Total steps", "Total func")):
                parts = l.split()
                rows[" ".join(parts[:-2])] = (float(parts[-2]), float(parts[-1]))
        return rows
    got, want = table(r.stdout), table(open(os.path.join(APPS, "integrator_perf_test.txt")).read())
    assert set(got) == set(want) and len(want) == 11
    for key, (w4, w45) in want.items():
        g4, g45 = got[key]
        assert g4 == w4, (key, g4, w4)                                     # RK4: identical
        assert abs(g45 - w45) <= max(2.0, 2e-3 * abs(w45)), (key, g45, w45)


def test_reference_self_tests_pass_on_the_hip_path():
    """src/tests/raytrace_rk4_test.cpp and emissivity_rk45_test.cpp, built against the HIP path, still PASS and
    report the same classification counts as their CPU runs."""
    for app, keys in (("raytrace_rk4_test", ("Hit disc", "Hit r_max", "Hit horizon", "Skipped")), ("emissivity_rk45_test", ())):
        exe = need(app)
        r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        lines = r.stdout.strip().splitlines()
        assert "PASS" in lines[-5:], lines[-8:]
        ref = open(os.path.join(APPS, f"{app}.txt")).read()
        for k in keys:
            gl = [l for l in lines if k in l][0].split()
            rl = [l for l in ref.splitlines() if k in l][0].split()
            assert gl == rl, (gl, rl)


@pytest.mark.parametrize("par,app", [("caustic_discplane", "caustic_discplane"), ("caustic_discplane_rk45", "caustic_discplane"),
                                     ("caustic_sourceplane", "caustic_sourceplane"), ("caustic_plane", "caustic_plane")])
def test_caustic_apps_match_cpu_output(par, app):
    """the caustic applications (ImagePlaneBundles 5-ray bundles / ImagePlane, DiscWithISCO and FlatPlane
    destinations, rdot_flips / equatorial_crossings outputs), unmodified, on the HIP path.  Classification maps must agree on
    >= 99 % of the pixels (a photon-ring pixel may change image order), hit coordinates to 1e-6 on >= 99 % of them (RK4 and RK45); det(J)
    is a central difference over 1 % of a pixel, i.e. it amplifies end-point differences by ~1e2..1e3, and is held to 1e-3."""
    exe = need(app)
    with tempfile.TemporaryDirectory() as w:
        out = os.path.join(w, "out.fits")
        subprocess.run([exe, f"--parfile={os.path.join(APPS, par + '.par')}", f"--outfile={out}"], check=True, stdout=subprocess.DEVNULL, env=ENV, timeout=600)
        got = {h["name"]: h for h in fits_lite.read(out)}
        got_cards = fits_lite.header_cards(out)
    want = {h["name"]: h for h in fits_lite.read(os.path.join(APPS, par + ".fits"))}
    assert list(got) == list(want)
    # headers written by this repo's fits_output.h vs cfitsio's: identical except for cards that carry a ray count
    for gc_, wc_ in zip(got_cards, fits_lite.header_cards(os.path.join(APPS, par + ".fits"))):
        diff = [(a, b) for a, b in zip(gc_, wc_) if a != b]
        assert len(gc_) == len(wc_) and all(a[:8] == b[:8] and a[:8].strip() in COUNT_KEYS for a, b in diff), diff[:3]
    rk45 = par.endswith("rk45")
    for name in list(want)[1:]:
        g, w = got[name]["data"], want[name]["data"]
        nan_same = np.isnan(g) == np.isnan(w)
        assert nan_same.mean() >= 0.99, (name, nan_same.mean())
        ok = ~np.isnan(w) & ~np.isnan(g)
        if name in ("SIGN_J", "ORDER", "HIT", "HIT_PLANE", "ESCAPED", "RDOT_FLIPS", "EQUAT_CROSS"):
            assert (g[ok] == w[ok]).mean() >= 0.99, (name, (g[ok] == w[ok]).mean())
            continue
        rtol = 1e-3 if name == "DET_J" else 1e-6                 # (RK4 and RK45 alike)
        close = np.isclose(g[ok], w[ok], rtol=rtol, atol=1e-9)
        need_frac = 0.97 if name == "DET_J" else 0.99
        if name in ("PHI", "PHI_S"):                       # angles may differ by a 2 pi wrap on the branch cut
            close |= np.isclose(np.abs(g[ok] - w[ok]), 2 * np.pi, rtol=0, atol=1e-5)
        assert close.mean() >= need_frac, (name, close.mean())


# ---- the device-resident applications (raytrace_cpu_amd/apps): same inputs, same output files ----------------------
NATIVE = os.path.join(gc.ROOT, "raytrace_cpu_amd", "apps", "_build")


def need_native(app):
    path = os.path.join(NATIVE, app)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (make -C raytrace_cpu_amd/apps)")
    return path


@pytest.mark.parametrize("arithmetic", ["hybrid", "strict"])
def test_native_emissivity_app_matches_cpu_output(arithmetic):
    exe = need_native("kr_emissivity")
    with tempfile.TemporaryDirectory() as w:
        out = os.path.join(w, "out.dat")
        r = subprocess.run([exe, f"--parfile={os.path.join(APPS, 'emissivity.par')}", f"--outfile={out}", f"--arithmetic={arithmetic}", "--timing"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "timing: rays" in r.stdout
        got_text = open(out).read()
        rows = [l.split() for l in got_text.splitlines() if l.strip()]
    got, want = np.array([[float(x) for x in r_] for r_ in rows]), load_dat("emissivity.dat")
    want_text = open(os.path.join(APPS, "emissivity.dat")).read()
    assert got.shape == want.shape
    # columns 1-2 (radii, areas: host code) are the same TEXT as the reference's file
    assert [l[:40] for l in got_text.splitlines()] == [l[:40] for l in want_text.splitlines()]
    dcount = np.abs(got[:, 2] - want[:, 2])
    assert dcount.max() <= 1
    same = dcount == 0
    assert (~same).sum() <= 2, (~same).sum()
    for col in (3, 4, 5, 6):
        g, w_ = got[same, col], want[same, col]
        assert (np.isnan(g) == np.isnan(w_)).all()
        ok = ~np.isnan(w_)
        np.testing.assert_allclose(g[ok], w_[ok], rtol=1e-6, err_msg=f"column {col}")


@pytest.mark.parametrize("par", ["imageplane_rk4", "imageplane_rk45"])
def test_native_imageplane_app_matches_cpu_output(par):
    exe = need_native("kr_imageplane_disc_image")
    with tempfile.TemporaryDirectory() as w:
        out = os.path.join(w, "out.fits")
        r = subprocess.run([exe, f"--parfile={os.path.join(APPS, par + '.par')}", f"--outfile={out}"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        got = {h["name"]: h for h in fits_lite.read(out)}
        got_cards = fits_lite.header_cards(out)
        size = os.path.getsize(out)
    golden = os.path.join(APPS, par + ".fits")
    want = {h["name"]: h for h in fits_lite.read(golden)}
    assert size == os.path.getsize(golden) and got_cards == fits_lite.header_cards(golden)      # every header byte, incl. DISCRAYS
    for name in list(want)[1:]:
        g, w_ = got[name]["data"], want[name]["data"]
        assert (np.isnan(g) == np.isnan(w_)).all(), name
        ok = ~np.isnan(w_)
        close_to_cpu("test_native_imageplane_app_matches_cpu_output", f"{par}-{name}", g[ok], w_[ok], rtol=1e-6, atol=1e-12)
