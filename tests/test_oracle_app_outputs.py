"""CPU: the oracle's full pipeline (source -> redshift_start -> trace -> range_phi -> redshift -> reducer) reproduces
the OUTPUT FILES of the reference's own applications (tests/golden/apps/, written by the CPU builds of
src/emissivity/emissivity.cpp and src/imageplane/imageplane_disc_image.cpp; see tests/golden/make_app_golden.sh).
This pins the reducers, which the reference only has inside its app main()s."""
import ctypes as C
import os

import numpy as np
import pytest

import fits_lite
import golden_cases as gc
import oracle_lib as ol
from raytrace_cpu_amd import capi

APPS = os.path.join(gc.GOLDEN_DIR, "apps")


def read_par(name):
    d = {}
    for line in open(os.path.join(APPS, name)):
        line = line.split("#")[0]
        if "=" in line:
            k, v = line.split("=", 1)
            d[k.strip()] = v.strip()
    return d


def emissivity_setup(par_name):
    par = read_par(par_name)
    src = [float(x) for x in par["source"].split()]
    spin = float(par["spin"])
    spec = ol.pointsource_spec(src, float(par.get("V", 0)), spin, float(par["dcosalpha"]), float(par["dbeta"]),
                               cosalpha0=-0.995, cosalphamax=0.995, beta0=-np.pi, betamax=np.pi)   # emissivity.cpp:38-42
    o = ol.oracle()
    nr = int(par["Nr"])
    b = capi.EmisBins()
    b.r_isco = o.kro_kerr_isco(spin, 1)
    b.r_min = b.r_isco                                             # rmin default -1 -> r_isco, :58
    b.dr = float(np.exp(np.log(500.0 / b.r_min) / nr))             # r_disc reads key r_esc, default 500, :51,:59
    b.gamma, b.spin = 2.0, spin
    b.num_primary_rays = float(int(((spec.cosalphamax - spec.cosalpha0) / spec.dcosalpha) * ((spec.betamax - spec.beta0) / spec.dbeta)))
    b.nr, b.logbin = nr, 1
    return spec, b, spin


def load_dat(name):
    rows = [l.split() for l in open(os.path.join(APPS, name)) if l.strip()]
    return np.array([[float(x) for x in r] for r in rows])


def check_against_dat(dat, hist, rtol=2e-8):
    """columns: r, area, count, flux/area, emis/area, <g>, <t> (emissivity.cpp:128-146), 8 significant digits."""
    area = dat[:, 1]
    assert (dat[:, 2].astype(np.int64) == hist["count"]).all()
    n = hist["count"]
    with np.errstate(invalid="ignore", divide="ignore"):
        expect = {3: hist["flux"] / area, 4: hist["emis"] / area, 5: hist["sum_redshift"] / n, 6: hist["sum_time"] / n}
    for col, e in expect.items():
        assert (np.isnan(dat[:, col]) == np.isnan(e)).all(), col
        ok = ~np.isnan(e)
        np.testing.assert_allclose(dat[ok, col], e[ok], rtol=rtol, err_msg=f"column {col}")


def oracle_hist(bins, rays):
    nr = bins.nr
    count = np.zeros(nr, dtype=np.int64)
    flux, emis, sg, stt = (np.zeros(nr) for _ in range(4))
    dc = C.c_int64()
    ol.oracle().kro_reduce_emissivity_f64(C.byref(bins), ol.ptr(rays), len(rays), ol.ptr(count), ol.ptr(flux), ol.ptr(emis), ol.ptr(sg), ol.ptr(stt), C.byref(dc))
    return {"count": count, "flux": flux, "emis": emis, "sum_redshift": sg, "sum_time": stt, "disc_count": dc.value}


def test_emissivity_app_output():
    # src/emissivity/emissivity.cpp: RK45, theta_max = pi/2, r_max = r_esc default 1000; range_phi; redshift(-1, false)
    spec, bins, spin = emissivity_setup("emissivity.par")
    o = ol.oracle()
    rays = ol.oracle_pointsource(spec)
    o.kro_redshift_start_f64(spin, spec.V, 0, 0, ol.ptr(rays), len(rays))
    p = capi.default_params(spin)
    p.integrator, p.r_max = capi.RK45, 1000.0
    out, _ = ol.oracle_trace(p, rays)
    o.kro_range_phi_f64(-np.pi, np.pi, ol.ptr(out), len(out))
    o.kro_redshift_f64(spin, -1.0, 0, 0, 0, ol.ptr(out), len(out))
    dat = load_dat("emissivity.dat")
    assert dat.shape == (bins.nr, 7)
    np.testing.assert_allclose(dat[:, 0], bins.r_min * bins.dr ** np.arange(bins.nr), rtol=2e-8)
    check_against_dat(dat, oracle_hist(bins, out))


def image_setup(par_name):
    par = read_par(par_name)
    g = lambda k, d=None: float(par.get(k, d))
    Nx = int(par["Nx"])
    img = int(par.get("img_Nx", Nx))
    x0, xmax = g("x0"), g("xmax")
    y0, ymax = g("y0", x0), g("ymax", xmax)
    dx, dy = (xmax - x0) / Nx, (ymax - y0) / int(par.get("Ny", Nx))
    spin = g("spin")
    spec = ol.imageplane_spec(g("dist"), g("incl"), x0, xmax, dx, y0, ymax, dy, spin, g("plane_phi0", 0), g("precision", 100))
    b = capi.ImageBins()
    b.x0, b.y0, b.img_dx, b.img_dy = x0, y0, (xmax - x0) / img, (ymax - y0) / img
    b.r_isco, b.r_disc = ol.oracle().kro_kerr_isco(spin, 1), g("r_disc")
    b.q1, b.rb1, b.q2, b.rb2, b.q3 = g("q1", 3), g("rb1", 4), g("q2", 3), g("rb2", 10), g("q3", 3)
    b.img_nx, b.img_ny, b.flip_image, b.pad = img, img, 1, 0
    method = {"euler": capi.EULER, "rk4": capi.RK4}.get(par.get("integrator", "rk45"), capi.RK45)
    return spec, b, spin, method, g("dist")


def finish_image(red):
    """the divisions of imageplane_disc_image.cpp:165-174 (0/0 -> NaN in empty pixels, flux guarded)"""
    n = red["nrays"].astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        out = {"FLUX": np.where(n > 0, red["flux"] / np.where(n > 0, n, 1), red["flux"])}
        for name, k in (("RADIUS", "r"), ("PHI", "phi"), ("ENSHIFT", "enshift"), ("TIME", "time"), ("EMIS", "emis")):
            out[name] = red[k] / n
    return out


def compare_fits(path, planes, shape, rtol):
    hdus = {h["name"]: h for h in fits_lite.read(path)}
    for name, got in planes.items():
        want = hdus[name]["data"]                      # written with write_image(Array2D, Nx, Ny): FITS axis1 = first index
        got2 = got.reshape(shape)
        cands = [want, want.T]
        ok = False
        for w in cands:
            if w.shape == got2.shape and (np.isnan(w) == np.isnan(got2)).all() and np.allclose(w[~np.isnan(w)], got2[~np.isnan(w)], rtol=rtol, atol=1e-300):
                ok = True
        assert ok, name
    return hdus


@pytest.mark.parametrize("par", ["imageplane_rk4.par", "imageplane_rk45.par"])
def test_imageplane_app_output(par):
    spec, bins, spin, method, dist = image_setup(par)
    o = ol.oracle()
    rays = ol.oracle_imageplane(spec)
    o.kro_redshift_start_f64(-spin, 0.0, 1, 0, ol.ptr(rays), len(rays))
    p = capi.default_params(-spin)
    p.integrator, p.r_max = method, 1.1 * dist
    out, _ = ol.oracle_trace(p, rays)
    o.kro_redshift_f64(-spin, -1.0, 1, 0, 0, ol.ptr(out), len(out))
    o.kro_range_phi_f64(-np.pi, np.pi, ol.ptr(out), len(out))
    from test_gpu_parity import oracle_reduce_image
    red = oracle_reduce_image(bins, out)
    hdus = compare_fits(os.path.join(APPS, par.replace(".par", ".fits")), finish_image(red), (bins.img_nx, bins.img_ny), rtol=1e-12)
    assert int(float(hdus["PRIMARY"]["header"]["DISCRAYS"])) == red["disc_count"]
