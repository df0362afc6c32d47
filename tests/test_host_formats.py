"""CPU: the host-side file formats are byte-comparable with what the reference writes.
tests/cpp/formats_test drives raytrace_cpu_amd/host/include/{text_output,fits_output,par_file,par_args,array,disc}.h and the
writers the device-resident applications use (raytrace_cpu_amd/apps/*.h); the expected bytes are the output files of the
reference's own CPU builds (tests/golden/apps/, written with cfitsio 3.47 / libstdc++; tests/golden/make_app_golden.sh)."""
import os
import subprocess

import numpy as np
import pytest

import fits_lite
import golden_cases as gc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APPS = os.path.join(gc.GOLDEN_DIR, "apps")
EXE = os.path.join(ROOT, "tests", "cpp", "formats_test")


@pytest.fixture(scope="module")
def exe():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), EXE], check=True)
    return EXE


@pytest.mark.parametrize("name", ["emissivity.dat", "emissivity_rd.dat"])
def test_text_table_is_byte_identical(exe, tmp_path, name):
    """TextOutput: width 20, scientific, 8 digits, integer count column, '-nan' for empty bins (text_output.h:28)."""
    out = tmp_path / name
    subprocess.run([exe, "dat", os.path.join(APPS, name), str(out)], check=True, stdout=subprocess.DEVNULL)
    assert out.read_bytes() == open(os.path.join(APPS, name), "rb").read()


@pytest.mark.parametrize("stem", ["imageplane_rk4", "imageplane_rk45"])
def test_fits_file_is_byte_identical(exe, tmp_path, stem):
    """The whole 7-HDU file: cfitsio's card formatting (HIERARCH, %.15G reals, quoted strings, COMMENT), block padding,
    big-endian IEEE data with the NaN bit patterns of the empty pixels."""
    golden = os.path.join(APPS, stem + ".fits")
    hdus = fits_lite.read(golden)
    assert [h["name"] for h in hdus] == ["PRIMARY", "FLUX", "RADIUS", "PHI", "ENSHIFT", "TIME", "EMIS"]
    raw = open(golden, "rb").read()
    planes = tmp_path / "planes.bin"
    with open(planes, "wb") as f:
        for h in hdus[1:]:
            # the file holds [iy][ix] big-endian; Array2D is [ix][iy] native: copy the bits (NaN payloads included)
            nx, ny = int(h["header"]["NAXIS1"]), int(h["header"]["NAXIS2"])
            bits = np.frombuffer(raw, dtype=">u8", count=nx * ny, offset=h["data_offset"]).reshape(ny, nx)
            f.write(np.ascontiguousarray(bits.T).astype("<u8").tobytes())
    out = tmp_path / (stem + ".fits")
    subprocess.run([exe, "fits", os.path.join(APPS, stem + ".par"), str(planes), str(hdus[0]["header"]["DISCRAYS"]), str(out)],
                   check=True, stdout=subprocess.DEVNULL)
    got = out.read_bytes()
    assert len(got) == len(raw)
    if got != raw:
        first = next(i for i in range(len(raw)) if got[i] != raw[i])
        card = first // 80 * 80
        pytest.fail(f"first difference at byte {first}: got {got[card:card + 80]!r} want {raw[card:card + 80]!r}")


def test_parameter_file_and_arguments(exe):
    r = subprocess.run([exe, "par", os.path.join(APPS, "emissivity.par"), "--spin=0.5", "positional_arg"], check=True, capture_output=True, text=True)
    lines = dict(l.split(": ", 1) for l in r.stdout.strip().splitlines() if ": " in l)
    assert r.stdout.splitlines()[0] == "source: 0 10 0.001 1.5707"
    assert lines["spin"] == "0.5"                       # --spin overrides the file
    assert lines["Nr"] == "30" and lines["logbin_r"] == "1"
    assert lines["outfile"] == "unused.dat" and lines["gamma(default)"] == "2" and lines["positional"] == "1"
    assert lines["missing"] == "ParameterFile ERROR : ParameterFile ERROR: no_such_key not found in parameter file"   # reference wording (sic)
    assert lines["unparsable"].startswith("ParameterFile ERROR : Could not parse value of outfile")


def test_disc_area_matches_reference_tables(exe):
    """Column 2 of the emissivity tables is integrate_disc_area(r_i, r_i * dr, spin) (emissivity.cpp:76-79)."""
    rows = [l.split() for l in open(os.path.join(APPS, "emissivity.dat")) if l.strip()]
    r = np.array([float(x[0]) for x in rows])
    area = np.array([float(x[1]) for x in rows])
    import ctypes as C
    import oracle_lib as ol
    r_isco = ol.oracle().kro_kerr_isco(C.c_double(0.998), 1)
    dr = float(np.exp(np.log(500.0 / r_isco) / len(rows)))
    for i in (0, 1, len(rows) // 2, len(rows) - 1):
        r0 = r_isco * dr ** i
        got = float(subprocess.run([exe, "area", "0.998", repr(r0), repr(r0 * dr)], check=True, capture_output=True, text=True).stdout)
        assert abs(r0 - r[i]) <= 1e-8 * r[i]
        assert abs(got - area[i]) <= 1e-8 * area[i], (i, got, area[i])


MANDATORY = ("SIMPLE", "BITPIX", "NAXIS", "NAXIS1", "NAXIS2", "EXTEND", "XTENSION", "PCOUNT", "GCOUNT", "END")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(APPS) if f.endswith(".fits")))
def test_every_reference_header_card_round_trips(exe, tmp_path, name):
    """Every keyword / comment card of every FITS file the reference's CPU builds wrote (imageplane + caustic programs:
    HIERARCH names, split COMMENTs, %.15G and exponent-form reals, long and short strings, integers) is regenerated,
    byte for byte, by replaying it as a FITSOutput call."""
    golden = os.path.join(APPS, name)
    hdus = fits_lite.read(golden)
    cards = fits_lite.header_cards(golden)
    spec = []
    for i, (h, cs) in enumerate(zip(hdus, cards)):
        spec.append("P" if i == 0 else f"I {h['header']['NAXIS1']} {h['header']['NAXIS2']}")
        for c in cs:
            key = c[:8].strip()
            if not c.strip() or key in MANDATORY or (i == 0 and c.startswith("COMMENT   ")):
                continue
            if key == "COMMENT":
                spec.append("C " + c[8:].rstrip())
                continue
            if c.startswith("HIERARCH "):
                key, rest = c[9:].split("=", 1)
                key = key.strip()
            else:
                rest = c[10:]
            rest = rest.strip()
            if rest.startswith("'"):
                end = rest.index("'", 1)
                value, kind = rest[1:end].rstrip(), "s"
                comment = rest[end + 1:].split("/", 1)[1][1:] if "/" in rest[end + 1:] else ""
            else:
                value, comment = (rest.split("/", 1) + [""])[:2]
                value, comment = value.strip(), comment[1:] if comment.startswith(" ") else comment
                kind = "b" if value in ("T", "F") else ("d" if any(ch in value for ch in ".E") else "l")
            if key == "EXTNAME":
                spec.append("E " + value)
            else:
                spec.append(f"K{kind} {key}|{comment.rstrip()}|{value}")
    spec_file = tmp_path / "spec.txt"
    spec_file.write_text("\n".join(spec) + "\n")
    out = tmp_path / name
    subprocess.run([exe, "cards", str(spec_file), str(out)], check=True, stdout=subprocess.DEVNULL)
    got = fits_lite.header_cards(str(out))
    assert len(got) == len(cards)
    for g, w in zip(got, cards):
        # a comment cut at column 80 by cfitsio cannot be told from one that ended there: compare what the card holds
        assert g == w, [(a, b) for a, b in zip(g, w) if a != b][:3]
    assert os.path.getsize(out) == os.path.getsize(golden)


def test_non_ascii_header_text_becomes_blanks(exe, tmp_path):
    """caustic_sourceplane.cpp writes a comment with a UTF-8 dash; cfitsio stores three blanks for its three bytes."""
    spec = tmp_path / "spec.txt"
    spec.write_text("P\nC mapping \u2014 source sphere\n", encoding="utf-8")
    out = tmp_path / "x.fits"
    subprocess.run([exe, "cards", str(spec), str(out)], check=True, stdout=subprocess.DEVNULL)
    cards = fits_lite.header_cards(str(out))[0]
    assert "COMMENT mapping     source sphere".ljust(80) in cards


def test_fits_writer_reports_a_failed_write(exe, tmp_path):
    """The threaded writer records I/O errors of its background pwrite slices; close() must surface them (a program that goes on to print
    "Done" over a truncated file is worse than one that stops).  The 2-MB image is written under a 1-MB file-size limit (EFBIG)."""
    r = subprocess.run([exe, "fullimage", str(tmp_path / "cut.fits"), str(1 << 20)], capture_output=True, text=True)
    assert r.returncode == 3, (r.returncode, r.stderr[-300:])
    assert "writing the file failed" in r.stderr
    ok = subprocess.run([exe, "fullimage", str(tmp_path / "ok.fits"), "0"], capture_output=True, text=True)
    assert ok.returncode == 0 and os.path.getsize(tmp_path / "ok.fits") % 2880 == 0
