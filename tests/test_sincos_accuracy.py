"""CPU: the sincos the strict path uses for polar angles (raytrace_cpu_amd/csrc/kr_sincos.hpp: Cody-Waite reduction, double-double
head, one final rounding), compiled for the host: correctly rounded as far as long double can tell (<= 0.5005 ulp of sinl / cosl) and
bit-equal to glibc's double sin / cos wherever glibc itself is correctly rounded (>= 99.7 % of arguments)."""
import os
import subprocess

import golden_cases as gc


def test_compact_sincos_error_bounds(tmp_path):
    exe = str(tmp_path / "sincos_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(gc.ROOT, "tests", "sincos_check.cpp"), "-lm"])
    for args in (["2000000", "-1.0", "4.2"], ["1000000", "-40", "40"], ["500000", "1.5707", "1.5709"], ["500000", "-1e-3", "1e-3"]):
        n, ms, mc, gs, gcs, es, ec = (float(x) for x in subprocess.check_output([exe] + args, text=True).split())
        assert ms <= 0.5005 and mc <= 0.5005, (args, ms, mc)  # vs long double (whose own rounding is the 0.0005)
        assert gs <= 1.0 and gcs <= 1.0, (args, gs, gcs)      # vs glibc double
        assert es > 0.997 and ec > 0.997, (args, es, ec)      # bit-equal to glibc except where glibc is not correctly rounded (measured 99.83-99.99 %)
    # the small-angle branch (|x| < 2^-7, kr_sincos_small_f64): polar-axis rays live there for their whole life
    for args in (["2000000", "-0.0078125", "0.0078125"], ["1000000", "0.00099999", "0.00100001"], ["500000", "-1e-9", "1e-9"]):
        n, ms, mc, gs, gcs, es, ec = (float(x) for x in subprocess.check_output([exe] + args, text=True).split())
        assert gs <= 1.0 and gcs <= 1.0, (args, gs, gcs)
        assert es > 0.9999 and ec > 0.9999, (args, es, ec)     # correctly rounded in practice: bit-equal to glibc


def test_fast_path_sincos_error_bounds(tmp_path):
    """kr_sincos_fast_f64 (fast arithmetic path only): no tail corrections, a few-ulp contract -- held to 2 ulp."""
    exe = str(tmp_path / "sincos_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(gc.ROOT, "tests", "sincos_check.cpp"), "-lm"])
    for args in (["2000000", "-1.0", "4.2"], ["1000000", "-40", "40"], ["500000", "1.5707", "1.5709"], ["500000", "-1e-3", "1e-3"]):
        n, ms, mc, gs, gcs, es, ec = (float(x) for x in subprocess.check_output([exe] + args + ["fast"], text=True).split())
        assert ms <= 2.0 and mc <= 2.0, (args, ms, mc)
