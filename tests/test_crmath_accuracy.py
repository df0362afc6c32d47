"""CPU: raytrace_cpu_amd/csrc/kr_crmath.hpp -- the device ImagePlane constructor's atan2 / asin / acos / tan -- compiled for the host and compared with
libquadmath's 113-bit results rounded once (tests/crmath_check.cpp): correctly rounded on (all but a few in 10^6 of) random arguments, including the
awkward ones (|x| -> 1 for asin / acos, tan next to pi/2, tiny arguments).  The C library the reference calls differs from that on ~1e-3 of arguments:
that, not the device, is what is left between a device-built image-plane ray and the reference constructor's."""
import os
import subprocess

import golden_cases as gc


def test_inverse_trig_and_tan_are_correctly_rounded(tmp_path):
    exe = str(tmp_path / "crmath_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(gc.ROOT, "tests", "crmath_check.cpp"), "-lquadmath", "-lm"])
    out = subprocess.run([exe, "500000"], capture_output=True, text=True, check=True).stdout.split("\n")
    rows = {l.split()[0]: l.split()[1:] for l in out if l.strip()}
    assert set(rows) == {"atan2", "asin", "acos", "tan"}
    for name, (n, misrounded, differs_libm, worst) in rows.items():
        n, misrounded, differs_libm, worst = int(n), int(misrounded), int(differs_libm), float(worst)
        assert misrounded <= (0 if name != "tan" else 5), (name, misrounded)      # measured: 0 / 0 / 0 / 1-4 in 2e6 (tan: the quotient of two 2^-66 pairs)
        assert worst <= 0.5001, (name, worst)
        assert differs_libm <= 0.005 * n, (name, differs_libm)                    # glibc 2.35: 0.06-0.2 % of arguments off the correctly rounded value
