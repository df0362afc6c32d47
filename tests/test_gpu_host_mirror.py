"""GPU: C++ test program over the host-side mirror of the reference class API (tests/cpp/host_api_test.cpp)."""
import os
import subprocess

import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu


def test_host_api_program():
    d = os.path.join(gc.ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-s", "-C", os.path.join(gc.ROOT, "raytrace_cpu_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", d])
    r = subprocess.run([os.path.join(d, "host_api_test")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASS"), r.stdout[-3000:] + r.stderr[-1000:]


def test_krtrace_steplim_environment_knob():
    """KRTRACE_STEPLIM bounds the rays of an unchanged program where it passes no step limit itself ('one ray is one sequential chain')."""
    d = os.path.join(gc.ROOT, "tests", "cpp")
    r = subprocess.run([os.path.join(d, "host_api_test"), "steplim-env"], capture_output=True, text=True, timeout=600, env=dict(os.environ, KRTRACE_STEPLIM="40"))
    assert r.returncode == 0 and r.stdout.strip().endswith("PASS"), r.stdout[-3000:] + r.stderr[-1000:]


def test_show_progress_reports_from_the_running_kernels():
    """run_raytrace(show_progress) through the class mirror: the integrator's banner, then report lines whose counts are multiples of |show_progress|,
    increasing, and the newline of ProgressBar::done() (raytracer.cpp:76-85, :107-115, :126)."""
    import re
    d = os.path.join(gc.ROOT, "tests", "cpp")
    r = subprocess.run([os.path.join(d, "host_api_test"), "progress"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASS"), r.stdout[-3000:] + r.stderr[-1000:]
    assert "Running raytracer (RK4)..." in r.stdout
    marks = [(int(a), int(b)) for a, b in re.findall(r"^Ray\s+(\d+)/(\d+)$", r.stdout, flags=re.M)]
    assert marks and all(a % 250000 == 0 and 0 < a <= b for a, b in marks) and all(x[0] < y[0] for x, y in zip(marks, marks[1:])), r.stdout[-2000:]
