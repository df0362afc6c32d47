"""CPU: the closed-form replay of n identical fp64 additions (raytrace_cpu_amd/csrc/kr_replay.hpp, compiled for the host) is
bit-identical to the literal loop `for (i < n) x = x + dx` in every regime: growing, shrinking through zero, absorbed, exact
ties, a few ulps per step with many binade crossings, subnormals, from zero, just below a power of two, long runs of tiny
increments (what step_rk45's fixed-point replay feeds it: t and phi of a captured ray, up to 1e5 additions each)."""
import os
import subprocess

import golden_cases as gc


def test_replay_equals_the_literal_loop(tmp_path):
    exe = str(tmp_path / "replay_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(gc.ROOT, "tests", "replay_check.cpp")])
    out = subprocess.check_output([exe, "150000"], text=True).strip().splitlines()
    cases, bad, literal, total = (int(x) for x in out[-1].split())
    assert cases == 150000 and bad == 0, out[:6]
    assert literal < 0.2 * total              # the closed form did the work (ties / subnormals / boundary steps are the literal ones)
