"""Builds raytrace_cpu_amd/csrc/libkrtrace.so with hipcc for gfx950 (cross-compiles without a GPU).

  python -m raytrace_cpu_amd._build [--force] [--verbose]

-ffp-contract=off keeps every product/sum rounded separately, like the reference's CPU build: the only
arithmetic difference from the CPU path is then the device libm (sin, cos, pow).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["kr_trace.hip", "kr_post.hip", "kr_capi.hip"]
HEADERS = ["kr_device.hpp", "kr_crmath.hpp", "kr_arith.hpp", "kr_fast.hpp", "kr_rk45.hpp", "kr_post_device.hpp", "kr_sincos.hpp", "kr_replay.hpp", "kr_common.hpp", os.path.join("..", "..", "include", "kr_trace.h")]
LIB = os.path.join(CSRC, "libkrtrace.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), tag=""):
    """tag != "": an experiment variant, built next to the product library as libkrtrace_<tag>.so (never loaded by default)."""
    deps = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    lib = LIB if not tag else LIB.replace(".so", f"_{tag}.so")
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", f"{'_' + tag if tag else ''}.o"))
        objs.append(o)
        if force or _stale(o, [s] + deps):
            jobs.append([hipcc(), *FLAGS, *extra_flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=3) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
