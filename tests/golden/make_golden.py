#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libkr_ref.so, built by
oracle/Makefile from the sources under /root/reference -- run in the build container only).

Each fixture holds, for one ray source of tests/golden_cases.py:
  init            rays[] right after the reference ctor + redshift_start()   (144-B Ray<double> records)
  final__<run>    rays[] after run_raytrace() + range_phi() + redshift()     for every run of the case
  steps__<run>    sum of per-ray steps taken by that run (int64)
Fields the reference leaves indeterminate are given defined values first (oracle_lib.RefSource.sanitize).
Fixtures are data only: inputs and the reference's outputs.

usage: python tests/golden/make_golden.py [case ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden_cases as gc  # noqa: E402
import oracle_lib as ol  # noqa: E402
from raytrace_cpu_amd import capi  # noqa: E402


def run_reference(case, params):
    src = ol.RefSource(case["source"])
    V, rev, proj = case["start"]
    src.lib.ref_redshift_start(src.h, V, rev, proj)
    init = src.snapshot()
    steps_before = src.rays["steps"].copy()
    src.run(params)
    src.lib.ref_range_phi(src.h, -np.pi, np.pi)
    V, rev, proj = case["post"]
    if params.stop_kind == capi.STOP_THETA:
        src.lib.ref_redshift(src.h, V, rev, proj, 0)
    else:
        sp = (ol.C.c_double * 4)(*params.stop_params)
        src.lib.ref_redshift_dest(src.h, params.stop_kind, sp, rev)
    final = src.snapshot()
    steps = int((np.abs(final["steps"].astype(np.int64)) - np.abs(steps_before.astype(np.int64)))[final["steps"] != -1].sum())
    src.close()
    return init, final, steps


def main(argv):
    assert ol.ref() is not None, "needs oracle/_ref/libkr_ref.so (make -C oracle ref, in the build container)"
    all_cases = gc.cases()
    names = argv or list(all_cases)
    for name in names:
        case = all_cases[name]
        out = {}
        for run, params in case["runs"].items():
            init, final, steps = run_reference(case, params)
            if "init" in out:
                assert not ol.rays_equal_bitwise(out["init"], init), "reference init not deterministic?"
            out["init"] = init
            out[f"final__{run}"] = final
            out[f"steps__{run}"] = np.int64(steps)
            print(f"{name}/{run}: {len(init)} rays, {steps} steps")
        np.savez_compressed(gc.golden_path(name), **out)
        print(f"  -> {gc.golden_path(name)} ({os.path.getsize(gc.golden_path(name)) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main(sys.argv[1:])
