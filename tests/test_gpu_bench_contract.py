"""GPU: bench.py's output line keeps the driver's contract, and the hybrid launch keeps its two kernels concurrent in a
process that has torch.distributed / RCCL initialised (the way every multi-GPU rank runs).  HIP multiplexes streams onto a
few hardware queues per priority level; with RCCL's streams present a default-priority side stream ended up behind the
caller's stream and the two launches serialised (200 ms instead of 102 ms) -- hence its own priority level (kr_trace.hip)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_and_kernel_overlap_under_rccl():
    env = dict(os.environ, KR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE json line ...
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1, r.stdout[:400]     # ... and nothing else on stdout (RCCL's version banner goes to stderr)
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["metric"] == "rays_per_sec" and d["unit"] == "rays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f64"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "longest_ray"):
        assert key in roof, key
    assert roof["longest_ray"]["steps"] == roof["longest_ray"]["strict_side_steps"] > 30000        # the polar-axis crawler of the beta = -pi column
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert d["config"]["rays_on_strict_side_launch"] == 3162
    # value is what the timed region did: rays * steps / wall
    assert abs(d["value"] - d["config"]["rays_total"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    # concurrency: overlapped, the hybrid pass takes max(side launch ~100 ms, fast launch ~90 ms) = ~1.2 x the all-fast pass (85 ms);
    # serialised it would be their sum, ~2.2 x
    fast_ms = d["other_arithmetic_modes"]["fast"]["avg_kernel_ms"]
    assert roof["avg_kernel_ms"] < 1.6 * fast_ms, (roof["avg_kernel_ms"], fast_ms)


def _bench_line(extra, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-fast-math-extra"] + extra,
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_returning_radiation_line_is_the_same_with_and_without_the_collective_path():
    """The grouped, pipelined returning-radiation pass writes into whichever of the two result tables the pass was given (under torch.distributed
    the table is double-buffered for the exchange beside the next pass): the fractions of the last pass are finite and the same either way."""
    args = ["--workload", "return_radiation", "--radii", "12", "--rays", "2e5"]
    plain = _bench_line(args)
    dist = _bench_line(args, dict(KR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29562", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    single = _bench_line(args + ["--rr-groups", "1"])
    for key, want in plain["fractions_escape_return_lost"].items():
        for other in (dist, single):
            got = other["fractions_escape_return_lost"][key]
            assert all(abs(g - w) <= 1e-12 and g == g for g, w in zip(got, want)), (key, got, want)
    assert dist["rccl_ranks"] == 1 and plain["rccl_ranks"] is None
    assert dist["rk_steps_per_launch"] == plain["rk_steps_per_launch"] == single["rk_steps_per_launch"]


@pytest.mark.parametrize("extra,keys", [(["--workload", "emissivity", "--rays", "3e5"], ("disc_hits",)),
                                        (["--workload", "imageplane", "--rays", "2e5"], ("disc_hits", "lit_pixels")),
                                        (["--workload", "imageplane", "--rays", "2e5", "--image-exchange", "allreduce"], ("disc_hits", "lit_pixels")),
                                        (["--workload", "emissivity", "--rays", "3e5", "--scaling", "strong"], ("disc_hits",))],
                         ids=["emissivity", "imageplane-gather", "imageplane-allreduce", "emissivity-strong"])
def test_summaries_agree_with_and_without_the_collective_path(extra, keys):
    """What the last pass left in the (double-buffered, exchanged-beside-the-next-pass) result buffer under torch.distributed with one rank is
    what the plain run leaves: disc-hit counts and lit pixels are exact counts."""
    plain = _bench_line(extra)
    dist = _bench_line(extra, dict(KR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29563", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    for k in keys:
        assert plain[k] == dist[k] and plain[k] > 0, (k, plain[k], dist[k])
    assert dist["rccl_ranks"] == 1 and dist["rk_steps_per_launch"] == plain["rk_steps_per_launch"]


@pytest.mark.parametrize("workload,key,rays", [("emissivity", "pipeline_bins_check", "4e5"), ("imageplane", "pipeline_planes_check", "2.5e5")])
def test_cpu_baseline_leg_checks_the_timed_pipeline(workload, key, rays):
    """bench.py's cpu_baseline leg on a small sample: the line carries `cpu_baseline` with its check of the TIMED, device-resident pipeline
    (device-built rays, fused passes) beside the checks of the reference-constructed rays -- for the PointSource every bin count, the disc-ray count
    and the step total equal the CPU's (the device constructor carries the reference's bits)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--rays", rays, "--cpu-sample-rays", "60000", "--steps", "1", "--warmup", "1",
                        "--no-fast-math-extra"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1
    chk = cb[key]
    if workload == "emissivity":
        assert chk["bins_count_mismatch"] == 0 and chk["rk_steps_equal"] and chk["disc_rays_device"] == chk["disc_rays_cpu"] > 1000, chk
        assert chk["max_rel_diff_on_matching_bins"] <= 1e-6
    else:
        # device-built image-plane rays differ from the reference constructor's where glibc is not correctly rounded (~1e-3 of the rays, in a last bit)
        assert chk["disc_rays_cpu"] > 1000 and abs(chk["disc_rays_device"] - chk["disc_rays_cpu"]) <= 2, chk
        assert chk["pixels_count_mismatch"] <= 4, chk
        assert max(chk["max_rel_diff_of_pixel_sums(RADIUS, ENSHIFT, FLUX, TIME)"].values()) <= 1e-6, chk
