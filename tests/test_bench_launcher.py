"""CPU: `python bench.py --gpus N` (N > 1) starts its own ranks -- torch.distributed.run as a CHILD process, before anything touches the
GPU -- and relays rank 0's line and the child's exit code.  (The reference's parallel loop needs no launcher, raytracer.cpp:104.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_argv():
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "3", "--workload", "imageplane"]
    cmd = bench.self_launch_argv(argv, 8, 29555, python="/usr/bin/python3")
    assert cmd[:3] == ["/usr/bin/python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv                           # the bench arguments are passed on unchanged, after the script
    assert 1024 < bench.free_port() < 65536


def test_rankless_multi_gpu_invocation_starts_its_ranks_and_relays_their_failure():
    # no GPU here: every rank must fail with the no-GPU message (from the CHILD ranks, not a launcher complaint), rc != 0
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert "needs a GPU" in r.stderr
    assert "must be launched with" not in r.stderr


def test_a_rank_with_the_wrong_world_size_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
