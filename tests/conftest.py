import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the configuration the profiles and DESIGN figures describe (include/kr_trace.h::kr_configure_process): read when the HIP runtime starts,
# i.e. it must be in the environment before torch initialises below -- the library itself no longer sets it
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def torch_hip_runtime_first():
    """A process that uses both PyTorch-ROCm (which brings its own HIP runtime) and libkrtrace (linked against /opt/rocm's) must let
    torch initialise first: the other way round torch.cuda finds "No HIP GPUs" (seen when test_gpu_fullsize.py ran on its own, its
    torch-based return-radiation test after the ctypes-only ones).  bench.py does the same.  Without a GPU nothing is initialised."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session")
def krlib(torch_hip_runtime_first):
    """The HIP shared library behind the C ABI; session-wide.  Fails (not skips) when it cannot be loaded."""
    from raytrace_cpu_amd import capi
    return capi.load()


def pytest_sessionfinish(session, exitstatus):
    """GPU runs leave the measured parity margins behind (tests/parity.py::record_margin)."""
    try:
        import parity
        parity.dump_margins(os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    except Exception:
        pass
