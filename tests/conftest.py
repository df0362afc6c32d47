import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def krlib():
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
