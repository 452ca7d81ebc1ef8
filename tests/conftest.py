import os
import sys

import pytest

# the CPU oracle's OpenMP Gram: at most 16 threads (a GPU box shows hundreds of hardware threads; its share per GPU is 16)
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import bnr_amd
        return bnr_amd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _have_gpu():
        pytest.fail("no GPU / libbnr_hip.so not usable: the HIP path must run, there is no fallback")
    return 0
