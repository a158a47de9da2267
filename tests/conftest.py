import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    # the CPU oracle runs inside GPU tests too; several xdist workers x all host cores oversubscribe badly
    try:
        import torch
        n = os.cpu_count() or 8
        torch.set_num_threads(max(1, min(8, n // 4 if os.environ.get("PYTEST_XDIST_WORKER") else n)))
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["bf16x3", "fp32"])
def gemm_precision(request):
    """run a GPU test under both GEMM arithmetic modes (default bf16x3; fp32 = fp32-input MFMA)"""
    from mdvit_amd import ops
    prev = ops.gemm_precision()
    ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision(prev)
