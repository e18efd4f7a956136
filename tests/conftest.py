import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the bench.py runs of the suite skip the live PMC collection of roofline.traffic (two more child runs under rocprofv3, about a minute each
# time); tests/test_gpu_bench_sharded.py::test_live_traffic... switches it back on for its own run
os.environ.setdefault("PQT_BENCH_NO_LIVE_TRAFFIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: the HIP path is the product.
    # Plain runs (no -m) on a CPU-only box skip the gpu tests.
    if _has_gpu():
        return
    mexpr = config.getoption("-m") or ""
    if "gpu" in mexpr and "not gpu" not in mexpr:
        return
    skip = pytest.mark.skip(reason="no GPU here; run with -m gpu on the GPU box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
