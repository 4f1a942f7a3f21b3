import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["throughput", "latency"])
def render_build(request, monkeypatch):
    """Runs a test once with each build of the per-tile render kernels (csrc/t4d_raster.hip: LAT = false / true), whatever
    the launch size would have picked."""
    monkeypatch.setenv("T4D_LATENCY_TILES", "0" if request.param == "throughput" else "1000000000")
    return request.param
