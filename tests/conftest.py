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


@pytest.fixture(params=["throughput", "latency", "segments"])
def render_build(request, monkeypatch):
    """Runs a test once with each way the per-tile render kernels can run (csrc/t4d_raster.hip), whatever the launch size would
    have picked: "throughput" = LAT false, whole tiles in the backward (what a 24-view launch runs); "latency" = LAT true with
    the depth-segmented backward (what a one-view call runs); "segments" = LAT false with the segmented backward (2-4 views)."""
    monkeypatch.setenv("T4D_LATENCY_TILES", "1000000000" if request.param == "latency" else "0")
    if request.param == "throughput":
        monkeypatch.setenv("T4D_NO_SEGMENTS", "1")
    else:
        monkeypatch.delenv("T4D_NO_SEGMENTS", raising=False)
    return request.param
