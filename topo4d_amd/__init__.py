"""
topo4d_amd — MI355X-native differentiable Gaussian-splatting rasterizer for Topo4D's render hot path.

Scope: the `GaussianRasterizer` / `GaussianRasterizationSettings` surface Topo4D calls at
reference train.py:307,388,463,484, implemented as hand-written HIP kernels for gfx950 behind the C ABI of
include/topo4d_raster.h, plus the view-sharded multi-GPU driver.  Nothing else of Topo4D is rebuilt.
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, ViewBatch, get_sync_mode,
                         pack_views, poll_truncation, rasterize_views, set_sync_mode)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "ViewBatch", "rasterize_views", "pack_views",
           "set_sync_mode", "get_sync_mode", "poll_truncation"]
