"""
Drop-in module name: Topo4D does `from diff_gaussian_rasterization import GaussianRasterizer,
GaussianRasterizationSettings` (reference train.py:19, helpers.py:18-19).  Putting this repository on
PYTHONPATH makes those imports resolve to the MI355X-native implementation.
"""
from topo4d_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
