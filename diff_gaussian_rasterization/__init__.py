"""
Drop-in module name: Topo4D does `from diff_gaussian_rasterization import GaussianRasterizer,
GaussianRasterizationSettings` (reference train.py:19, helpers.py:18-19).  Putting this repository on
PYTHONPATH makes those imports resolve to the MI355X-native implementation.

One host-side setting rides along with the import.  Topo4D's loop (train.py:661-673) calls `loss.backward()` once per rendered
view, thousands of times per frame; torch hands every such call over to the autograd engine's per-device worker thread and
waits for it.  With 85 us of kernels per iteration that hand-over IS the iteration: the same build ran at 4-5 k or at 8-9 k
iterations/s depending on how quickly the sleeping worker woke up, flipping between the two within one process
(tools/dropin_clock_probe.py; profiles/r06_dropin_regimes.txt: the kernels last 83 us in both regimes, the host's time per
iteration is 182 us in one and 115 us in the other).  A single-GPU optimisation loop gains nothing from that thread, so this
module runs the backward on the calling thread (`torch.autograd.set_multithreading_enabled(False)`: 7.7-8.0 k iterations/s in
every block of every process).  T4D_AUTOGRAD_ENGINE_THREAD=1 in the environment leaves torch's default alone.
"""
import os as _os

import torch as _torch

from topo4d_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

if _os.environ.get("T4D_AUTOGRAD_ENGINE_THREAD", "0") != "1" and hasattr(_torch.autograd, "set_multithreading_enabled"):
    _torch.autograd.set_multithreading_enabled(False)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
