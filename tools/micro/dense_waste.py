#!/usr/bin/env python
"""Dense one-view scene: of the segments of the long tiles (>= 2,048 pairs), how many lie behind the last contributor of every pixel of
their tile (blended by launch 1 of the depth-parallel forward for nothing)?   GPU box."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scaffold import reference_boundary as boundary, scene
from tests import util
H, W = 3008, 4096
p = scene.make_gaussians(1000, 1000, opacity="A", seed=0)
rv = {k: v.detach() for k, v in boundary.params2rendervar(p).items()}
for view in (12, 4):
    cams = scene.camera_rig(H, W, n_views=24)[view:view + 1]
    out, _, batch = util.hip_render(cams, rv)
    st = util.decode_state(batch)
    tc = st["tile_count"][0]; nc = st["n_contrib"][0]
    gx, gy = W // 16, H // 16
    tmax = nc[:gy * 16, :gx * 16].reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    long_ = tc >= 2048
    nb = (tc[long_] + 127) // 128
    need = (tmax[long_] + 127) // 128
    print(f"view {view}: {int(long_.sum())} long tiles, {int(nb.sum())} segments, {int(need.sum())} reach a contributor ({need.sum() / nb.sum():.2f}), "
          f"tiles whose pixels all stop within the first 8 segments: {int((need <= 8).sum())}")
