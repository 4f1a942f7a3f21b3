"""Per-kernel HIP-event times of ONE view per call (Topo4D's geometry-pass shape: P = 8,280, 512x375): the regime the reference's
per-view loop runs in.  usage: python tools/prof_single_view.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, _lib, pack_views
dev = torch.device("cuda"); H, W = 512, 375
p = scene.make_gaussians(69, 120, opacity="A", seed=0)
cams = scene.camera_rig(H, W, n_views=24, device=dev)
rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
dc = (torch.randn(1, 3, H, W) / (3 * H * W)).to(dev)
b = ViewBatch(pack_views(cams[:1], dev), H, W)
f = lambda: (b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b.backward(dc))
f(); topo4d_amd.set_sync_mode("lazy")
for _ in range(50): f()
torch.cuda.synchronize()
_lib.profile_begin()
for _ in range(200): f()
torch.cuda.synchronize()
prof = _lib.profile_end()
tot = 0.0
for name, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    if n == 0: continue                      # (one view of <= 1,024 tiles: the scan is part of the scatter launch)
    print(f"{name:24s} {1000 * ms / n:8.1f} us  x{n // 200}/iter"); tot += 1000 * ms / 200
print(f"sum of kernels {tot:.1f} us per iteration; pairs {b.fetch_status().total_pairs}")
