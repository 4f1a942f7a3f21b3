#!/usr/bin/env python
"""Times the fused photometric loss (forward + gradient, 24 x 3 x 512 x 512) against the torch restatement of the
reference (5 conv2d + autograd per view).  Prints one JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo4d_amd import loss
V, H, W = 24, 512, 512
g = torch.Generator().manual_seed(0)
im = torch.rand(V, 3, H, W, generator=g).cuda().requires_grad_(True)
gt = torch.rand(V, 3, H, W, generator=g).cuda()
cm = (torch.randn(V, 3, generator=g) * 0.1).cuda().requires_grad_(True)
cc = (torch.randn(V, 3, generator=g) * 0.05).cuda().requires_grad_(True)
def fused():
    l = loss.photometric_loss(im, gt, cm, cc); l.sum().backward()
def ref():
    l = sum(loss.photometric_loss_torch(im[v], gt[v], cm[v], cc[v]) for v in range(V)); l.backward()
out = {}
for name, fn, reps in (("fused_hip", fused, 50), ("torch_per_view", ref, 5)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    out[name + "_ms_per_24_views"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
out["speedup"] = round(out["torch_per_view_ms_per_24_views"] / out["fused_hip_ms_per_24_views"], 1)
print(json.dumps(out))
