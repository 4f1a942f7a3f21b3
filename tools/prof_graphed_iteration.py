#!/usr/bin/env python
"""Runs ON THE GPU BOX under rocprofv3 (PROF_CMD of tools/prof.sh): the whole optimisation iteration of ONE view (train.py:661-700:
activations -> render -> fused photometric loss -> backward -> fused Adam + pins; P = 8,280, 512x375) replayed from one HIP graph
per camera - bench.py's full_iteration.v1_graphed.  The kernel stats of this run are the per-iteration launch list."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scaffold import scene
from topo4d_amd import loop as t4d_loop
from topo4d_amd.optim import FusedAdamPins
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
H, W = 512, 375
p = scene.make_gaussians(69, 120, opacity="A", seed=0)
params = {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}
P = params["means3D"].shape[0]
opt = FusedAdamPins([{"params": [v], "name": k, "lr": 1e-5} for k, v in params.items()], eps=1e-15, capturable=True)
opt.set_pin("means3D", torch.arange(0, P, 5), params["means3D"][::5].detach().clone())
cams = scene.camera_rig(H, W, n_views=24, device=dev)
data = [{"cam": cams[i], "im": torch.rand(3, H, W, generator=g).to(dev), "id": i} for i in range(24)]
gv = t4d_loop.GraphedViews(params, data, opt)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 480
for i in range(48):
    gv.step(i % 24)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    gv.step(i % 24)
torch.cuda.synchronize()
print("graphed iterations/s %.1f  (%.1f us per iteration, %d replays)" % (n / (time.perf_counter() - t0), 1e6 * (time.perf_counter() - t0) / n, n))
gv.check()
from topo4d_amd import rasterizer as R
print("longest tile list over the cameras (warm-up, checked):", {k: sc.longest_bin for k, sc in R._SCENES.items()})
