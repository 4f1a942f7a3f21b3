#!/usr/bin/env python
"""The unmodified drop-in's loop (GaussianRasterizer(...)(**rv) -> backward, P = 8,280, 512x375) has two regimes on one box and one
build: 4.9-5.0 k and 8.4-8.7 k iterations/s (round 5: the SAME three source versions measured back to back gave 4.9 / 8.6 / 4.9 k,
unrelated to the version).  Its host work is 130 us per iteration (cProfile below: 52 ms per 400), its GPU work 85 us; in "auto"
mode the backward waits for ITS forward's binning status before it launches (no gradient of a truncated render may exist), so host
and device hand over to each other once per iteration and the loop runs at the sum of what does not overlap.
    python tools/dropin_regimes.py <label> [prof]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from diff_gaussian_rasterization import GaussianRasterizer as Renderer
from scaffold import reference_boundary as boundary, scene
dev = torch.device("cuda")
H, W = 512, 375
p = scene.make_gaussians(69, 120, opacity="A", seed=0)
params = {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}
cams = scene.camera_rig(H, W, n_views=24, device=dev)
g = torch.Generator().manual_seed(0)
dcs = [(torch.randn(3, H, W, generator=g) / (3 * H * W)).to(dev) for _ in range(24)]
rv_fixed = {k: v.detach().clone().requires_grad_(True) for k, v in boundary.params2rendervar(params).items()}
def it(i):
    im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv_fixed)
    im.backward(dcs[i % 24])
for i in range(100): it(i)
torch.cuda.synchronize()
runs = []
for _ in range(5):
    t0 = time.perf_counter()
    for i in range(400): it(i)
    torch.cuda.synchronize()
    runs.append(400 / (time.perf_counter() - t0))
print(sys.argv[1], [round(x) for x in runs])
if len(sys.argv) > 2:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(400): it(i)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
