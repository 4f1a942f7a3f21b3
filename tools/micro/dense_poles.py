#!/usr/bin/env python
"""How much of dense_1m's time is the lat-long grid's poles (thousands of tiny splats per tile there)?  The same probe on the
full grid and on a grid with the polar caps cut off, P ~ 10^6 both.  GPU box."""
import json, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, _lib, pack_views
dev = torch.device("cuda")
H, W = 3008, 4096
def run(p, tag, view=12):
    cams = scene.camera_rig(H, W, n_views=24, device=dev)[view:view + 1]
    rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
    dc = torch.randn(1, 3, H, W, device=dev) / (3 * H * W)
    b = ViewBatch(pack_views(cams, dev), H, W)
    f = lambda: (b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b.backward(dc))
    topo4d_amd.set_sync_mode("checked"); f(); st = b.fetch_status(); topo4d_amd.set_sync_mode("lazy")
    for _ in range(3): f()
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(10): f()
    torch.cuda.synchronize()
    kern = {n: round(1e3 * ms / c, 1) for n, (ms, c) in _lib.profile_end().items() if c}
    print(json.dumps({"scene": tag, "P": int(rv["means3D"].shape[0]), "view": view, "pairs": int(st.total_pairs), "longest": int(st.max_tile_pairs),
                      "kernels_us": kern, "sum_us": round(sum(kern.values()), 1)}))
full = scene.make_gaussians(1000, 1000, opacity="A", seed=0)
for v in (12, 4):
    run(full, "full lat-long grid", v)
big = scene.make_gaussians(1150, 1150, opacity="A", seed=0)
keep = big["means3D"][:, 1].abs() < scene.SEMI_AXES[1] * math.sin(math.radians(60))
cut = {k: v[keep].contiguous() for k, v in big.items()}
for v in (12, 4):
    run(cut, "polar caps beyond 60 deg cut off", v)
