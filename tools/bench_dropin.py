#!/usr/bin/env python
"""The reference's own call shape: one camera per call through the drop-in module (train.py:303-315,667): P = 8,280 Gaussians
(Topo4D's mesh), 512x375 image, params2rendervar -> Renderer(cam)(**rv) -> photometric loss -> backward.  Reports
iterations/s (= views/s at one view per iteration, the reference's schedule) for: the rasterizer alone with a supplied
dL/dcolor, and the full iteration with the torch loss and with the fused HIP loss.  One JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo4d_amd
from diff_gaussian_rasterization import GaussianRasterizer as Renderer
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import boundary as fused_boundary
from oracle import loss_oracle
from topo4d_amd import loss

dev = torch.device("cuda")
H, W = 512, 375
p = scene.make_gaussians(69, 120, opacity="A", seed=0)            # 8,280 vertex-bound Gaussians
params = {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}
cams = scene.camera_rig(H, W, n_views=24, device=dev)
g = torch.Generator().manual_seed(0)
gts = [torch.rand(3, H, W, generator=g).to(dev) for _ in range(24)]
dcs = [(torch.randn(3, H, W, generator=g) / (3 * H * W)).to(dev) for _ in range(24)]
from topo4d_amd.optim import FusedAdamPins
groups = [{"params": [v], "name": k, "lr": 1e-4} for k, v in params.items()]
opt = torch.optim.Adam(groups, eps=1e-15)
fopt = FusedAdamPins(groups, eps=1e-15)
fopt.set_pin("means3D", torch.arange(0, 8280, 5), params["means3D"][::5].detach().clone())      # a "static region"
static_idx = torch.arange(0, 8280, 5, device=dev); static_vals = params["means3D"][::5].detach().clone()

def it_raster(i):
    rv = boundary.params2rendervar(params)
    im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv)
    im.backward(dcs[i % 24])
def it_torch_loss(i):
    rv = boundary.params2rendervar(params)
    im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv)
    l = loss_oracle.photometric_loss_torch(im, gts[i % 24]); l.backward(); opt.step(); opt.zero_grad(set_to_none=True)
def it_fused_loss(i):
    rv = boundary.params2rendervar(params)
    im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv)
    l = loss.photometric_loss(im[None], gts[i % 24][None]).sum(); l.backward(); opt.step(); opt.zero_grad(set_to_none=True)

def it_fused_all(i):
    rv = boundary.params2rendervar(params)
    im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv)
    l = loss.photometric_loss(im[None], gts[i % 24][None]).sum(); l.backward(); fopt.step(); fopt.zero_grad(set_to_none=True)
def it_fused_all_act(i):
    rv = fused_boundary.params2rendervar_fused(params)
    im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv)
    l = loss.photometric_loss(im[None], gts[i % 24][None]).sum(); l.backward(); fopt.step(); fopt.zero_grad(set_to_none=True)
def it_torch_all(i):
    rv = boundary.params2rendervar(params)
    im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv)
    l = loss_oracle.photometric_loss_torch(im, gts[i % 24]); l.backward(); opt.step(); opt.zero_grad(set_to_none=True)
    with torch.no_grad():
        for _ in range(8):                                   # train.py:676-700: ~8-16 masked assignments per iteration
            params["means3D"][static_idx] = static_vals

out = {"workload": "1 view per call, P=8280, 512x375, opacity 1.0 (Topo4D geometry pass shape)"}

# the same full iteration recorded in one HIP graph per camera and replayed (loop.GraphedViews)
from topo4d_amd import loop as t4d_loop
gparams = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
gopt = FusedAdamPins([{"params": [v], "name": k, "lr": 1e-4} for k, v in gparams.items()], eps=1e-15, capturable=True)
gopt.set_pin("means3D", torch.arange(0, 8280, 5), gparams["means3D"][::5].detach().clone())
gdata = [{"cam": cams[i], "im": gts[i], "id": i} for i in range(24)]
gv = t4d_loop.GraphedViews(gparams, gdata, gopt)
for i in range(48): gv.step(i % 24)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(2000): gv.step(i % 24)
torch.cuda.synchronize()
out["iter_fused_everything_hip_graph_it_per_s"] = round(2000 / (time.perf_counter() - t0), 1)
gv.check()
for mode in ("checked", "auto", "lazy"):
    topo4d_amd.set_sync_mode("checked")
    for name, fn in (("raster_only", it_raster), ("iter_torch_loss", it_torch_loss), ("iter_fused_loss", it_fused_loss),
                     ("iter_torch_loss_adam_freezes", it_torch_all), ("iter_fused_loss_adam_pins", it_fused_all),
                     ("iter_fused_loss_adam_pins_activations", it_fused_all_act)):
        for i in range(30): fn(i)
        topo4d_amd.set_sync_mode(mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 300
        for i in range(n): fn(i)
        torch.cuda.synchronize()
        out[f"{name}_{mode}_it_per_s"] = round(n / (time.perf_counter() - t0), 1)
        topo4d_amd.set_sync_mode("checked")
print(json.dumps(out))
