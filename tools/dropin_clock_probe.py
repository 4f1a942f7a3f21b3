#!/usr/bin/env python
"""Why does the unmodified drop-in's loop (one 512x375 view of 8,280 Gaussians per iteration through autograd: 85 us of kernels, ~130 us of
host work) run at 4-5 k OR 8-9 k iterations/s, flipping between blocks of one process?  Per block of 400 iterations this prints the
rate, the host time spent inside the forward / backward calls, and - from a following block of 100 iterations under the library's
HIP-event timer - what the SAME kernels lasted on the device, plus the current shader clock where sysfs shows it."""
import glob, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from diff_gaussian_rasterization import GaussianRasterizer as Renderer
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import _lib
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
def sclk():
    out = []
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            out += [l.strip() for l in open(f) if "*" in l]
        except OSError:
            pass
    return out[:1]
# (importing the drop-in module puts the backward on the calling thread; T4D_AUTOGRAD_ENGINE_THREAD=1 keeps torch's engine thread)
print("autograd multithreading:", torch.autograd.is_multithreading_enabled())
for i in range(100): it(i)
torch.cuda.synchronize()
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t0 = time.perf_counter()
    for i in range(400): it(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    clk = sclk()
    _lib.profile_begin()
    for i in range(100): it(i)
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    k_us = sum(1e3 * ms for ms, n in prof.values()) / 100
    print(f"block {blk}: {400 / dt:7.0f} it/s  host enqueue {1e6 * t_enq / 400:6.1f} us/it  wall {1e6 * dt / 400:6.1f} us/it  kernels (next 100 it) {k_us:6.1f} us/it  sclk {clk}")
