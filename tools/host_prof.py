import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, pack_views, rasterize_views
from diff_gaussian_rasterization import GaussianRasterizer as Renderer
dev = torch.device("cuda"); H, W = 512, 375
p = scene.make_gaussians(69, 120, opacity="A", seed=0)
params = {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}
cams = scene.camera_rig(H, W, n_views=24, device=dev)
dc = (torch.randn(1, 3, H, W) / (3 * H * W)).to(dev)
rv = {k: v.detach() for k, v in boundary.params2rendervar(params).items()}
views = pack_views(cams[:1], dev)
b = ViewBatch(views, H, W)
def t(fn, n=500):
    for i in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn()
    te = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    return round(te / n * 1e6, 1), round(tt / n * 1e6, 1)
def a():
    b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]); b.backward(dc)
b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"])
for mode in ("checked", "lazy"):
    topo4d_amd.set_sync_mode(mode)
    print(mode, "ViewBatch fwd+bwd (enqueue us, total us):", t(a))
    rvg = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    def c():
        im, r, d, al = Renderer(raster_settings=cams[0])(**rvg); im.backward(dc[0])
    print(mode, "Renderer + autograd:", t(c))
    def d_():
        rr = boundary.params2rendervar(params)
        im, r, d, al = Renderer(raster_settings=cams[0])(**rr); im.backward(dc[0])
    print(mode, "params2rendervar + Renderer + autograd:", t(d_))
def e():
    rr = boundary.params2rendervar(params); l = sum(v.sum() for v in rr.values()); l.backward()
print("params2rendervar + trivial loss only:", t(e))
