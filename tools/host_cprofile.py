"""cProfile of the drop-in call path (one camera per call, Topo4D's geometry-pass shape) - where the host time goes."""
import cProfile, os, pstats, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from diff_gaussian_rasterization import GaussianRasterizer as Renderer
dev = torch.device("cuda"); H, W = 512, 375
p = scene.make_gaussians(69, 120, opacity="A", seed=0)
cams = scene.camera_rig(H, W, n_views=24, device=dev)
dc = (torch.randn(3, H, W) / (3 * H * W)).to(dev)
rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
rvg = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
def it(i):
    im, r, d, al = Renderer(raster_settings=cams[i % 24])(**rvg); im.backward(dc)
for i in range(50): it(i)
topo4d_amd.set_sync_mode(sys.argv[1] if len(sys.argv) > 1 else "auto")
for i in range(50): it(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(2000): it(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
