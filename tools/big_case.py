"""One-off scale check (GPU box): Topo4D's texture-pass shape - ~1M Gaussians at ~4096x3008, one view per call - through the
C ABI: no integer overflow, binning invariants, image identities, finite and reproducible gradients."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, pack_views
import util
dev = torch.device("cuda")
H, W = 3008, 4096
n_lat, n_lon = 1000, 1000                                        # P = 1,000,000
p = scene.make_gaussians(n_lat, n_lon, opacity="A", seed=0)
cams = scene.camera_rig(H, W, n_views=24, device=dev)[:2]
rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
P = rv["means3D"].shape[0]
b = ViewBatch(pack_views(cams, dev), H, W)
t0 = time.perf_counter()
color, radii, depth, alpha = b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"])
torch.cuda.synchronize(); t1 = time.perf_counter()
st = b.fetch_status()
print(f"P={P} {W}x{H} V=2: pairs/view max {st.max_pairs_per_view}, total {st.total_pairs}, overflow {st.overflow}, "
      f"capacity {b.prob.pair_capacity}, state {b.state.numel() / 2**30:.2f} GiB, first forward {1e3 * (t1 - t0):.1f} ms")
assert st.overflow == 0
dc = torch.randn(2, 3, H, W, device=dev) / (3 * H * W)
g1 = b.backward(dc)
torch.cuda.synchronize()
for k, v in g1.items():
    if v is not None:
        assert torch.isfinite(v).all(), k
assert torch.isfinite(color).all() and torch.isfinite(depth).all()
s = util.decode_state(b)
assert int(s["tile_count"].sum()) == st.total_pairs
fT = torch.from_numpy(s["final_T"]).to(dev)
err = (alpha[:, 0] - (1 - fT)).abs().max().item()
print("max |alpha - (1 - final_T)|", err); assert err < 2e-5
# per-tile order on a sample of tiles: keys ascending
rng = np.random.default_rng(0)
T = s["T"]
for v in range(2):
    for t in rng.choice(T, 200, replace=False):
        n, off = int(s["tile_count"][v, t]), int(s["tile_off"][v, t])
        if n > 1:
            k = s["keys"][v, off:off + n]
            assert np.all(k[1:] > k[:-1]), (v, t)
tc = s["tile_count"].ravel()
print("longest tile list", int(tc.max()), "; tiles > 512:", int((tc > 512).sum()), " > 2048:", int((tc > 2048).sum()), " > 4096:", int((tc > 4096).sum()),
      "; keys in tiles > 2048:", int(tc[tc > 2048].sum()), "of", int(tc.sum()), "; non-empty tiles", int((tc > 0).sum()))
# determinism + timing
b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]); g2 = b.backward(dc)
for k in g1:
    if g1[k] is not None:
        assert torch.equal(g1[k], g2[k]), k
topo4d_amd.set_sync_mode("lazy")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]); b.backward(dc)
torch.cuda.synchronize()
print(f"fwd+bwd of 2 views: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms"); print("BIG CASE OK")
from topo4d_amd import _lib
_lib.profile_begin()
for _ in range(5):
    b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]); b.backward(dc)
torch.cuda.synchronize()
for name, (ms, n) in sorted(_lib.profile_end().items(), key=lambda kv: -kv[1][0]):
    print(f"  {name:22s} {1000 * ms / n:9.1f} us")
