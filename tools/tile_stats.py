"""Tile-list statistics of one forward (GPU): how long the per-tile lists are, how deep pixels actually read them.
usage: python tools/tile_stats.py [C2|C4]"""
import json, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, pack_views
import util

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = dict(scene.CONFIGS[name])
dev = torch.device("cuda:0")
H, W, V = cfg["H"], cfg["W"], min(cfg["n_views"], 4)
params = scene.make_gaussians(cfg["n_lat"], cfg["n_lon"], opacity="A", sh_degree=cfg["sh_degree"], seed=0)
cams = scene.camera_rig(H, W, n_views=cfg["n_views"], device=dev, true_campos=cfg["sh_degree"] is not None)
if cfg["sh_degree"] is not None:
    cams = [c._replace(sh_degree=cfg["sh_degree"]) for c in cams]
cams = cams[::cfg["n_views"] // V][:V]
views = pack_views(cams, dev)
rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(params).items()}
if cfg["sh_degree"] is not None:
    rv["shs"] = params["shs"].to(dev); rv.pop("colors_precomp")
b = ViewBatch(views, H, W, 1.0, cfg["sh_degree"] or 0)
b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv.get("colors_precomp"), rv.get("shs"))
torch.cuda.synchronize()
st = util.decode_state(b)
tc = st["tile_count"].astype(np.int64)
nc = st["n_contrib"].astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
ncpad = np.zeros((V, gy * 16, gx * 16), np.int64); ncpad[:, :H, :W] = nc
tile_max = ncpad.reshape(V, gy, 16, gx, 16).max(axis=(2, 4)).reshape(V, -1)
blk_max = ncpad.reshape(V, gy * 2, 8, gx * 2, 8).max(axis=(2, 4))
out = dict(config=name, views=V, tiles_per_view=int(tc.shape[1]), pairs_per_view=float(tc.sum() / V),
           empty_tile_frac=float((tc == 0).mean()), mean_len_nonempty=float(tc[tc > 0].mean()),
           len_percentiles={p: float(np.percentile(tc[tc > 0], p)) for p in (10, 50, 90, 99, 100)},
           mean_tile_max_contrib=float(tile_max[tc > 0].mean()), consumed_frac=float(tile_max.sum() / tc.sum()),
           mean_pixel_n_contrib=float(nc[nc > 0].mean()) if (nc > 0).any() else 0.0,
           covered_pixel_frac=float((nc > 0).mean()),
           blocks_nonempty_frac=float((blk_max > 0).mean()),
           radii_mean=float(b.radii[b.radii > 0].float().mean()), radii_p90=float(b.radii[b.radii > 0].float().quantile(0.9)),
           visible_frac=float((b.radii > 0).float().mean()))
print(json.dumps(out))
