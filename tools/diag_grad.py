"""Diagnostic (GPU box): worst gradient deviations HIP vs C oracle for one view of a config; dumps details to gpurun_out/."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util
from scaffold import scene
cfgname, opacity, view = sys.argv[1], sys.argv[2], int(sys.argv[3])
cfg = scene.CONFIGS[cfgname]
H, W, V = cfg["H"], cfg["W"], cfg["n_views"]
rv, cams = util.make_scene(cfg["n_lat"], cfg["n_lon"], H, W, V, opacity=opacity, sh_degree=cfg["sh_degree"], seed=0)
if cfg["sh_degree"] is not None and len(sys.argv) > 4:
    rv["shs"][::11, 0, :] = -3.0
dc, _, _ = scene.output_cotangents(V, H, W, seed=0)
out, g, batch = util.hip_render([cams[view]], rv, dc[view:view + 1])
st = util.decode_state(batch)
r, gref = util.c_oracle_render(cams[view], rv, dc[view])
res = {}
for k in ("means3D", "means2D", "opacities", "scales", "rotations", "shs", "colors_precomp"):
    if g.get(k) is None or k not in gref: continue
    a = g[k][0].astype(np.float64); b = gref[k].astype(np.float64).reshape(a.shape)
    err = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
    scale = np.abs(b).max()
    worst = np.argsort(err)[-5:][::-1]
    print(k, "rel", err.max() / scale, "scale", scale)
    for i in worst:
        print("   g", i, "err", err[i], "hip", a[i].ravel()[:4], "ref", b[i].ravel()[:4], "xy", st["xy"][0][i], "radius", out["radii"][0][i],
              "conic_op", st["conic_opacity"][0][i], "depth", st["depth"][0][i])
    res[k] = dict(worst=worst.tolist(), err=err[worst].tolist(), scale=float(scale))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"diag_{cfgname}_{opacity}_{view}.json"), "w"))
