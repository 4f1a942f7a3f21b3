#!/usr/bin/env python
"""BASELINE config 5: 8192^2 texture bake of a ~10^6-vertex UV mesh, GPU (t4d_texture_bake) vs the reference's CPU
rasterizer (oracle/_ref = /root/reference's mesh_core.cpp compiled as is; falls back to the C port when absent).
Prints one JSON line.   python tools/bench_bake.py [--res 8192] [--n 1025] [--reps 10] [--no-cpu]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import texture_oracle as TX
from scaffold.scene import uv_mesh
from topo4d_amd import texture

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=8192)
ap.add_argument("--n", type=int, default=1025)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()
verts, tris, colors = uv_mesh(a.n, a.res, a.res, seed=0)
tris = np.sort(tris.view([("a", np.int32), ("b", np.int32), ("c", np.int32)]), order=["a"], axis=0).view(np.int32)   # mesh order
dev = torch.device("cuda")
v, t, c = (torch.as_tensor(x).to(dev) for x in (verts, tris, colors))
img = texture.render_colors(v, t, c, a.res, a.res)          # warm-up + capacity
torch.cuda.synchronize()
runs = []
for _ in range(5):                                           # as bench.py's bake_probe: min of 5 runs of `reps` bakes (the first runs of a process warm the clocks)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        img = texture.render_colors(v, t, c, a.res, a.res)
    torch.cuda.synchronize()
    runs.append((time.perf_counter() - t0) / a.reps)
gpu_s = min(runs)
out = {"metric": "texture bake texels/s", "res": a.res, "triangles": int(tris.shape[0]), "vertices": int(verts.shape[0]),
       "gpu_ms": round(gpu_s * 1e3, 3), "gpu_texels_per_s": round(a.res * a.res / gpu_s, 1),
       "includes": "output/depth buffer allocation+fill, binning (count/scan/fill), render; inputs resident in HBM"}
if not a.no_cpu:
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        ref = TX.render_colors_cpu(verts, tris, colors, a.res, a.res)
        best = min(best, time.perf_counter() - t0)
    out["cpu_baseline"] = {"value": round(a.res * a.res / best, 1), "unit": "texels/s", "seconds": round(best, 3), "cores": 1,
                           "kind": "reference" if TX.have_ref() else "port",
                           "sample": "the same full mesh, min of 2 (the reference code is single-threaded)"}
    out["bit_identical_to_cpu"] = bool(np.array_equal(img.cpu().numpy(), ref))
    out["speedup"] = round(best / gpu_s, 1)
print(json.dumps(out))
