#!/usr/bin/env python
"""Counting build (tools/ab_build.sh count -DT4D_COUNT; T4D_LIB=...lib_count.so) on the dense one-view scene: wave-steps of the
render walks per view and scene -> is a slow view more WORK or a longer TAIL?   GPU box."""
import ctypes as C, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, _lib, pack_views
dev = torch.device("cuda")
H, W = 3008, 4096
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
def run(p, tag, view):
    cams = scene.camera_rig(H, W, n_views=24, device=dev)[view:view + 1]
    rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
    dc = torch.randn(1, 3, H, W, device=dev) / (3 * H * W)
    b = ViewBatch(pack_views(cams, dev), H, W)
    f = lambda: (b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b.backward(dc))
    topo4d_amd.set_sync_mode("checked"); f(); st = b.fetch_status(); topo4d_amd.set_sync_mode("lazy")
    f(); torch.cuda.synchronize()
    lib.t4d_debug_read_counters(buf, 1)
    f(); torch.cuda.synchronize()
    lib.t4d_debug_read_counters(buf, 1)
    c = [int(x) for x in buf]
    out = {"scene": tag, "view": view, "pairs": int(st.total_pairs), "longest": int(st.max_tile_pairs)}
    for name, o in (("bwd", 0), ("fwd", 8)):
        tiles, batches, steps, visits, lanes = c[o:o + 5]
        out[name] = {"tiles": tiles, "wave_batches": batches, "wave_steps": steps, "row_visits": visits, "lane_steps": lanes,
                     "useful": round(lanes / max(1, 64 * steps), 3)}
    print(json.dumps(out))
full = scene.make_gaussians(1000, 1000, opacity="A", seed=0)
for v in (12, 4): run(full, "full", v)
big = scene.make_gaussians(1150, 1150, opacity="A", seed=0)
keep = big["means3D"][:, 1].abs() < scene.SEMI_AXES[1] * math.sin(math.radians(60))
cut = {k: v[keep].contiguous() for k, v in big.items()}
run(cut, "caps cut", 12)
