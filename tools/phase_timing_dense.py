"""Experiment (GPU box, timing build: tools/ab_build.sh timing -DT4D_TIMING; run with T4D_LIB=.../lib_timing.so): how long does
render_bwd's workgroup 0 (the longest tile) run compared with the whole kernel, at the dense-pass size (P = 1M, 4096x3008)?"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, _lib, pack_views
dev = torch.device("cuda"); H, W = 3008, 4096
p = scene.make_gaussians(1000, 1000, opacity="A", seed=0)
cams = scene.camera_rig(H, W, n_views=24, device=dev)
rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
dc = (torch.randn(1, 3, H, W) / (3 * H * W)).to(dev)
b = ViewBatch(pack_views(cams[9:10], dev), H, W)
f = lambda: (b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b.backward(dc))
f(); topo4d_amd.set_sync_mode("lazy")
for _ in range(5): f()
torch.cuda.synchronize()
_lib.profile_begin()
for _ in range(5): f()
torch.cuda.synchronize()
prof = _lib.profile_end()
lib = _lib.load()
buf = (C.c_ulonglong * 512)()
lib.t4d_debug_read_timing(buf, 512)
t = np.array(buf[:], dtype=np.int64)
print("longest tile n", t[2], "tile_max", t[3], "workgroup 0 lifetime: cycles", t[4] - t[0], "=> us", (t[6] - t[5]) / 100.0)
for name, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    if n: print(f"  {name:22s} {1000 * ms / n:9.1f} us")
