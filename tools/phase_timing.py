"""Experiment (GPU box, needs the timing build: tools/ab_build.sh timing -DT4D_TIMING; run with T4D_LIB=.../lib_timing.so):
s_memtime stamps of the phases of render_bwd's workgroup 0 (the longest tile) at the reference's one-view call shape."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, _lib, pack_views
dev = torch.device("cuda"); H, W = 512, 375
p = scene.make_gaussians(69, 120, opacity="A", seed=0)
cams = scene.camera_rig(H, W, n_views=24, device=dev)
rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
dc = (torch.randn(1, 3, H, W) / (3 * H * W)).to(dev)
b = ViewBatch(pack_views(cams[12:13], dev), H, W)
f = lambda: (b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b.backward(dc))
f(); topo4d_amd.set_sync_mode("lazy")
for _ in range(200): f()
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 512)()
lib.t4d_debug_read_timing(buf, 512)
t = np.array(buf[:], dtype=np.int64)
t0 = t[0]
print("n", t[2], "tile_max", t[3], "total cycles", t[4] - t0, "wall ticks (100 MHz?)", t[6] - t[5], "=> us", (t[6] - t[5]) / 100.0,
      "=> clock MHz", (t[4] - t0) / max((t[6] - t[5]) / 100.0, 1e-9))
print("prologue (item -> pixel state, sub-block max, barrier):", t[1] - t0)
nb = (int(t[2]) + 127) // 128
for i in range(nb):
    s = t[8 + 8 * i: 16 + 8 * i]
    print(f"batch {i}: stage+barrier {s[1]-s[0]:7d}  lists {s[2]-s[1]:7d}  replay {s[3]-s[2]:7d} ({s[5]} steps, {(s[3]-s[2])/max(s[5],1):.0f} cyc/step)  "
          f"barrier wait {s[4]-s[3]:7d}  -> next {(t[8+8*(i+1)] if i + 1 < nb else t[4]) - s[4]:7d}")
