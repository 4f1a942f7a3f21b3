"""GPU box, under tools/prof.sh (PROF_CMD): the workload `bench.py`'s `dense_1m` times - Topo4D's texture pass shape, ONE 4096 x 3008
view of 10^6 Gaussians per call (train.py:729-741) - forward + backward, from the rig's camera 12 (one tile list of 12,614 pairs) and
camera 4 (a polar cap: hundreds of lists of 2,000 - 9,000).  The launches of its long tiles (k_sort_long_chunks, k_sort_long,
k_fwd_long_seg<0|1>, k_fwd_long_prefix, k_render_bwd<.., LONG>) only exist in this call shape.
    PROF_CMD="python tools/prof_dense_1v.py" tools/prof.sh r06_dense_1v        (cameras: argv, default "12 4")"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import ViewBatch, pack_views

dev = torch.device("cuda")
H, W = 3008, 4096
cams_idx = [int(a) for a in sys.argv[1:]] or [12, 4]
p = scene.make_gaussians(1000, 1000, opacity="A", seed=0)
rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
dc = torch.randn(1, 3, H, W, device=dev) / (3 * H * W)
rig = scene.camera_rig(H, W, n_views=24, device=dev)
for ci in cams_idx:
    b = ViewBatch(pack_views(rig[ci:ci + 1], dev), H, W)
    f = lambda: (b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b.backward(dc))
    topo4d_amd.set_sync_mode("checked")
    f()
    st = b.fetch_status()
    topo4d_amd.set_sync_mode("lazy")
    for _ in range(8):
        f()
    torch.cuda.synchronize()
    print(f"camera {ci}: pairs {st.total_pairs}, longest tile list {st.max_tile_pairs}")
    del b
