"""One-off randomized parity stress (GPU box): HIP path vs the C oracle over many seeds, splat sizes, anisotropy and
ragged image sizes.  Same checks and tolerances as tests/test_gpu_parity.py."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util
from tests.test_gpu_parity import check_outputs, check_grads
from topo4d_amd import scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(123)
for trial in range(n):
    H, W = int(rng.integers(40, 150)), int(rng.integers(40, 150))
    V = 2
    opacity = "AB"[trial % 2]
    rv, cams = util.make_scene(int(rng.integers(8, 30)), int(rng.integers(10, 40)), H, W, V, opacity=opacity, seed=100 + trial)
    sc = float(rng.choice([0.3, 0.7, 1.0, 2.0, 4.0, 8.0]))
    aniso = torch.tensor(rng.uniform(0.3, 3.0, size=(rv["scales"].shape[0], 3)), dtype=torch.float32)
    rv["scales"] = rv["scales"] * sc * aniso
    q = torch.tensor(rng.normal(size=(rv["rotations"].shape[0], 4)), dtype=torch.float32)
    rv["rotations"] = torch.nn.functional.normalize(q)
    dc, dd, da = scene.output_cotangents(V, H, W, seed=trial, depth_alpha=True)
    use_da = trial % 3 != 0
    hip, hg, batch = util.hip_render(cams, rv, dc, dd if use_da else None, da if use_da else None)
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v], dd[v] if use_da else None, da[v] if use_da else None)
        np.testing.assert_array_equal(hip["radii"][v], r.radii)
        ncd = (st["n_contrib"][v] != r.state()["n_contrib"]).mean()
        assert ncd <= 2e-4, f"n_contrib differs on {ncd:.2e} of the pixels"
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v)
    print(f"trial {trial}: {H}x{W} P={rv['means3D'].shape[0]} scale x{sc} opacity {opacity} DA={use_da} pairs/view={int(st['view_total'].mean())} max tile={int(st['tile_count'].max())} ok", flush=True)
print("STRESS OK")
