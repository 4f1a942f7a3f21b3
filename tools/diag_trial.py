"""Diagnostic (GPU box): one trial of the randomised parity scene (tests/test_gpu_configs.py::_randomised_trial) - for every
gradient tensor the worst Gaussian of HIP vs the fp32 C oracle, with the float64 autograd oracle beside both.
    python tools/diag_trial.py <trial> [view]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util
from scaffold import scene
trial = int(sys.argv[1]); v = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(1000 + trial)
H, W = int(rng.integers(40, 150)), int(rng.integers(40, 150)); V = 2
opacity = "AB"[trial % 2]
rv, cams = util.make_scene(int(rng.integers(8, 30)), int(rng.integers(10, 40)), H, W, V, opacity=opacity, seed=100 + trial)
sc = float(rng.choice([0.3, 0.7, 1.0, 2.0, 4.0, 8.0]))
aniso = torch.tensor(rng.uniform(0.3, 3.0, size=(rv["scales"].shape[0], 3)), dtype=torch.float32)
rv["scales"] = rv["scales"] * sc * aniso
rv["rotations"] = torch.nn.functional.normalize(torch.tensor(rng.normal(size=(rv["rotations"].shape[0], 4)), dtype=torch.float32))
dc, dd, da = scene.output_cotangents(V, H, W, seed=trial, depth_alpha=True)
use_da = trial % 3 != 0
print(f"trial {trial}: {H}x{W}, P={rv['means3D'].shape[0]}, opacity {opacity}, scale x{sc}, depth/alpha cotangents {use_da}")
for tiles in ("0", "1000000000"):
    os.environ["T4D_LATENCY_TILES"] = tiles
    hip, hg, batch = util.hip_render(cams, rv, dc, dd if use_da else None, da if use_da else None)
    r, g = util.c_oracle_render(cams[v], rv, dc[v], dd[v] if use_da else None, da[v] if use_da else None)
    _, g64 = util.torch_oracle_render(cams[v], rv, dc[v], dd[v] if use_da else None, da[v] if use_da else None)
    print("build", "throughput" if tiles == "0" else "latency")
    for k in util.GRAD_KEYS:
        a = hg[k][v].astype(np.float64); b = np.asarray(g[k], np.float64).reshape(a.shape); c = g64[k].numpy().astype(np.float64).reshape(a.shape)
        e = np.abs(a - b).reshape(a.shape[0], -1).max(1); i = int(e.argmax()); s = np.abs(b).max()
        print(f"  {k:15s} max|.|={s:.3e}  worst Gaussian {i}: |hip-C|={e[i]:.3e} ({e[i]/s:.2e} of max)  |hip-f64|={np.abs(a[i]-c[i]).max():.3e}  |C-f64|={np.abs(b[i]-c[i]).max():.3e}")
