#!/usr/bin/env python
"""Soak run of the fused photometric loss (GPU box): random batches - 1 to 6 views, images from 1 x 1 to 200 x 420, random camera
affines and view weights - against the float64 torch restatement of the reference (topo4d_amd.loss_oracle.photometric_loss_torch, pinned
by golden G3), under every kernel a launch can take (T4D_PH_TILE 1 / 32 / 0 = tile kernel in both shapes / strips), and dL/dim
compared bit for bit between the kernels.     python tools/soak_loss.py [first_seed] [n]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loss_oracle
from topo4d_amd import loss

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, worst_l, worst_g = [], 0.0, 0.0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    V, H, W = int(rng.integers(1, 7)), int(rng.integers(1, 201)), int(rng.integers(1, 421))
    g = torch.Generator().manual_seed(seed)
    im = torch.rand(V, 3, H, W, generator=g)
    gt = (im + torch.randn(V, 3, H, W, generator=g) * float(rng.uniform(0.01, 0.3))).clamp(0, 1)
    cm, cc = torch.randn(V, 3, generator=g) * 0.1, torch.randn(V, 3, generator=g) * 0.05
    b = [t.double().requires_grad_(True) for t in (im, cm, cc)]
    lref = torch.stack([loss_oracle.photometric_loss_torch(b[0][v], gt[v].double(), b[1][v], b[2][v]) for v in range(V)])
    lref.sum().backward()
    l1_step = 2 * 0.8 / (3 * H * W)
    outs = []
    for tile in ("1", "32", "0"):
        os.environ["T4D_PH_TILE"] = tile
        l, d_im, d_m, d_c = loss.photometric_loss_raw(im.cuda(), gt.cuda(), cm.cuda(), cc.cuda())
        outs.append(d_im.cpu())
        el = float((l.double().cpu() - lref.detach()).abs().max())
        err = (d_im.double().cpu() - b[0].grad).abs()
        tol = 5e-5 * float(b[0].grad.abs().max()) + 1e-12
        eg = float(err.max()) / max(float(b[0].grad.abs().max()), 1e-30)
        ok = el <= 3e-6 and float(err.max()) <= tol + l1_step * 1.01 and float((err > tol).float().mean()) <= 1e-3
        for x, y in ((d_m, b[1].grad), (d_c, b[2].grad)):
            ok = ok and float((x.double().cpu() - y).abs().max()) <= 5e-5 * float(y.abs().max()) + 1e-12 + l1_step * H * W
        worst_l = max(worst_l, el)
        if float(err.max()) <= tol:
            worst_g = max(worst_g, eg)
        if not ok:
            bad.append((seed, tile, V, H, W)); print("seed %d tile=%s %dx%dx%d FAILED: loss err %.2e, grad err %.2e of max" % (seed, tile, V, H, W, el, eg), flush=True)
    if not (torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])):
        bad.append((seed, "kernels differ", V, H, W)); print("seed %d: dL/dim differs between the kernels" % seed, flush=True)
os.environ.pop("T4D_PH_TILE", None)
print("loss soak: %d batches x 3 kernels (seeds %d..%d), largest loss error %.2e, largest gradient error %.2e of the gradient's maximum "
      "(where no x' == gt tie is involved), %d failures %s" % (n, first, first + n - 1, worst_l, worst_g, len(bad), bad[:10]))
