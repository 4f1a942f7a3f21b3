#!/usr/bin/env python
"""Soak run of the round-5 kernels around the loss (GPU box): t4d_label_mask_target against the torch restatement of helpers.get_mask +
masked_gt (oracle/loss_oracle.py, pinned by golden G9: bit for bit), on random label images - 8-bit ones and ones pushed off the
grid to within float32 round-off of the `< 1` threshold - with 0..5 selected labels, 1..6 cameras, 1 x 1 .. 200 x 420 pixels; and
t4d_soft_color_loss against float64 (value) and the closed form (weight / rows) * sign(x - y) (gradient, bit for bit), rows from 1 to
3 million, with exact ties.     python tools/soak_targets.py [first_seed] [n]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loss_oracle
from scaffold import scene
from topo4d_amd import loss

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cmap = scene.parsing_colormap_bgr(14)
bad, masked_px, near = [], 0, 0
worst_soft = 0.0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    V, H, W = int(rng.integers(1, 7)), int(rng.integers(1, 201)), int(rng.integers(1, 421))
    labels = torch.randint(0, 14, (V, H, W), generator=g).numpy()
    img = cmap[labels].astype(np.int64)                                               # [V,H,W,3]
    img = np.clip(img + (rng.integers(-2, 3, size=img.shape) * (rng.random(img.shape) < 0.2)), 0, 255)
    mask = torch.tensor(img.astype(np.uint8) / 255.0).float().permute(0, 3, 1, 2).contiguous()
    if seed % 2:                                                                      # off the 8-bit grid, many values AT the threshold
        edge = torch.where(torch.rand(mask.shape, generator=g) < 0.5, 1.0, -1.0) * (1.0 + torch.randint(-4, 5, mask.shape, generator=g).float() * 1e-7)
        mask = mask + torch.where(torch.rand(mask.shape, generator=g) < 0.3, edge / 255.0, (torch.rand(mask.shape, generator=g) - 0.5) * (2.4 / 255))
    sel = rng.choice(14, size=int(rng.integers(0, 6)), replace=False)
    colors = torch.tensor(cmap[sel].reshape(-1, 3))
    gt = torch.rand(V, 3, H, W, generator=g)
    scale = float(rng.choice([0.1, 0.5, 0.0, 0.3333]))
    filt, target = loss.label_mask_target(mask.cuda(), colors, gt.cuda(), scale)
    for v in range(V):
        f_ref = loss_oracle.label_mask_torch(mask[v], colors)
        t_ref = loss_oracle.masked_target_torch(gt[v], f_ref, scale)
        if not (torch.equal(filt[v].cpu(), f_ref) and torch.equal(target[v].cpu(), t_ref)):
            bad.append(("mask", seed, v, V, H, W)); print("seed %d view %d: label mask / target differ from the reference's torch ops" % (seed, v), flush=True)
        masked_px += int(f_ref[0].sum())
        m255 = mask[v] * 255
        for c in colors:
            near += int(((torch.abs(m255 - c.reshape(3, 1, 1)) - 1).abs() < 1e-4).any(0).sum())
    # soft colour
    rows = int(rng.choice([1, 2, 255, 257, 1024, int(rng.integers(3, 300000)), int(rng.integers(300000, 3000000))]))
    width = int(rng.choice([1, 3, 3, 3, 4]))
    x = torch.randn(rows, width, generator=g)
    y = x + torch.randn(rows, width, generator=g) * 0.1
    k = int(rng.integers(2, 9))
    y[::k] = x[::k]                                                                   # exact ties: sign(0) = 0
    w = float(rng.choice([0.02, 1.0, 0.5]))
    l, gr = loss.soft_color_loss_raw(x.cuda(), y.cuda(), w)
    ref = (x.double() - y.double()).abs().sum(-1).mean()
    es = abs(float(l) - float(ref)) / max(float(ref), 1e-30)
    worst_soft = max(worst_soft, es)
    want = torch.sign(x - y) * (torch.tensor(w, dtype=torch.float32) / torch.tensor(float(rows), dtype=torch.float32))
    if es > 3e-6 or not torch.equal(gr.cpu(), want):
        bad.append(("soft", seed, rows, width)); print("seed %d: soft colour rows=%d width=%d FAILED (value err %.2e)" % (seed, rows, width, es), flush=True)
print("targets soak: %d batches (seeds %d..%d): label mask + masked target bit-identical to the reference's torch ops on every view "
      "(%d masked pixels, %d pixel-label pairs within 1e-4 of the `< 1` threshold); soft colour: largest relative value error %.2e, "
      "gradients bit-identical to the closed form; %d failures %s" % (n, first, first + n - 1, masked_px, near, worst_soft, len(bad), bad[:10]))
