"""
The per-frame multi-view optimisation loop of Topo4D (reference train.py:661-673, get_batch :105-112, the photometric
branch of get_loss :303-328) on top of the fused MI355X pieces:

    curr = get_batch(todo, dataset)                      random view, without replacement           (train.py:105-112)
    rv   = params2rendervar(params)                      activations                                 (helpers.py:91-100)
    im,… = Renderer(raster_settings=curr['cam'])(**rv)   t4d_rasterize_forward                       (train.py:307)
    loss = 0.8 L1 + 0.2 (1-SSIM) on exp(cam_m) im + cam_c   t4d_photometric_loss (or the torch ops)     (train.py:310,315)
    loss.backward(); optimizer.step(); freezes           t4d_rasterize_backward, t4d_adam_pin_step    (train.py:667-700)

Same schedule as the reference: one view per iteration, one Adam step per view.  Regularisers (train.py:330-368) are out of
scope; `extra_loss` lets a caller add them as plain torch.
"""
from __future__ import annotations

from random import Random
from typing import Callable, List, Optional

import torch

from . import loss as t4d_loss
from .boundary import params2rendervar, params2rendervar_fused
from .rasterizer import GaussianRasterizer


def get_batch(todo_dataset: list, dataset: list, rng: Random, idx: Optional[int] = None):
    """train.py:105-112: refill when empty, then pop a random entry (or peek at `idx`).  Returns (entry, todo)."""
    if not todo_dataset:
        todo_dataset = dataset.copy()
    if idx is None:
        curr = todo_dataset.pop(rng.randint(0, len(todo_dataset) - 1))
    else:
        curr = todo_dataset[idx]
    return curr, todo_dataset


def photometric_iteration(params, curr_data, fused_loss: bool = True, extra_loss: Optional[Callable] = None,
                          fused_activations: bool = True):
    """One forward of get_loss's photometric branch (train.py:303-328, use_mask False); returns (loss, radius)."""
    on_gpu = params['means3D'].is_cuda
    rendervar = params2rendervar_fused(params) if (fused_activations and on_gpu) else params2rendervar(params)
    rendervar['means2D'].retain_grad()
    im, radius, _, _ = GaussianRasterizer(raster_settings=curr_data['cam'])(**rendervar)
    cid = curr_data['id']
    if fused_loss:
        cm = params['cam_m'][cid][None] if 'cam_m' in params else None
        cc = params['cam_c'][cid][None] if 'cam_c' in params else None
        l = t4d_loss.photometric_loss(im[None], curr_data['im'][None], cm, cc)[0]
    else:
        cm = params['cam_m'][cid] if 'cam_m' in params else None
        cc = params['cam_c'][cid] if 'cam_c' in params else None
        l = t4d_loss.photometric_loss_torch(im, curr_data['im'], cm, cc)
    if extra_loss is not None:
        l = l + extra_loss(params, rendervar)
    return l, radius, rendervar


def optimise_views(params, dataset: List[dict], optimizer, n_iters: int, seed: int = 0, fused_loss: bool = True,
                   extra_loss: Optional[Callable] = None, max_2D_radius: Optional[torch.Tensor] = None):
    """train.py:661-673 for `n_iters` iterations.  Returns the list of per-iteration losses (device scalars, no sync)."""
    rng = Random(seed)
    todo: list = []
    losses = []
    for _ in range(n_iters):
        curr, todo = get_batch(todo, dataset, rng)
        l, radius, _ = photometric_iteration(params, curr, fused_loss, extra_loss)
        l.backward()
        with torch.no_grad():
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
            if max_2D_radius is not None:                      # train.py:373-375 bookkeeping
                seen = radius > 0
                max_2D_radius[seen] = torch.max(radius[seen], max_2D_radius[seen])
        losses.append(l.detach())
    return losses
