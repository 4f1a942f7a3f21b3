"""
The per-frame multi-view optimisation loops of Topo4D on top of the fused MI355X pieces - BOTH hot loops of train.py and the
loss branches train.py actually executes (train.py:631-632 hard-code use_mask = True, use_mask_dense = False):

  geometry loop (train.py:661-673; get_loss :300-328)                          this module
    curr = get_batch(todo, dataset)                 random view, no replacement     get_batch                       (train.py:105-112)
    rv   = params2rendervar(params)                 activations                     inside the rasterizer (T4D_FLAG_RAW_PARAMS) / t4d_activate_*
    im,… = Renderer(raster_settings=cam)(**rv)      render                          t4d_rasterize_forward           (train.py:307)
    im   = exp(cam_m[id]) im + cam_c[id]            camera affine       \\
    first frame : target = gt                                            |          t4d_photometric_loss            (train.py:310,318)
    later frames: target = gt, x 0.1 where the parsing mask says         |          t4d_label_mask_target, ONCE per (frame, camera)
                  "inner_mouth" (helpers.get_mask)                       |          (prepare_masked_targets)        (train.py:320-327)
    loss = 0.8 L1 + 0.2 (1 - SSIM)                                      /
    loss.backward(); optimizer.step(); freezes                                      t4d_rasterize_backward, t4d_adam_pin_step (train.py:667-700)

  texture loop (train.py:729-741; get_loss_dense :380-417 with use_mask False)
    dense_rgb_colors[static | dynamic | mouth_inner] = 0     pins BEFORE the render  FusedAdamPins.apply_pins       (train.py:731-734)
    rv   = params2rendervar_dense(params)           the dense_* parameters          as above, names prefixed 'dense_'
    loss = 0.8 L1 + 0.2 (1 - SSIM) on the render itself (NO affine)                 t4d_photometric_loss            (train.py:393)
         + 0.02 * l1_loss_v2(dense_rgb_colors, dense_init_colors)                   t4d_soft_color_loss             (train.py:407,541-543)
    loss.backward(); optimizer.step()                                               t4d_rasterize_backward, t4d_adam_pin_step (train.py:738-741)

Same schedule as the reference: one view per iteration, one Adam step per view.  The topology regularisers (train.py:330-368) are
out of scope; `extra_loss` lets a caller add them as plain torch.  This package has no torch loss of its own:
`loss_fn` takes the caller's (Topo4D's own l1_loss_v1 / calc_ssim, or the checker in oracle/loss_oracle.py) when the fused
kernel is not wanted.
"""
from __future__ import annotations

import warnings
from random import Random
from typing import Callable, List, Optional

import torch

from . import loss as t4d_loss
from .boundary import activate_backward, activate_forward, params2rendervar_fused
from .rasterizer import GaussianRasterizer

SOFT_COLOR_WEIGHT = 0.02              # losses_weights_dense['soft_color'], train.py:541-543


def get_batch(todo_dataset: list, dataset: list, rng: Random, idx: Optional[int] = None):
    """train.py:105-112: refill when empty, then pop a random entry (or peek at `idx`).  Returns (entry, todo)."""
    if not todo_dataset:
        todo_dataset = dataset.copy()
    if idx is None:
        curr = todo_dataset.pop(rng.randint(0, len(todo_dataset) - 1))
    else:
        curr = todo_dataset[idx]
    return curr, todo_dataset


# ------------------------------------------------------------------------------------------------------------
# targets: the image an iteration is compared with (train.py:315-327)
# ------------------------------------------------------------------------------------------------------------
def prepare_masked_targets(dataset: List[dict], label_colors, scale: float = t4d_loss.MASK_SCALE) -> None:
    """The later-frame target of every camera of a frame in ONE launch: entry['masked_im'] = entry['im'] with the elements
    helpers.get_mask(["inner_mouth"], entry['mask'], ...) selects multiplied by 0.1 (train.py:320-326; `label_colors` = the
    colours of the selected labels, helpers.py:806 `cmap[cmap_index["inner_mouth"]]`).  The reference rebuilds this in every
    one of the 1,100 iterations of a frame; it depends on (frame, camera) only.  Call it once after get_dataset."""
    todo = [e for e in dataset if 'masked_im' not in e]
    if not todo:
        return
    if any(e.get('mask') is None for e in todo):
        raise ValueError("prepare_masked_targets: every dataset entry needs its 'mask' image (get_dataset(..., use_mask=True), train.py:84-92)")
    shapes = {tuple(e['im'].shape) for e in todo}
    if len(shapes) == 1:
        _, target = t4d_loss.label_mask_target(torch.stack([e['mask'] for e in todo]), label_colors,
                                               torch.stack([e['im'] for e in todo]), scale, want_mask=False)
        for e, t in zip(todo, target.unbind(0)):
            e['masked_im'] = t
    else:                                                          # cameras of different sizes: one launch each
        for e in todo:
            e['masked_im'] = t4d_loss.label_mask_target(e['mask'], label_colors, e['im'], scale, want_mask=False)[1]


def target_image(curr_data: dict, use_mask: bool = False, is_initial_timestep: bool = True, label_colors=None) -> torch.Tensor:
    """What get_loss compares the render with (train.py:315-327): the camera's image, except with use_mask in the frames after the
    first, where it is the masked target (computed here and kept in the entry if prepare_masked_targets has not run)."""
    if not use_mask or is_initial_timestep:
        return curr_data['im']
    if 'masked_im' not in curr_data:
        if label_colors is None:
            raise ValueError("use_mask in a later frame needs prepare_masked_targets(dataset, label_colors) or label_colors here")
        prepare_masked_targets([curr_data], label_colors)
    return curr_data['masked_im']


def _names(dense: bool):
    p = 'dense_' if dense else ''
    return {k: p + k for k in ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales')}


_RENDER_KEYS = tuple(_names(False).values())
_DENSE_KEYS = tuple(_names(True).values())


def _dense_view(params):
    """The dense_* parameters under the names params2rendervar reads (helpers.py:102-112 is :91-100 on the dense set)."""
    return {k: params[v] for k, v in _names(True).items()}


# ------------------------------------------------------------------------------------------------------------
# one iteration through autograd
# ------------------------------------------------------------------------------------------------------------
def photometric_iteration(params, curr_data, loss_fn: Optional[Callable] = None, extra_loss: Optional[Callable] = None,
                          params2rendervar: Optional[Callable] = None, use_mask: bool = False, is_initial_timestep: bool = True,
                          label_colors=None):
    """One forward of get_loss's photometric part (train.py:303-327); returns (loss, radius, rendervar).
    `params2rendervar`: the function that turns the optimiser's parameters into rasterizer kwargs - Topo4D's own
    helpers.params2rendervar (helpers.py:91-100) may be passed; the default is its fused equivalent (GPU only).
    `loss_fn(im, target, cam_m_row, cam_c_row) -> scalar`: the caller's loss instead of the fused kernel."""
    rendervar = (params2rendervar or params2rendervar_fused)(params)
    rendervar['means2D'].retain_grad()
    im, radius, _, _ = GaussianRasterizer(raster_settings=curr_data['cam'])(**rendervar)
    cid = curr_data['id']
    target = target_image(curr_data, use_mask, is_initial_timestep, label_colors)
    cm = params['cam_m'][cid] if 'cam_m' in params else None
    cc = params['cam_c'][cid] if 'cam_c' in params else None
    if loss_fn is None:
        one = lambda t: None if t is None else t[None]
        l = t4d_loss.photometric_loss(im[None], target[None], one(cm), one(cc))[0]
    else:
        l = loss_fn(im, target, cm, cc)
    if extra_loss is not None:
        l = l + extra_loss(params, rendervar)
    return l, radius, rendervar


def dense_iteration(params, variables, curr_data, loss_fn: Optional[Callable] = None, soft_color_fn: Optional[Callable] = None,
                    soft_color_weight: float = SOFT_COLOR_WEIGHT, params2rendervar_dense: Optional[Callable] = None):
    """One forward of get_loss_dense as train.py:735 calls it (use_mask False: train.py:392-393,407-410): the dense parameters,
    NO camera affine, 0.8 L1 + 0.2 (1-SSIM) + 0.02 l1_loss_v2(dense_rgb_colors, variables['dense_init_colors']).
    Returns (total, radius, rendervar, {'im': ..., 'soft_color': ...} unweighted)."""
    rendervar = params2rendervar_dense(params, variables) if params2rendervar_dense else params2rendervar_fused(_dense_view(params))
    rendervar['means2D'].retain_grad()
    im, radius, _, _ = GaussianRasterizer(raster_settings=curr_data['cam'])(**rendervar)
    if loss_fn is None:
        l_im = t4d_loss.photometric_loss(im[None], curr_data['im'][None])[0]
    else:
        l_im = loss_fn(im, curr_data['im'], None, None)
    soft = (soft_color_fn or t4d_loss.soft_color_loss)(params['dense_rgb_colors'], variables['dense_init_colors'])
    return l_im + soft_color_weight * soft, radius, rendervar, {'im': l_im, 'soft_color': soft}


# ------------------------------------------------------------------------------------------------------------
# one iteration chained by hand (no autograd)
# ------------------------------------------------------------------------------------------------------------
def explicit_iteration(params, curr_data, cam_grads=None, status_sink=None, target: Optional[torch.Tensor] = None,
                       dense: bool = False, soft_color=None, extra_loss: Optional[Callable] = None):
    """photometric_iteration (or, with `dense`, dense_iteration) + loss.backward() WITHOUT autograd: t4d_rasterize_forward,
    t4d_photometric_loss, t4d_rasterize_backward chained by hand - the activations and their backward inside the rasterizer
    (T4D_FLAG_RAW_PARAMS: the arithmetic of t4d_activate_forward / t4d_activate_backward, no launch of their own) - and none of
    the launches autograd puts around them (the zeros + 0 of means2D, ones_like for the root, a fill and a copy for every
    `[cid]` / `[0]` it differentiates through, the multiplication of dL/dim by a cotangent of one: nine launches of 4-5 us per
    iteration, a third of a 148 us graphed iteration).  For the precomputed-RGB, scale + rotation parametrisation of
    train.py:303-315 / :385-393 with the fused loss.
    `target`: the image to compare with (target_image(...); default curr_data['im']).
    `dense`: the dense_* parameters, no camera affine (get_loss_dense); `soft_color` = (dense_init_colors, weight) adds
    weight * l1_loss_v2 to the loss and its gradient to dL/d dense_rgb_colors (t4d_soft_color_loss).
    `extra_loss(params, rendervar) -> scalar`: further loss terms on the parameters (Topo4D's topology regularisers, train.py:330-368,
    which EVERY iteration of the real loop carries) - differentiated through autograd on their own (the fused activations give them
    `rendervar`), their gradients ADDED to the render's hand-chained ones: the photometric part of the iteration still runs without
    autograd.  (The sum of two gradients is order-independent; a parameter with more extra terms than one is summed in autograd's
    order: equal to the all-autograd iteration to rounding, not bit for bit.)
    `cam_grads`: {'cam_m': [n_cams, 3], 'cam_c': ...} persistent ZERO buffers; the loss kernel writes row `id` of each.
    `status_sink`: data pointer of 16 bytes of pinned host memory for the forward's status block (ViewBatch.status_sink).
    Returns (loss: device scalar, radius, grads: {parameter name: gradient tensor} for the parameters that require a gradient,
    ViewBatch, dL/dmeans2D)."""
    from . import rasterizer as R
    cam = curr_data['cam']
    n = _names(dense)
    dev = params[n['means3D']].device
    d = lambda k: params[n[k]].detach()
    batch = R.ViewBatch(R.pack_views([cam], dev), int(cam.image_height), int(cam.image_width), float(cam.scale_modifier),
                        int(cam.sh_degree), debug=bool(cam.debug), prefiltered=bool(cam.prefiltered), cam_key=id(cam),
                        sync_mode=R.get_sync_mode(drop_in=True))      # the mode a differentiated drop-in call runs ("auto" by default)
    batch.flat_grads = True                               # the drop-in's shapes: no view axis
    batch.status_sink = status_sink                       # (lazy mode: pinned host words for the forward's status block)
    # T4D_FLAG_RAW_PARAMS: the rasterizer takes the optimiser's parameters as they are and returns their gradients - the
    # arithmetic of t4d_activate_forward / t4d_activate_backward inside the binning kernel and the per-Gaussian backward
    batch.raw_params = True
    im, radius, _, _ = batch.forward(d('means3D'), d('logit_opacities'), d('log_scales'), d('unnorm_rotations'),
                                     colors_precomp=d('rgb_colors'))
    cid = curr_data['id']
    cm = cc = dcm = dcc = None
    if not dense and 'cam_m' in params:
        cm, cc = params['cam_m'].detach()[cid:cid + 1], params['cam_c'].detach()[cid:cid + 1]
        if cam_grads is not None:
            dcm, dcc = cam_grads['cam_m'][cid:cid + 1], cam_grads['cam_c'][cid:cid + 1]
    gt = curr_data['im'] if target is None else target
    if gt.dtype != torch.float32:
        gt = gt.float()                                   # (the autograd path coerces too; get_dataset's images are float32 already)
    l, d_im, dcm, dcc = t4d_loss.photometric_loss_raw(im[None], gt[None] if gt.is_contiguous() else gt.contiguous()[None], cm, cc, dcm, dcc)
    try:
        g = batch.backward(d_im)
    except Exception:
        if cam_grads is not None and cm is not None:      # no optimiser step will clear the row the loss kernel has just written
            dcm.zero_(); dcc.zero_()
        raise
    total = l[0]
    if soft_color is not None:
        init, weight = soft_color
        l_soft, _ = t4d_loss.soft_color_loss_raw(d('rgb_colors'), init, float(weight), grad=g['colors_precomp'], accumulate=True)
        total = torch.add(total, l_soft, alpha=float(weight))
    by_role = {'means3D': g['means3D'], 'rgb_colors': g['colors_precomp'], 'unnorm_rotations': g['rotations'],
               'logit_opacities': g['opacities'], 'log_scales': g['scales']}
    # a parameter the caller froze with requires_grad_(False) gets no gradient and no step, as under autograd (dense_means3D
    # is such a tensor in Topo4D: train.py:259-261)
    grads = {n[k]: v for k, v in by_role.items() if params[n[k]].requires_grad}
    if cm is not None:
        for k, row in (('cam_m', dcm), ('cam_c', dcc)):
            if not params[k].requires_grad:
                if cam_grads is not None:
                    row.zero_()
                continue
            if cam_grads is not None:
                grads[k] = cam_grads[k]
            else:                                         # no persistent buffers: a full-size gradient with one row set
                full = torch.zeros_like(params[k])
                full[cid:cid + 1] = row
                grads[k] = full
    if extra_loss is not None:
        with torch.enable_grad():
            rendervar = params2rendervar_fused(_dense_view(params) if dense else params)
            l_extra = extra_loss(params, rendervar)
            names = [k for k, p in params.items() if isinstance(p, torch.Tensor) and p.requires_grad]
            g_extra = torch.autograd.grad(l_extra, [params[k] for k in names], allow_unused=True)
        for k, ge in zip(names, g_extra):
            if ge is None:
                continue
            if k in grads:
                grads[k].add_(ge)                         # (cam_m / cam_c: the persistent buffer the Adam step clears)
            else:
                grads[k] = ge
        total = total + l_extra.detach()
    return total, radius, grads, batch, g['means2D']


def explicit_frame_iteration(params, frame: List[dict], gt: Optional[torch.Tensor] = None, cam_grads=None, use_mask: bool = False,
                             is_initial_timestep: bool = True, label_colors=None):
    """explicit_iteration for ALL cameras of a frame in one launch set (24 views cost 0.45 ms where one costs 0.08): activations,
    one multi-view render, one batched loss, one multi-view backward, view-summed gradients (t4d_sum_views), activation backward -
    chained by hand, no autograd.  `frame`: the cameras' dataset entries; `gt`: their target images stacked [V,3,H,W] (stacked
    here from target_image(...) when None - pass it to keep that copy out of the loop).  The entries' ids must be one ascending
    range (rows of cam_m / cam_c are passed as a view).  Returns (per-view losses [V], radii [V,P], grads of the loss SUMMED
    over the views, ViewBatch)."""
    from . import rasterizer as R
    cams = [e['cam'] for e in frame]
    V = len(cams)
    dev = params['means3D'].device
    H, W, smod, deg = R._check_common(cams)
    d = lambda k: params[k].detach()
    ur = d('unnorm_rotations')
    rot, op, sc = activate_forward(ur, d('logit_opacities'), d('log_scales'))
    batch = R.ViewBatch(R.pack_views(cams, dev), H, W, smod, deg)
    im, radii, _, _ = batch.forward(d('means3D'), op, sc, rot, colors_precomp=d('rgb_colors'))
    if gt is None:
        if use_mask and not is_initial_timestep:
            prepare_masked_targets(frame, label_colors)
        gt = torch.stack([target_image(e, use_mask, is_initial_timestep, label_colors) for e in frame])
    cm = cc = dcm = dcc = None
    if 'cam_m' in params:
        i0 = frame[0]['id']
        if [e['id'] for e in frame] != list(range(i0, i0 + V)):
            raise ValueError("explicit_frame_iteration: the frame's camera ids must be one ascending range")
        cm, cc = d('cam_m')[i0:i0 + V], d('cam_c')[i0:i0 + V]
        if cam_grads is not None:
            dcm, dcc = cam_grads['cam_m'][i0:i0 + V], cam_grads['cam_c'][i0:i0 + V]
    l, d_im, dcm, dcc = t4d_loss.photometric_loss_raw(im, gt, cm, cc, dcm, dcc)
    g = R._sum_views(batch.backward(d_im), V, need_means2D=False)
    d_ur, d_lo, d_ls = activate_backward(ur, op, sc, g['rotations'], g['opacities'], g['scales'])
    grads = {'means3D': g['means3D'], 'rgb_colors': g['colors_precomp'], 'unnorm_rotations': d_ur, 'logit_opacities': d_lo,
             'log_scales': d_ls}
    grads = {k: v for k, v in grads.items() if params[k].requires_grad}
    if cm is not None:
        if cam_grads is not None:
            grads['cam_m'], grads['cam_c'] = cam_grads['cam_m'], cam_grads['cam_c']
        else:
            for k, rows in (('cam_m', dcm), ('cam_c', dcc)):
                full = torch.zeros_like(params[k])
                full[i0:i0 + V] = rows
                grads[k] = full
    return l, radii, grads, batch


# ------------------------------------------------------------------------------------------------------------
# the loops
# ------------------------------------------------------------------------------------------------------------
def _can_chain(params, optimizer, keys, loss_fn) -> bool:
    from .optim import FusedAdamPins
    return loss_fn is None and isinstance(optimizer, FusedAdamPins) and all(k in params for k in keys) \
        and ('cam_m' in params) == ('cam_c' in params) and params[keys[0]].is_cuda


def _bookkeep(radius, max_2D_radius) -> None:
    if max_2D_radius is not None:                                  # train.py:373-375 / :411-413 bookkeeping
        seen = radius > 0
        max_2D_radius[seen] = torch.max(radius[seen], max_2D_radius[seen])


def _adopt_cam_grads(params, optimizer):
    """Persistent gradient buffers of the per-camera affine (an iteration writes ONE row, the step leaves zeros behind)."""
    if 'cam_m' not in params:
        return None
    cam_grads = {k: torch.zeros_like(params[k]) for k in ('cam_m', 'cam_c')}
    optimizer.clear_grad |= {g["name"] for g in optimizer.param_groups if g["params"][0] is params['cam_m'] or g["params"][0] is params['cam_c']}
    return cam_grads


def optimise_views(params, dataset: List[dict], optimizer, n_iters: int, seed: int = 0, loss_fn: Optional[Callable] = None,
                   extra_loss: Optional[Callable] = None, max_2D_radius: Optional[torch.Tensor] = None,
                   explicit: Optional[bool] = None, use_mask: bool = False, is_initial_timestep: bool = True, label_colors=None):
    """train.py:661-673 for `n_iters` iterations.  Returns the list of per-iteration losses (device scalars, no sync).
    `use_mask`, `is_initial_timestep`: get_loss's branch (train.py:315-327; Topo4D runs use_mask=True, i.e. the masked target in
    every frame after the first - `label_colors` as in prepare_masked_targets, which is called here once).
    `explicit` (default: when possible - fused loss, a FusedAdamPins optimiser, the scale + rotation / RGB parametrisation): the
    render, its loss and its backward are chained by hand (explicit_iteration) instead of going through autograd - same
    arithmetic, a third of the host time; `extra_loss` terms (the regularisers of train.py:330-368) are differentiated on their
    own and their gradients added."""
    rng = Random(seed)
    todo: list = []
    losses = []
    masked = use_mask and not is_initial_timestep
    if masked:
        prepare_masked_targets(dataset, label_colors)
    can = _can_chain(params, optimizer, _RENDER_KEYS, loss_fn)
    if explicit and not can:
        raise ValueError("explicit=True needs the fused loss, a FusedAdamPins optimiser and the parameters " + ", ".join(_RENDER_KEYS))
    if can if explicit is None else explicit:
        before = set(optimizer.clear_grad)
        cam_grads = _adopt_cam_grads(params, optimizer)
        try:
            for _ in range(n_iters):
                curr, todo = get_batch(todo, dataset, rng)
                l, radius, grads, _, _ = explicit_iteration(params, curr, cam_grads, target=curr['masked_im'] if masked else None,
                                                            extra_loss=extra_loss)
                for k, gr in grads.items():
                    params[k].grad = gr
                optimizer.step()
                optimizer.zero_grad(set_to_none=True)
                _bookkeep(radius, max_2D_radius)
                losses.append(l)
        finally:
            optimizer.clear_grad = before
        return losses
    for _ in range(n_iters):
        curr, todo = get_batch(todo, dataset, rng)
        l, radius, _ = photometric_iteration(params, curr, loss_fn, extra_loss, use_mask=use_mask,
                                             is_initial_timestep=is_initial_timestep, label_colors=label_colors)
        l.backward()
        with torch.no_grad():
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
            _bookkeep(radius, max_2D_radius)
        losses.append(l.detach())
    return losses


def optimise_dense_views(params, variables, dataset: List[dict], optimizer, n_iters: int, seed: int = 0,
                         loss_fn: Optional[Callable] = None, soft_color_fn: Optional[Callable] = None,
                         pre_iteration: Optional[Callable] = None, max_2D_radius: Optional[torch.Tensor] = None,
                         explicit: Optional[bool] = None, soft_color_weight: float = SOFT_COLOR_WEIGHT):
    """The texture loop, train.py:729-741, for `n_iters` iterations (args.dense_opt_num = 301 per frame): pins on
    dense_rgb_colors BEFORE each render (train.py:731-734) -> get_loss_dense(use_mask=False) -> backward -> Adam.
    `variables['dense_init_colors']`: the soft-colour anchor (train.py:258,502).
    Pins: with a FusedAdamPins optimiser, the rows set by `optimizer.set_pin('dense_rgb_colors', index, 0.0)` are written before
    every render (apply_pins) and NOT after the step - the reference leaves the last step's values in those rows (they are what
    save_mesh exports); any other optimiser: pass `pre_iteration`, a callable run under no_grad before every render.
    Returns the list of per-iteration total losses (device scalars)."""
    from .optim import FusedAdamPins
    rng = Random(seed)
    todo: list = []
    losses = []
    fused_opt = isinstance(optimizer, FusedAdamPins)
    can = _can_chain(params, optimizer, _DENSE_KEYS, loss_fn) and soft_color_fn is None and pre_iteration is None
    if explicit and not can:
        raise ValueError("explicit=True needs the fused losses, no pre_iteration callable, a FusedAdamPins optimiser and the parameters " + ", ".join(_DENSE_KEYS))
    chain = can if explicit is None else explicit
    init = variables['dense_init_colors']
    for _ in range(n_iters):
        curr, todo = get_batch(todo, dataset, rng)
        with torch.no_grad():
            if pre_iteration is not None:
                pre_iteration()
            elif fused_opt:
                optimizer.apply_pins(('dense_rgb_colors',))
        if chain:
            l, radius, grads, _, _ = explicit_iteration(params, curr, dense=True, soft_color=(init, soft_color_weight))
            for k, gr in grads.items():
                params[k].grad = gr
        else:
            l, radius, _, _ = dense_iteration(params, variables, curr, loss_fn, soft_color_fn, soft_color_weight)
            l.backward()
            l = l.detach()
        with torch.no_grad():
            if fused_opt:
                optimizer.step(pins=False)
            else:
                optimizer.step()
            optimizer.zero_grad(set_to_none=True)
            _bookkeep(radius, max_2D_radius)
        losses.append(l)
    return losses


class GraphedViews:
    """The same iteration (activations -> render -> photometric loss -> backward -> Adam + pins), recorded ONCE per camera
    in a HIP graph and replayed: the reference's loop is launch-bound (about 25 launches of microseconds each per
    iteration, train.py:661-700), so replaying a graph removes the host from the loop.

        opt = FusedAdamPins(groups, eps=1e-15, capturable=True)
        gv = GraphedViews(params, dataset, opt)            # warm-up (learns the pair-arena size), then one capture per camera
        for it in range(n):
            curr_index = rng.randint(0, len(dataset) - 1)  # any schedule: the graphs are indexed by camera
            loss = gv.step(curr_index)                     # device scalar of that iteration (no synchronisation)
        gv.check()                                         # once in a while: raises if a replay outgrew its pair arena
        gv.load_frame(next_dataset)                        # next frame: new target images into the recorded buffers
                                                           # (dense=True: ALSO the re-bound dense_init_colors / dense_means3D, below)

    `use_mask` / `is_initial_timestep` / `label_colors`: get_loss's branch, as optimise_views.  `dense=True` (with `variables`)
    records the texture loop's iteration instead (pins before the render, dense parameters, no affine, soft colour).
    The recorded kernels read `variables['dense_init_colors']` and `params['dense_means3D']` through the pointers they had at
    capture time, and the reference RE-BINDS both to new tensors in every later frame (update_dense_states, train.py:498-507).
    `load_frame` therefore copies whatever the two dict entries hold now INTO the recorded buffers and binds the entries back to
    them; `step` refuses to replay while an entry points elsewhere (call load_frame after update_dense_states).
    Parameters, optimiser state and pins are the caller's tensors (updated in place by the replays); `opt.param_groups[i]
    ['lr']` may be changed between steps (helpers.update_optimizer does) - step() pushes it to the device copy.
    Replays are un-synchronised like the rasterizer's "lazy" mode: the arena learned during warm-up has 1.5x head-room;
    gv.check() reads the overflow flags back.
    """

    def __init__(self, params, dataset: List[dict], optimizer, loss_fn: Optional[Callable] = None, extra_loss: Optional[Callable] = None,
                 explicit: Optional[bool] = None, use_mask: bool = False, is_initial_timestep: bool = True, label_colors=None,
                 dense: bool = False, variables: Optional[dict] = None, soft_color_weight: float = SOFT_COLOR_WEIGHT):
        from . import rasterizer as R
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedViews needs FusedAdamPins(..., capturable=True)")
        self.params, self.dataset, self.opt = params, dataset, optimizer
        self.dense, self.variables, self.soft_color_weight = bool(dense), variables, float(soft_color_weight)
        self._branch = (bool(use_mask), bool(is_initial_timestep), label_colors)
        keys = _DENSE_KEYS if dense else _RENDER_KEYS
        if dense and (variables is None or 'dense_init_colors' not in variables):
            raise ValueError("GraphedViews(dense=True) needs variables['dense_init_colors'] (train.py:258,502)")
        dev = params[keys[0]].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedViews runs on the GPU only")
        # explicit: the iteration chained by hand (explicit_iteration) instead of recorded through autograd - the same
        # arithmetic without autograd's fill / copy / multiply launches (parameters after any number of steps are bit-identical,
        # tests/test_gpu_loop.py).  Needs the fused loss and the scale + rotation / RGB parametrisation; an extra loss term is
        # differentiated on its own inside the recorded iteration and its gradients added.
        can = loss_fn is None and all(k in params for k in keys) and ('cam_m' in params) == ('cam_c' in params)
        if explicit and not can:
            raise ValueError("explicit=True needs the fused loss and the parameters " + ", ".join(keys))
        self.explicit = can if explicit is None else bool(explicit)
        if dense and not self.explicit:
            raise ValueError("GraphedViews(dense=True) records the hand-chained iteration only (fused losses)")
        self._loss_fn, self._extra_loss = loss_fn, extra_loss
        self._cam_grads = None
        self._clear_before = set(optimizer.clear_grad)
        self.graphs, self._status_host = [], None
        # the targets the recorded loss kernels read: own buffers, so that load_frame() can bring in the next frame's images
        self._targets = [target_image(e, *self._branch).float().contiguous().clone() for e in dataset]
        # dense: the two non-parameter inputs of the recorded iteration, (dict, key, the tensor whose storage the graphs read)
        self._rebound = [(variables, 'dense_init_colors', variables['dense_init_colors']),
                         (params, 'dense_means3D', params['dense_means3D'])] if dense else []
        try:
            if self.explicit and not dense:
                self._cam_grads = _adopt_cam_grads(params, optimizer)
            self._build(R, dev)
        finally:
            optimizer.clear_grad = self._clear_before      # (the flag is recorded in the graphs; eager steps keep their gradients)
        optimizer.zero_grad(set_to_none=True)
        if self._cam_grads is not None:
            # the replays read the persistent buffers through the pointers recorded in the graphs; `.grad` stays None between steps
            for b in self._cam_grads.values():
                b.zero_()

    def _build(self, R, dev) -> None:
        params, dataset, optimizer = self.params, self.dataset, self.opt
        leaves = [g["params"][0] for g in optimizer.param_groups]
        # the warm-up below takes real optimisation steps; everything it touches is restored before the captures
        snap_p = [p.detach().clone() for p in leaves]
        prev_mode = R._save_sync_mode()              # restored without a trace (the drop-in's default must stay the default)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.stream(side):
                R.set_sync_mode("checked")
                for i, data in enumerate(dataset):                     # learns the pair-arena capacity of every camera
                    if self.explicit:
                        self._explicit_step(i, data)
                    else:
                        l, _, _ = self._autograd_forward(i, data)
                        l.backward()
                        with torch.no_grad():
                            optimizer.step()
                    optimizer.zero_grad(set_to_none=True)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            with torch.no_grad():
                for p, s0 in zip(leaves, snap_p):
                    p.copy_(s0)
                for p in leaves:
                    st = optimizer.state.get(p)
                    if st is not None and "exp_avg" in st:
                        st["exp_avg"].zero_(); st["exp_avg_sq"].zero_(); st["step"] = 0
                optimizer._hyper(dev)[0].zero_()
            optimizer.sync_hyper()
            self.graphs, self.losses, self.radii, self.means2D_grads = [], [], [], []
            # every tensor a captured kernel reads through a raw pointer must outlive the graphs: the packed camera records are
            # otherwise owned only by the rasterizer's (evicting) view cache
            self._keep = []
            # binning status words (overflow flag, pairs needed) of every captured forward, copied out INSIDE its graph: the
            # graphs share one memory pool, so a state buffer is only meaningful until the next graph replays
            self._status = torch.zeros(len(dataset), 4, dtype=torch.int32, device=dev)
            # explicit iterations of ONE small view: the binning kernel writes its status block into pinned host memory itself
            # (T4D_FLAG_ASYNC_STATUS): no copy node in the graph.  Each camera's 16 bytes start out as "nothing wrong".
            self._status_host = torch.zeros(len(dataset), 2, dtype=torch.int64).pin_memory() if self.explicit else None
            # per camera: True = its status lands in the pinned host words, False = in the device copy recorded in its graph
            self._status_on_host: List[bool] = []
            self._caps = []
            R.set_sync_mode("lazy")
            with warnings.catch_warnings():
                # the warm-up ran on a side stream, the captures run on torch's capture stream: torch points that out for
                # every AccumulateGrad node; both orders are synchronised above, so the hint does not apply
                warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")
                self._capture_all(R)
        finally:
            R._BATCH_LOG = None
            R._restore_sync_mode(prev_mode)

    def _autograd_forward(self, i, data):
        entry = dict(data)
        entry['im'] = self._targets[i]                 # (the branch is resolved already: the recorded buffer IS the target)
        return photometric_iteration(self.params, entry, self._loss_fn, self._extra_loss)

    def _explicit_step(self, i, data):
        """One iteration chained by hand: gradients handed to the optimiser as `.grad`, then the fused step."""
        sink = None
        if self._status_host is not None and torch.cuda.is_current_stream_capturing():
            sink = self._status_host.data_ptr() + 16 * i
        if self.dense:
            self.opt.apply_pins(('dense_rgb_colors',))                                   # train.py:731-734, before the render
            l, radius, grads, batch, g2d = explicit_iteration(self.params, data, None, sink, target=self._targets[i], dense=True,
                                                              soft_color=(self.variables['dense_init_colors'], self.soft_color_weight))
        else:
            l, radius, grads, batch, g2d = explicit_iteration(self.params, data, self._cam_grads, sink, target=self._targets[i],
                                                              extra_loss=self._extra_loss)
        for k, gr in grads.items():
            self.params[k].grad = gr
        self.opt.step(pins=not self.dense)
        return l, radius, batch, (grads, g2d)

    def _capture_all(self, R) -> None:
        pool = None
        for i, data in enumerate(self.dataset):
            self.opt.zero_grad(set_to_none=True)
            R._BATCH_LOG = []
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                if self.explicit:
                    l, radius, batch, keep = self._explicit_step(i, data)
                    self._keep.append((batch, keep))
                    self.means2D_grads.append(keep[1])
                    # did THIS camera's forward hand its status to the pinned host words?  (a checked or debug forward copies
                    # it itself and leaves the sink alone: then the device copy is recorded for this camera, and only for it)
                    on_host = self._status_host is not None and batch.status_sink is not None and \
                        bool(batch.prob.flags & R._lib.T4D_FLAG_ASYNC_STATUS) and not (batch.prob.flags & R._lib.T4D_FLAG_CHECKED)
                    if not on_host:
                        self._status[i].copy_(batch.state[:16].view(torch.int32))
                    self._status_on_host.append(on_host)
                else:
                    l, radius, _ = self._autograd_forward(i, data)
                    batch = R._BATCH_LOG[-1]
                    self._status[i].copy_(batch.state[:16].view(torch.int32))
                    self._status_on_host.append(False)
                    l.backward()
                    self.opt.step()
                self._keep.append(batch.views)
            pool = g.pool()
            self.graphs.append(g)
            self.losses.append(l.detach())
            self.radii.append(radius)
            self._caps.append(int(batch.prob.pair_capacity))

    def load_frame(self, dataset: List[dict]) -> None:
        """The next frame's images for the same cameras (train.py:653: get_dataset per frame): copies each camera's target -
        masked as the recorded branch says - into the buffer its graph reads.  The graphs, cameras and arenas stay."""
        if len(dataset) != len(self.dataset):
            raise ValueError("load_frame: the new frame must hold the same cameras in the same order")
        use_mask, initial, colors = self._branch
        if use_mask and not initial:
            prepare_masked_targets(dataset, colors)
        for buf, e in zip(self._targets, dataset):
            buf.copy_(target_image(e, use_mask, initial, colors))
        self.dataset = dataset
        # dense: update_dense_states (train.py:498-507) bound NEW tensors to these entries; the graphs read the old storage
        for d, key, recorded in self._rebound:
            cur = d[key]
            if cur is not recorded and cur.data_ptr() != recorded.data_ptr():
                if cur.shape != recorded.shape:
                    raise ValueError(f"load_frame: {key} changed its shape ({tuple(cur.shape)} vs the recorded {tuple(recorded.shape)}); "
                                     "build a new GraphedViews")
                with torch.no_grad():
                    recorded.copy_(cur)
            d[key] = recorded

    def step(self, index: int) -> torch.Tensor:
        """Replay the iteration of camera `index`; returns its loss (a device scalar that the next replay of the same camera
        overwrites)."""
        for d, key, recorded in self._rebound:
            if d[key].data_ptr() != recorded.data_ptr():
                raise RuntimeError(f"GraphedViews(dense=True): {key} was re-bound to a new tensor (update_dense_states, train.py:498-507) but the "
                                   "recorded iteration reads the old one: call load_frame(dataset) after update_dense_states")
        self.opt.sync_hyper()
        self.graphs[index].replay()
        return self.losses[index]

    def check(self) -> None:
        """Synchronising read of every captured forward's binning status: raises if a replay was truncated."""
        torch.cuda.synchronize(self._status.device)
        dev_words = self._status.tolist()
        host_words = self._status_host.tolist() if self._status_host is not None else None
        for i, on_host in enumerate(self._status_on_host):
            if on_host:
                w0 = int(host_words[i][0])
                overflow, need = w0 & 0xffffffff, (w0 >> 32) & 0xffffffff
            else:
                overflow, need = dev_words[i][0], dev_words[i][1]
            if overflow:
                raise RuntimeError(f"the graphed render of camera {i} needed {need} (Gaussian,tile) pairs per view, its recorded arena "
                                   f"holds {self._caps[i]}; build a new GraphedViews on the current scene")
