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

import warnings
from random import Random
from typing import Callable, List, Optional

import torch

from . import loss as t4d_loss
from .boundary import activate_backward, activate_forward, params2rendervar_fused
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
                          params2rendervar: Optional[Callable] = None):
    """One forward of get_loss's photometric branch (train.py:303-328, use_mask False); returns (loss, radius, rendervar).
    `params2rendervar`: the function that turns the optimiser's parameters into rasterizer kwargs - Topo4D's own
    helpers.params2rendervar (helpers.py:91-100) may be passed; the default is its fused equivalent (GPU only)."""
    rendervar = (params2rendervar or params2rendervar_fused)(params)
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


_RENDER_KEYS = ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales')


def explicit_iteration(params, curr_data, cam_grads=None, status_sink=None):
    """photometric_iteration + loss.backward() WITHOUT autograd: t4d_rasterize_forward, t4d_photometric_loss,
    t4d_rasterize_backward chained by hand - the activations and their backward inside the rasterizer (T4D_FLAG_RAW_PARAMS:
    the arithmetic of t4d_activate_forward / t4d_activate_backward, no launch of their own) - and none of the launches autograd
    puts around them (the zeros + 0 of means2D, ones_like for the root, a fill and a copy for every `[cid]` / `[0]` it
    differentiates through, the multiplication of dL/dim by a cotangent of one: nine launches of 4-5 us per iteration, a third
    of a 148 us graphed iteration).  For the precomputed-RGB, scale + rotation parametrisation of train.py:303-315 with the
    fused loss and no extra loss term.
    `cam_grads`: {'cam_m': [n_cams, 3], 'cam_c': ...} persistent ZERO buffers; the loss kernel writes row `id` of each.
    `status_sink`: data pointer of 16 bytes of pinned host memory for the forward's status block (ViewBatch.status_sink).
    Returns (loss: device scalar, radius, grads: {parameter name: gradient tensor}, ViewBatch, dL/dmeans2D)."""
    from . import rasterizer as R
    cam = curr_data['cam']
    dev = params['means3D'].device
    d = lambda k: params[k].detach()
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
    if 'cam_m' in params:
        cm, cc = d('cam_m')[cid:cid + 1], d('cam_c')[cid:cid + 1]
        if cam_grads is not None:
            dcm, dcc = cam_grads['cam_m'][cid:cid + 1], cam_grads['cam_c'][cid:cid + 1]
    gt = curr_data['im']
    l, d_im, dcm, dcc = t4d_loss.photometric_loss_raw(im[None], gt[None] if gt.is_contiguous() else gt.contiguous()[None], cm, cc, dcm, dcc)
    g = batch.backward(d_im)
    grads = {'means3D': g['means3D'], 'rgb_colors': g['colors_precomp'], 'unnorm_rotations': g['rotations'],
             'logit_opacities': g['opacities'], 'log_scales': g['scales']}
    if cm is not None:
        if cam_grads is not None:
            grads['cam_m'], grads['cam_c'] = cam_grads['cam_m'], cam_grads['cam_c']
        else:                                             # no persistent buffers: a full-size gradient with one row set
            for k, row in (('cam_m', dcm), ('cam_c', dcc)):
                full = torch.zeros_like(params[k])
                full[cid:cid + 1] = row
                grads[k] = full
    return l[0], radius, grads, batch, g['means2D']


def explicit_frame_iteration(params, frame: List[dict], gt: Optional[torch.Tensor] = None, cam_grads=None):
    """explicit_iteration for ALL cameras of a frame in one launch set (24 views cost 0.45 ms where one costs 0.08): activations,
    one multi-view render, one batched loss, one multi-view backward, view-summed gradients (t4d_sum_views), activation backward -
    chained by hand, no autograd.  `frame`: the cameras' dataset entries; `gt`: their target images stacked [V,3,H,W] (stacked
    here when None - pass it to keep that copy out of the loop).  The entries' ids must be one ascending range (rows of cam_m /
    cam_c are passed as a view).  Returns (per-view losses [V], radii [V,P], grads of the loss SUMMED over the views, ViewBatch)."""
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
        gt = torch.stack([e['im'] for e in frame])
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
    if cm is not None:
        if cam_grads is not None:
            grads['cam_m'], grads['cam_c'] = cam_grads['cam_m'], cam_grads['cam_c']
        else:
            for k, rows in (('cam_m', dcm), ('cam_c', dcc)):
                full = torch.zeros_like(params[k])
                full[i0:i0 + V] = rows
                grads[k] = full
    return l, radii, grads, batch


def optimise_views(params, dataset: List[dict], optimizer, n_iters: int, seed: int = 0, fused_loss: bool = True,
                   extra_loss: Optional[Callable] = None, max_2D_radius: Optional[torch.Tensor] = None,
                   explicit: Optional[bool] = None):
    """train.py:661-673 for `n_iters` iterations.  Returns the list of per-iteration losses (device scalars, no sync).
    `explicit` (default: when possible - fused loss, no extra loss, a FusedAdamPins optimiser, the scale + rotation / RGB
    parametrisation): every iteration is chained by hand (explicit_iteration) instead of going through autograd - same
    arithmetic, a third of the host time."""
    from .optim import FusedAdamPins
    rng = Random(seed)
    todo: list = []
    losses = []
    can = fused_loss and extra_loss is None and isinstance(optimizer, FusedAdamPins) and all(k in params for k in _RENDER_KEYS) \
        and ('cam_m' in params) == ('cam_c' in params) and params['means3D'].is_cuda
    if explicit and not can:
        raise ValueError("explicit=True needs the fused loss, no extra_loss, a FusedAdamPins optimiser and the parameters " + ", ".join(_RENDER_KEYS))
    if can if explicit is None else explicit:
        cam_grads, before = None, set(optimizer.clear_grad)
        if 'cam_m' in params:
            cam_grads = {k: torch.zeros_like(params[k]) for k in ('cam_m', 'cam_c')}
            optimizer.clear_grad |= {g["name"] for g in optimizer.param_groups if g["params"][0] is params['cam_m'] or g["params"][0] is params['cam_c']}
        try:
            for _ in range(n_iters):
                curr, todo = get_batch(todo, dataset, rng)
                l, radius, grads, _, _ = explicit_iteration(params, curr, cam_grads)
                for k, gr in grads.items():
                    params[k].grad = gr
                optimizer.step()
                optimizer.zero_grad(set_to_none=True)
                if max_2D_radius is not None:                      # train.py:373-375 bookkeeping
                    seen = radius > 0
                    max_2D_radius[seen] = torch.max(radius[seen], max_2D_radius[seen])
                losses.append(l)
        finally:
            optimizer.clear_grad = before
        return losses
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

    Parameters, optimiser state and pins are the caller's tensors (updated in place by the replays); `opt.param_groups[i]
    ['lr']` may be changed between steps (helpers.update_optimizer does) - step() pushes it to the device copy.
    Replays are un-synchronised like the rasterizer's "lazy" mode: the arena learned during warm-up has 1.5x head-room;
    gv.check() reads the overflow flags back.
    """

    def __init__(self, params, dataset: List[dict], optimizer, fused_loss: bool = True, extra_loss: Optional[Callable] = None,
                 explicit: Optional[bool] = None):
        from . import rasterizer as R
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedViews needs FusedAdamPins(..., capturable=True)")
        self.params, self.dataset, self.opt = params, dataset, optimizer
        dev = params['means3D'].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedViews runs on the GPU only")
        # explicit: the iteration chained by hand (explicit_iteration) instead of recorded through autograd - the same
        # arithmetic without autograd's fill / copy / multiply launches (parameters after any number of steps are bit-identical,
        # tests/test_gpu_loop.py).  Needs the fused loss, no extra loss term and the scale + rotation / RGB parametrisation.
        can = fused_loss and extra_loss is None and all(k in params for k in _RENDER_KEYS) and \
            ('cam_m' in params) == ('cam_c' in params)
        if explicit and not can:
            raise ValueError("explicit=True needs the fused loss, no extra_loss, and the parameters " + ", ".join(_RENDER_KEYS))
        self.explicit = can if explicit is None else bool(explicit)
        self._cam_grads = None
        self._clear_before = set(optimizer.clear_grad)
        if self.explicit and 'cam_m' in params:
            # persistent gradient buffers of the per-camera affine: an iteration writes ONE row, the step leaves zeros behind
            self._cam_grads = {k: torch.zeros_like(params[k]) for k in ('cam_m', 'cam_c')}
            optimizer.clear_grad |= {g["name"] for g in optimizer.param_groups if g["params"][0] is params['cam_m'] or g["params"][0] is params['cam_c']}
        self.graphs, self._status_host = [], None
        leaves = [g["params"][0] for g in optimizer.param_groups]
        # the warm-up below takes real optimisation steps; everything it touches is restored before the captures
        snap_p = [p.detach().clone() for p in leaves]
        prev_mode = R._save_sync_mode()              # restored without a trace (the drop-in's default must stay the default)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            R.set_sync_mode("checked")
            for data in dataset:                                   # learns the pair-arena capacity of every camera
                if self.explicit:
                    self._explicit_step(data)
                else:
                    l, _, _ = photometric_iteration(params, data, fused_loss, extra_loss)
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
        self._caps = []
        R.set_sync_mode("lazy")
        try:
            with warnings.catch_warnings():
                # the warm-up ran on a side stream, the captures run on torch's capture stream: torch points that out for
                # every AccumulateGrad node; both orders are synchronised above, so the hint does not apply
                warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")
                self._capture_all(R, fused_loss, extra_loss)
        finally:
            R._BATCH_LOG = None
            R._restore_sync_mode(prev_mode)
        optimizer.zero_grad(set_to_none=True)
        optimizer.clear_grad = self._clear_before          # (the flag is recorded in the graphs; eager steps keep their gradients)
        if self._cam_grads is not None:
            # the replays read the persistent buffers through the pointers recorded in the graphs; `.grad` stays None between steps
            for b in self._cam_grads.values():
                b.zero_()

    def _explicit_step(self, data):
        """One iteration chained by hand: gradients handed to the optimiser as `.grad`, then the fused step."""
        sink = None
        if self._status_host is not None and torch.cuda.is_current_stream_capturing():
            sink = self._status_host.data_ptr() + 16 * len(self.graphs)
        l, radius, grads, batch, g2d = explicit_iteration(self.params, data, self._cam_grads, sink)
        for k, gr in grads.items():
            self.params[k].grad = gr
        self.opt.step()
        return l, radius, batch, (grads, g2d)

    def _capture_all(self, R, fused_loss, extra_loss) -> None:
        pool = None
        for data in self.dataset:
            self.opt.zero_grad(set_to_none=True)
            R._BATCH_LOG = []
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                if self.explicit:
                    l, radius, batch, keep = self._explicit_step(data)
                    self._keep.append((batch, keep))
                    self.means2D_grads.append(keep[1])
                    if not (batch.prob.flags & R._lib.T4D_FLAG_ASYNC_STATUS):        # (the library took the status sink?)
                        self._status_host = None
                    if self._status_host is None:
                        self._status[len(self.graphs)].copy_(batch.state[:16].view(torch.int32))
                else:
                    l, radius, _ = photometric_iteration(self.params, data, fused_loss, extra_loss)
                    batch = R._BATCH_LOG[-1]
                    self._status[len(self.graphs)].copy_(batch.state[:16].view(torch.int32))
                    l.backward()
                    self.opt.step()
                self._keep.append(batch.views)
            pool = g.pool()
            self.graphs.append(g)
            self.losses.append(l.detach())
            self.radii.append(radius)
            self._caps.append(int(batch.prob.pair_capacity))

    def step(self, index: int) -> torch.Tensor:
        """Replay the iteration of camera `index`; returns its loss (a device scalar that the next replay of the same camera
        overwrites)."""
        self.opt.sync_hyper()
        self.graphs[index].replay()
        return self.losses[index]

    def check(self) -> None:
        """Synchronising read of every captured forward's binning status: raises if a replay was truncated."""
        if self._status_host is not None:
            torch.cuda.synchronize(self._status.device)
            words = [(int(w0) & 0xffffffff, (int(w0) >> 32) & 0xffffffff, 0, 0) for w0, _ in self._status_host.tolist()]
        else:
            words = self._status.tolist()
        for i, (overflow, need, _, _) in enumerate(words):
            if overflow:
                raise RuntimeError(f"the graphed render of camera {i} needed {need} (Gaussian,tile) pairs per view, its recorded arena "
                                   f"holds {self._caps[i]}; build a new GraphedViews on the current scene")
