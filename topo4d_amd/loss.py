"""
Loss assembly around the rasterizer - the consumers of the rasterizer output whose gradient is the rasterizer's backward input
(reference train.py:300-328 get_loss, :380-417 get_loss_dense) as fused HIP kernels behind the C ABI:

    photometric_loss[_raw]   t4d_photometric_loss    per-camera affine (train.py:310) + 0.8*L1 + 0.2*(1-SSIM) (train.py:315; helpers.py:115-116,
                                                     external.py:73-116), forward and gradient in one launch
    label_mask_target        t4d_label_mask_target   helpers.get_mask (helpers.py:811-823) + masked_gt (train.py:320-326), once per (frame, camera)
    soft_color_loss[_raw]    t4d_soft_color_loss     helpers.l1_loss_v2 (helpers.py:119-120): the 'soft_color' term of get_loss_dense (train.py:407)
    masked_l1_loss           t4d_masked_l1_loss      get_loss_dense's use_mask=True branch (train.py:394-405) - DISABLED in the reference
                                                     (train.py:632 use_mask_dense = False); kept because the API has it

GPU only: there is no torch fallback in this package.  The plain-torch restatements these kernels are checked against live in
oracle/loss_oracle.py (test infrastructure), pinned by goldens captured from the real reference (G3, G8, G9, G10).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch


# ------------------------------------------------------------------------------------------------------------
# fused HIP version (t4d_photometric_loss): forward + gradient in one launch set for a batch of views
# ------------------------------------------------------------------------------------------------------------
def photometric_loss_raw(im, gt, cam_m=None, cam_c=None, d_cam_m=None, d_cam_c=None):
    """t4d_photometric_loss without autograd: contiguous fp32 HIP tensors im, gt [V,3,H,W], cam_m / cam_c [V,3] or None.
    Returns (loss [V], dL/dim [V,3,H,W], dL/dcam_m, dL/dcam_c); `d_cam_m` / `d_cam_c`: [V,3] tensors that receive the camera
    gradients (e.g. rows of a persistent gradient buffer), allocated here when None."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    if not im.is_cuda:
        raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
    V, _, H, W = im.shape
    dev = im.device
    for name, t, shape in (("im", im, (V, 3, H, W)), ("gt", gt, (V, 3, H, W)), ("cam_m", cam_m, (V, 3)), ("cam_c", cam_c, (V, 3)),
                           ("d_cam_m", d_cam_m, (V, 3)), ("d_cam_c", d_cam_c, (V, 3))):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or tuple(t.shape) != shape):
            raise ValueError(f"photometric_loss_raw: {name} must be a contiguous float32 tensor of shape {shape} on {dev}")
    if (cam_m is None) != (cam_c is None) or (d_cam_m is None) != (d_cam_c is None):
        raise ValueError("photometric_loss_raw: cam_m / cam_c (and their gradient buffers) come in pairs")
    loss = torch.empty(V, dtype=torch.float32, device=dev)
    d_im = torch.empty_like(im)
    have_cam = cam_m is not None
    if have_cam and d_cam_m is None:
        d_cam_m = torch.empty(V, 3, dtype=torch.float32, device=dev)
        d_cam_c = torch.empty(V, 3, dtype=torch.float32, device=dev)
    nbytes = lib.t4d_photometric_scratch_bytes(V, H, W)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rc = lib.t4d_photometric_loss(V, H, W, p(im), p(gt), p(cam_m), p(cam_c), None, p(loss), p(d_im),
                                  p(d_cam_m) if have_cam else None, p(d_cam_c) if have_cam else None,
                                  p(scratch), nbytes, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"t4d_photometric_loss failed (code {rc}): {_lib.last_error()}")
    return loss, d_im, (d_cam_m if have_cam else None), (d_cam_c if have_cam else None)


class _FusedPhotometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, gt, cam_m, cam_c):
        if not im.is_cuda:
            raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
        have_cam = cam_m is not None
        loss, d_im, d_m, d_c = photometric_loss_raw(im.float().contiguous(), gt.float().contiguous(),
                                                    cam_m.float().contiguous() if have_cam else None,
                                                    cam_c.float().contiguous() if have_cam else None)
        ctx.save_for_backward(d_im, d_m, d_c)
        return loss

    @staticmethod
    def backward(ctx, go):
        d_im, d_m, d_c = ctx.saved_tensors
        g_im = d_im * go.view(-1, 1, 1, 1)
        g_m = None if d_m is None else d_m * go.view(-1, 1)
        g_c = None if d_c is None else d_c * go.view(-1, 1)
        return g_im, None, g_m, g_c


def photometric_loss(im: torch.Tensor, gt: torch.Tensor, cam_m: torch.Tensor = None, cam_c: torch.Tensor = None) -> torch.Tensor:
    """Per-view loss [V] for im, gt [V,3,H,W] (cam_m, cam_c [V,3] optional) — train.py:310,315 fused on the GPU.
    Differentiable w.r.t. im, cam_m, cam_c.  The reference's own call shape - ONE [3,H,W] render and target (cam_m, cam_c [3])
    - returns a scalar."""
    if im.dim() == 3:
        one = lambda t: None if t is None else t.unsqueeze(0)
        return _FusedPhotometric.apply(im.unsqueeze(0), one(gt), one(cam_m), one(cam_c))[0]
    return _FusedPhotometric.apply(im, gt, cam_m, cam_c)


# ------------------------------------------------------------------------------------------------------------
# masked L1 of the dense pass (reference train.py:394-405, get_loss_dense with use_mask=True: a branch train.py:632 disables)
# ------------------------------------------------------------------------------------------------------------
class _FusedMaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, gt, mask):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        if not im.is_cuda:
            raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
        im_c, gt_c, m_c = im.float().contiguous(), gt.float().contiguous(), mask.float().contiguous()
        ctx.im_dtype = im.dtype
        V, _, H, W = im_c.shape
        dev = im_c.device
        loss = torch.empty(V, dtype=torch.float32, device=dev)
        d_im = torch.empty_like(im_c)
        nbytes = lib.t4d_masked_l1_scratch_bytes(V)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = lib.t4d_masked_l1_loss(V, H, W, p(im_c), p(gt_c), p(m_c), None, p(loss), p(d_im), p(scratch), nbytes,
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"t4d_masked_l1_loss failed (code {rc}): {_lib.last_error()}")
        ctx.save_for_backward(d_im)
        return loss

    @staticmethod
    def backward(ctx, go):
        (d_im,) = ctx.saved_tensors
        g = d_im * go.view(-1, 1, 1, 1)
        return (g if g.dtype == ctx.im_dtype else g.to(ctx.im_dtype)), None, None


def masked_l1_loss(im: torch.Tensor, gt: torch.Tensor, filtered_mask: torch.Tensor) -> torch.Tensor:
    """Per-view masked L1 [V] for im, gt, filtered_mask [V,3,H,W] - train.py:394-405 fused on the GPU (loss + dL/dim in two
    launches).  Differentiable w.r.t. im.  The reference's own call shape (get_loss_dense: ONE [3,H,W] render, target and
    mask) returns a scalar."""
    if im.dim() == 3:
        return _FusedMaskedL1.apply(im.unsqueeze(0), gt.unsqueeze(0), filtered_mask.unsqueeze(0))[0]
    return _FusedMaskedL1.apply(im, gt, filtered_mask)


# ------------------------------------------------------------------------------------------------------------
# label mask + masked target (helpers.py:811-823, train.py:320-326): once per (frame, camera), not once per iteration
# ------------------------------------------------------------------------------------------------------------
MASK_SCALE = 0.1                      # train.py:326


def label_mask_target(mask_image: torch.Tensor, label_colors, gt: Optional[torch.Tensor] = None, scale: float = MASK_SCALE,
                      want_mask: bool = True):
    """helpers.get_mask for the selected labels and, with `gt`, the masked target of train.py:324-326, for one image [3,H,W] or
    the V cameras of a frame [V,3,H,W] in ONE launch.  `label_colors`: [n,3] colours of the selected labels (a tensor, array or
    nested list; the reference's are uint8: helpers.py:806 `cmap[cmap_index[label]]`), in the channel order of the mask image.
    Returns (filtered_mask or None, target or None), shaped like the inputs; bit-identical to the reference's torch ops."""
    from . import _lib
    lib = _lib.load()
    if not mask_image.is_cuda:
        raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
    one = mask_image.dim() == 3
    m = (mask_image[None] if one else mask_image)
    if m.dim() != 4 or m.shape[1] != 3:
        raise ValueError("label_mask_target: mask_image must be [3,H,W] or [V,3,H,W]")
    m = m.float().contiguous()
    V, _, H, W = m.shape
    g = None
    if gt is not None:
        g = (gt[None] if one else gt).float().contiguous()
        if g.shape != m.shape or g.device != m.device:
            raise ValueError("label_mask_target: gt must have the shape and device of mask_image")
    cols = torch.as_tensor(label_colors).detach().cpu().reshape(-1, 3).to(torch.float32)
    n = cols.shape[0]
    if n > _lib.T4D_MAX_MASK_LABELS:
        raise ValueError(f"label_mask_target: at most {_lib.T4D_MAX_MASK_LABELS} labels")
    host = (C.c_float * max(3 * n, 1))(*cols.flatten().tolist())
    filtered = torch.empty_like(m) if want_mask else None
    target = torch.empty_like(m) if g is not None else None
    if filtered is None and target is None:
        raise ValueError("label_mask_target: nothing to compute (no gt and want_mask False)")
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rc = lib.t4d_label_mask_target(V, H, W, p(m), host, n, p(g), float(scale), p(filtered), p(target),
                                   C.c_void_p(torch.cuda.current_stream(m.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"t4d_label_mask_target failed (code {rc}): {_lib.last_error()}")
    if one:
        filtered = None if filtered is None else filtered[0]
        target = None if target is None else target[0]
    return filtered, target


def get_mask(target_labels, mask, cmap_index, target_colors) -> torch.Tensor:
    """Drop-in for helpers.get_mask (helpers.py:811-823), same arguments: `target_labels` label names, `mask` the [3,H,W] label
    image (colours / 255), `cmap_index` name -> label index (train.py:50-55), `target_colors` the per-label colour tiles
    (train.py:635: a list of [3,H,W] tensors, one constant colour each - only their first texel is read).  One launch instead of
    four torch ops per label; returns the same float image of zeros and ones, bit for bit."""
    colors = [[float(target_colors[cmap_index[label]][c].reshape(-1)[0]) for c in range(3)] for label in target_labels]
    return label_mask_target(mask, colors, None)[0]


# ------------------------------------------------------------------------------------------------------------
# soft colour (helpers.py:119-120 l1_loss_v2; train.py:407 with weight 0.02, train.py:541-543)
# ------------------------------------------------------------------------------------------------------------
def soft_color_loss_raw(x: torch.Tensor, y: torch.Tensor, weight: float, grad: Optional[torch.Tensor] = None, accumulate: bool = False):
    """t4d_soft_color_loss without autograd: returns (UNWEIGHTED l1_loss_v2 as a device scalar, grad).  `grad` [rows,width]
    receives (accumulate: is increased by) (weight / rows) * sign(x - y); allocated here when None."""
    from . import _lib
    lib = _lib.load()
    if not x.is_cuda:
        raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
    if grad is None:
        if accumulate:
            raise ValueError("soft_color_loss_raw: accumulate needs a gradient tensor to add to")
        grad = torch.empty_like(x)
    for name, t in (("x", x), ("y", y), ("grad", grad)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != x.device or t.shape != x.shape or t.dim() != 2:
            raise ValueError(f"soft_color_loss_raw: {name} must be a contiguous float32 [rows,width] tensor on {x.device}")
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    nbytes = lib.t4d_soft_color_scratch_bytes()
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.t4d_soft_color_loss(x.shape[0], x.shape[1], p(x), p(y), float(weight), p(loss), p(grad), int(bool(accumulate)), p(scratch),
                                 nbytes, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"t4d_soft_color_loss failed (code {rc}): {_lib.last_error()}")
    return loss, grad


class _FusedSoftColor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        rows = x.shape[0]
        # weight = rows: the kernel leaves (rows / rows) * sign(x - y) = sign(x - y) exactly; backward scales it by grad / rows like
        # torch's mean backward.  The divisor is a DEVICE scalar on purpose: torch divides a GPU tensor by a host scalar as a
        # multiplication by the rounded reciprocal (one bit off); tensor / tensor is a true division - golden G10 bit for bit.
        loss, sign = soft_color_loss_raw(x.float().contiguous(), y.float().contiguous(), float(rows))
        ctx.save_for_backward(sign, torch.full((), float(rows), dtype=torch.float32, device=x.device))
        return loss

    @staticmethod
    def backward(ctx, go):
        sign, rows = ctx.saved_tensors
        return sign * (go / rows), None


def soft_color_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """helpers.l1_loss_v2(x, y) (helpers.py:119-120) on the GPU, differentiable w.r.t. x."""
    return _FusedSoftColor.apply(x, y)
