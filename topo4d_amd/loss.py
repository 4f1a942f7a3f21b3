"""
Photometric loss on the rendered image — the consumer of the rasterizer output whose gradient is the rasterizer's
backward input (reference train.py:310,315: per-camera affine, then 0.8*L1 + 0.2*(1-SSIM)).

`photometric_loss_torch` restates helpers.py:115-116 (`l1_loss_v1`) and external.py:73-116 (`calc_ssim`: 11x11
Gaussian window, sigma 1.5, zero padding, c1 = 0.01^2, c2 = 0.03^2, mean over all pixels) with plain torch ops;
it is pinned by tests/golden/g3_photometric.npz (values captured from the real reference functions).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

WINDOW = 11
SIGMA = 1.5
C1 = 0.01 ** 2
C2 = 0.03 ** 2


def gaussian_window_1d(dtype=torch.float32, device="cpu") -> torch.Tensor:
    g = torch.tensor([math.exp(-(x - WINDOW // 2) ** 2 / float(2 * SIGMA ** 2)) for x in range(WINDOW)], dtype=dtype)
    return (g / g.sum()).to(device)


def ssim_torch(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """external.py:85-116 with size_average=True.  img: [C,H,W] or [N,C,H,W]."""
    squeeze = img1.dim() == 3
    if squeeze:
        img1, img2 = img1[None], img2[None]
    ch = img1.shape[1]
    w1 = gaussian_window_1d(img1.dtype, img1.device)
    w2 = (w1[:, None] @ w1[None, :])[None, None].expand(ch, 1, WINDOW, WINDOW).contiguous()
    pad = WINDOW // 2
    conv = lambda x: F.conv2d(x, w2, padding=pad, groups=ch)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq
    s2 = conv(img2 * img2) - mu2_sq
    s12 = conv(img1 * img2) - mu1_mu2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def photometric_loss_torch(im: torch.Tensor, gt: torch.Tensor, cam_m: torch.Tensor = None, cam_c: torch.Tensor = None):
    """train.py:310,315: im' = exp(cam_m)[:,None,None]*im + cam_c[:,None,None]; 0.8*mean|im'-gt| + 0.2*(1-SSIM)."""
    if cam_m is not None:
        im = torch.exp(cam_m)[:, None, None] * im + cam_c[:, None, None]
    return 0.8 * torch.abs(im - gt).mean() + 0.2 * (1.0 - ssim_torch(im, gt))


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
# masked L1 of the dense pass (reference train.py:394-405, get_loss_dense with use_mask=True)
# ------------------------------------------------------------------------------------------------------------
def masked_l1_loss_torch(im: torch.Tensor, gt: torch.Tensor, filtered_mask: torch.Tensor) -> torch.Tensor:
    """train.py:400-405 restated: masked copies of the render and the target, L1 sum over the number of masked ELEMENTS
    (the mask image carries the same plane in its three channels; all of them count).  Pinned by tests/golden/g8."""
    masked_index = filtered_mask == 1
    masked_im = torch.zeros_like(im)
    masked_im[masked_index] = im[masked_index]
    masked_gt = torch.zeros_like(im)
    masked_gt[masked_index] = gt[masked_index]
    return (masked_im - masked_gt).abs().sum() / masked_index.sum()


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
