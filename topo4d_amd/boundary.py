"""
Fused parameter activations of Topo4D's render boundary (reference helpers.py:91-100, `params2rendervar`): the same dictionary
of rasterizer kwargs with `F.normalize` / `sigmoid` / `exp` and their autograd in one HIP launch each way
(t4d_activate_forward / t4d_activate_backward) instead of ~15 tiny torch launches per iteration.  Opt-in.

The plain mirrors of the reference helpers that the scene generator, bench and tests use live in scaffold/reference_boundary.py.
"""
from __future__ import annotations

import torch


def _check_raw(**tensors):
    """The raw entry points hand data pointers to the library: contiguous float32 HIP tensors only ([P,4] / [P,1] / [P,3])."""
    P = None
    for name, t in tensors.items():
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("params2rendervar_fused runs on the GPU only (no CPU fallback)")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous float32 tensor")
        P = t.shape[0] if P is None else P
        if t.shape[0] != P:
            raise ValueError(f"{name} holds {t.shape[0]} rows, expected {P}")


def activate_forward(unnorm_rotations, logit_opacities, log_scales):
    """rotations, opacities, scales = normalize(unnorm_rotations), sigmoid(logit_opacities), exp(log_scales) in ONE launch
    (t4d_activate_forward); no autograd - contiguous fp32 HIP tensors in, new tensors out."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    dev = unnorm_rotations.device
    if dev.type != "cuda":
        raise RuntimeError("params2rendervar_fused runs on the GPU only (no CPU fallback)")
    ur, lo, ls = unnorm_rotations, logit_opacities, log_scales
    _check_raw(ur=ur, lo=lo, ls=ls)
    rot, op, sc = torch.empty_like(ur), torch.empty_like(lo), torch.empty_like(ls)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.t4d_activate_forward(ur.shape[0], p(ur), p(lo), p(ls), p(rot), p(op), p(sc), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"t4d_activate_forward failed (code {rc}): {_lib.last_error()}")
    return rot, op, sc


def activate_backward(unnorm_rotations, opacities, scales, g_rot, g_op, g_sc, need=(True, True, True)):
    """The vector-Jacobian products of activate_forward in one launch (t4d_activate_backward): `opacities` / `scales` are the
    forward OUTPUTS, a None cotangent counts as zeros; returns (d_unnorm_rotations, d_logit_opacities, d_log_scales)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    ur, op, sc = unnorm_rotations, opacities, scales
    dev = ur.device
    _check_raw(ur=ur, op=op, sc=sc, g_rot=g_rot, g_op=g_op, g_sc=g_sc)
    d_ur = torch.empty_like(ur) if need[0] else None
    d_lo = torch.empty_like(op) if need[1] else None
    d_ls = torch.empty_like(sc) if need[2] else None
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rc = lib.t4d_activate_backward(ur.shape[0], p(ur), p(op), p(sc), p(g_rot), p(g_op), p(g_sc), p(d_ur), p(d_lo), p(d_ls),
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"t4d_activate_backward failed (code {rc}): {_lib.last_error()}")
    return d_ur, d_lo, d_ls


class _Activate(torch.autograd.Function):
    """rotations, opacities, scales = normalize(unnorm_rotations), sigmoid(logit_opacities), exp(log_scales) in ONE launch
    (t4d_activate_forward) and their vector-Jacobian products in one more (t4d_activate_backward)."""

    @staticmethod
    def forward(ctx, unnorm_rotations, logit_opacities, log_scales):
        if unnorm_rotations.device.type != "cuda":
            raise RuntimeError("params2rendervar_fused runs on the GPU only (no CPU fallback)")
        ur = unnorm_rotations.detach().contiguous().float()
        lo = logit_opacities.detach().contiguous().float()
        ls = log_scales.detach().contiguous().float()
        rot, op, sc = activate_forward(ur, lo, ls)
        ctx.save_for_backward(ur, op, sc)
        ctx.set_materialize_grads(False)
        return rot, op, sc

    @staticmethod
    def backward(ctx, g_rot, g_op, g_sc):
        ur, op, sc = ctx.saved_tensors
        c = lambda g: None if g is None else g.contiguous().float()
        return activate_backward(ur, op, sc, c(g_rot), c(g_op), c(g_sc), ctx.needs_input_grad)


def params2rendervar_fused(params):
    """Same dictionary as params2rendervar (helpers.py:91-100) with the three activations and their gradients fused into
    one kernel each (opt-in: replaces ~15 tiny torch launches per iteration with 2)."""
    rot, op, sc = _Activate.apply(params['unnorm_rotations'], params['logit_opacities'], params['log_scales'])
    return {
        'means3D': params['means3D'],
        'colors_precomp': params['rgb_colors'],
        'rotations': rot,
        'opacities': op,
        'scales': sc,
        'means2D': torch.zeros_like(params['means3D'], requires_grad=True) + 0,
    }
