"""
TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  The product path (topo4d_amd) never does.

torch_oracle.py — differentiable CPU restatement of the Gaussian-splatting rasterizer that
Topo4D calls at /root/reference train.py:307,388,463,484 through the boundary built in
helpers.py:63-112.  Gradients come from torch.autograd (float64 by default), which makes this
file the *ground truth for every backward formula* used by oracle/raster_oracle.c and by the
HIP kernels.

PARITY UNPINNED.  The rasterizer's own source (`diff_gaussian_rasterization`, ashawkey fork,
unpinned HEAD — README.md:22-24) is not vendored under /root/reference and the reference has no
tests.  What is restated here is the library's published algorithm
(SURVEY.md Appendix A); constants are parsed from include/t4d_config.h so that the oracles and
the kernels cannot drift apart.  The in-repo pieces that ARE pinned by golden vectors generated
from the real reference (tests/golden/, oracle/gen_golden.py): matrix layout produced by
setup_camera (helpers.py:63-88), activations of params2rendervar (helpers.py:91-100), the SH
basis (helpers.py:836-922).

Deliberate deviations of the published backward from exact calculus, reproduced here so that
autograd returns what the CUDA library returns:
  * alpha = min(0.99, opacity*G) is differentiated as if unclamped (straight-through);
  * when the EWA frustum clamp (±1.3·tanfov) is active, the clamped coordinate is treated as a
    constant (its gradient is dropped);
  * quaternions are used un-normalised (Topo4D normalises outside, helpers.py:95).
  * the conic backward divides by (det² + 1e-7) instead of det² (`_ConicOfCov2D.backward`, the same
    expression as oracle/raster_oracle.c and the kernels; with the +0.3 dilation det >= 0.09, so the
    difference from exact calculus is <= 1.3e-5 relative).
"""
from __future__ import annotations

import math
import os
import re
from typing import NamedTuple, Optional

import numpy as np
import torch


def _load_constants():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "t4d_config.h")
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"#define\s+(T4D_\w+)\s+\(?(-?[0-9.]+)f?\s*(?:/\s*([0-9.]+)f?)?\)?", txt):
        v = float(m.group(2))
        if m.group(3):
            v = v / float(m.group(3))
        out[m.group(1)] = v
    return out


C = _load_constants()
TILE = int(C["T4D_TILE_X"])
assert TILE == int(C["T4D_TILE_Y"]) == 16


class _ConicOfCov2D(torch.autograd.Function):
    """conic = (c, -b, a) / det of the dilated 2D covariance [[a, b], [b, c]].  The published library's backward
     multiplies by 1 / (det² + 1e-7) where calculus has 1 / det²; restated here so that
    autograd returns what the library returns."""

    @staticmethod
    def forward(ctx, a, b, c, det_safe):
        ctx.save_for_backward(a, b, c, det_safe)
        return torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=1)

    @staticmethod
    def backward(ctx, g):
        a, b, c, det = ctx.saved_tensors
        X, Y, Z = g[:, 0], g[:, 1], g[:, 2]              # dL/d(conic A, B, C)
        d2inv = 1.0 / (det * det + C["T4D_CONIC_BWD_EPS"])
        da = d2inv * (-c * c * X + b * c * Y + (det - a * c) * Z)
        dc = d2inv * (-a * a * Z + a * b * Y + (det - a * c) * X)
        db = d2inv * (2.0 * b * c * X - (det + 2.0 * b * b) * Y + 2.0 * a * b * Z)
        return da, db, dc, None


class View(NamedTuple):
    """Same twelve fields, same order, as GaussianRasterizationSettings (helpers.py:73-86)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor   # [1,4,4] or [4,4], TRANSPOSED world->view (helpers.py:67)
    projmatrix: torch.Tensor   # [1,4,4] or [4,4], TRANSPOSED full projection (helpers.py:72)
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


def f32(x):
    """Round a python float to the nearest float32 (constants are float literals in the kernels)."""
    return float(np.float32(x))


def sh_basis(deg: int, d: torch.Tensor):
    """Real SH basis values [P, (deg+1)^2] for unit directions d [P,3]; signs/order as helpers.py:876-905."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, C["T4D_SH_C0"])]
    if deg > 0:
        b += [-C["T4D_SH_C1"] * y, C["T4D_SH_C1"] * z, -C["T4D_SH_C1"] * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C["T4D_SH_C2_0"] * xy, C["T4D_SH_C2_1"] * yz, C["T4D_SH_C2_2"] * (2.0 * zz - xx - yy),
              C["T4D_SH_C2_3"] * xz, C["T4D_SH_C2_4"] * (xx - yy)]
    if deg > 2:
        b += [C["T4D_SH_C3_0"] * y * (3.0 * xx - yy), C["T4D_SH_C3_1"] * xy * z,
              C["T4D_SH_C3_2"] * y * (4.0 * zz - xx - yy), C["T4D_SH_C3_3"] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy),
              C["T4D_SH_C3_4"] * x * (4.0 * zz - xx - yy), C["T4D_SH_C3_5"] * z * (xx - yy),
              C["T4D_SH_C3_6"] * x * (xx - 3.0 * yy)]
    return torch.stack(b, dim=1)


def quat_to_rot(q: torch.Tensor):
    """Rotation from (r,x,y,z), NOT normalised here; same matrix as external.py:26-43 build_rotation."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def preprocess(view: View, means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
               cov3D_precomp, dtype):
    """Appendix A.1.  Returns a dict of per-Gaussian screen-space quantities (differentiable)."""
    P = means3D.shape[0]
    H, W = int(view.image_height), int(view.image_width)
    VT = view.viewmatrix.reshape(4, 4).to(dtype).cpu()
    PT = view.projmatrix.reshape(4, 4).to(dtype).cpu()
    ones = torch.ones(P, 1, dtype=dtype)
    ph = torch.cat([means3D, ones], dim=1)
    p_view = ph @ VT[:, :3]
    p_hom = ph @ PT
    valid = p_view[:, 2] > C["T4D_NEAR_CULL_Z"]

    p_w = 1.0 / (p_hom[:, 3] + C["T4D_HOM_W_EPS"])
    p_proj = p_hom[:, :3] * p_w[:, None]

    if cov3D_precomp is not None and cov3D_precomp.numel() > 0:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4],
                             c[:, 2], c[:, 4], c[:, 5]], dim=1).reshape(P, 3, 3)
    else:
        R = quat_to_rot(rotations)
        S = view.scale_modifier * scales
        RS = R * S[:, None, :]
        Sigma = RS @ RS.transpose(1, 2)

    focal_x = W / (2.0 * view.tanfovx)
    focal_y = H / (2.0 * view.tanfovy)
    tx, ty, tz = p_view[:, 0], p_view[:, 1], p_view[:, 2]
    tz_safe = torch.where(valid, tz, torch.ones_like(tz))
    limx = C["T4D_FRUSTUM_CLAMP"] * view.tanfovx
    limy = C["T4D_FRUSTUM_CLAMP"] * view.tanfovy
    txtz = tx / tz_safe
    tytz = ty / tz_safe
    in_x = (txtz >= -limx) & (txtz <= limx)
    in_y = (tytz >= -limy) & (tytz <= limy)
    txc = torch.where(in_x, tx, (torch.clamp(txtz, -limx, limx) * tz_safe).detach())
    tyc = torch.where(in_y, ty, (torch.clamp(tytz, -limy, limy) * tz_safe).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([focal_x / tz_safe, zero, -(focal_x * txc) / (tz_safe * tz_safe),
                     zero, focal_y / tz_safe, -(focal_y * tyc) / (tz_safe * tz_safe)], dim=1).reshape(P, 2, 3)
    Wm = VT[:3, :3].T  # math rotation (rows of world->view)
    T = J @ Wm
    cov = T @ Sigma @ T.transpose(1, 2)
    a = cov[:, 0, 0] + C["T4D_COV2D_DILATION"]
    b = cov[:, 0, 1]
    c_ = cov[:, 1, 1] + C["T4D_COV2D_DILATION"]
    det = a * c_ - b * b
    valid = valid & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    conic = _ConicOfCov2D.apply(a, b, c_, det_safe.detach())
    mid = 0.5 * (a + c_)
    lam1 = mid + torch.sqrt(torch.clamp(mid * mid - det, min=C["T4D_EIGEN_FLOOR"]))
    lam2 = mid - torch.sqrt(torch.clamp(mid * mid - det, min=C["T4D_EIGEN_FLOOR"]))
    radius = torch.ceil(C["T4D_RADIUS_SIGMAS"] * torch.sqrt(torch.maximum(lam1, lam2))).detach()

    m2 = means2D.to(dtype) if means2D is not None else torch.zeros(P, 3, dtype=dtype)
    px = ((p_proj[:, 0] + m2[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + m2[:, 1] + 1.0) * H - 1.0) * 0.5
    xy = torch.stack([px, py], dim=1)

    gx = (W + TILE - 1) // TILE
    gy = (H + TILE - 1) // TILE
    xyd = xy.detach().numpy().astype(np.float64)
    rad = radius.numpy()
    ok = valid.numpy() & np.isfinite(xyd).all(axis=1) & np.isfinite(rad)
    xyd = np.where(ok[:, None], xyd, 0.0)
    rad = np.where(ok, rad, 0.0)

    def trunc_div(v):  # (int)(float / 16): C cast truncates toward zero
        return np.trunc(v / TILE).astype(np.int64)
    rmin_x = np.clip(trunc_div(xyd[:, 0] - rad), 0, gx)
    rmin_y = np.clip(trunc_div(xyd[:, 1] - rad), 0, gy)
    rmax_x = np.clip(trunc_div(xyd[:, 0] + rad + TILE - 1), 0, gx)
    rmax_y = np.clip(trunc_div(xyd[:, 1] + rad + TILE - 1), 0, gy)
    area = (rmax_x - rmin_x) * (rmax_y - rmin_y)
    ok = ok & (area > 0)

    clamped = None
    if shs is not None and shs.numel() > 0:
        deg = int(view.sh_degree)
        campos = view.campos.reshape(3).to(dtype).cpu()
        d = means3D - campos[None, :]
        d = d / torch.sqrt((d * d).sum(dim=1, keepdim=True))
        basis = sh_basis(deg, d)                       # [P, K]
        K = basis.shape[1]
        rgb_raw = (basis[:, :, None] * shs[:, :K, :]).sum(dim=1) + 0.5
        clamped = (rgb_raw < 0).detach()
        rgb = torch.clamp(rgb_raw, min=0.0)
    else:
        rgb = colors_precomp

    return dict(valid=torch.from_numpy(ok), xy=xy, depth=p_view[:, 2], conic=conic,
                opacity=opacities.reshape(-1), rgb=rgb, radius=radius, clamped=clamped,
                rect=(rmin_x, rmin_y, rmax_x, rmax_y), grid=(gx, gy), cov2d=torch.stack([a, b, c_], 1),
                Sigma=Sigma)


def build_tile_lists(pre, dtype_key=np.float32):
    """Appendix A.2: per tile, Gaussian indices ordered by (float32 depth bits, index)."""
    gx, gy = pre["grid"]
    rmin_x, rmin_y, rmax_x, rmax_y = pre["rect"]
    ok = pre["valid"].numpy()
    depth = pre["depth"].detach().numpy().astype(dtype_key)
    lists = [[] for _ in range(gx * gy)]
    for i in np.nonzero(ok)[0]:
        for ty in range(rmin_y[i], rmax_y[i]):
            for tx in range(rmin_x[i], rmax_x[i]):
                lists[ty * gx + tx].append(i)
    out = []
    for l in lists:
        if l:
            idx = np.asarray(l, dtype=np.int64)
            order = np.lexsort((idx, depth[idx]))
            out.append(idx[order])
        else:
            out.append(np.zeros(0, dtype=np.int64))
    return out


def rasterize(view: View, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
              rotations=None, cov3D_precomp=None, dtype=torch.float64, return_aux=False):
    """Forward render; differentiable w.r.t. every floating-point input tensor.

    Returns (color[3,H,W], radii[P] int32, depth[1,H,W], alpha[1,H,W]) like the call sites at
    train.py:307 unpack, plus an aux dict when return_aux is set.
    """
    cast = lambda t: None if t is None else t.to(dtype)
    means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp = map(
        cast, (means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp))
    H, W = int(view.image_height), int(view.image_width)
    pre = preprocess(view, means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
                     cov3D_precomp, dtype)
    lists = build_tile_lists(pre)
    gx, gy = pre["grid"]
    bg = view.bg.reshape(3).to(dtype).cpu()

    color = torch.zeros(3, H, W, dtype=dtype)
    depth = torch.zeros(1, H, W, dtype=dtype)
    alpha = torch.zeros(1, H, W, dtype=dtype)
    final_T = torch.ones(H, W, dtype=dtype)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    color += bg[:, None, None]  # tiles with an empty list: C = 0 + T(=1)*bg

    a_max, a_min, t_stop = f32(C["T4D_ALPHA_MAX"]), f32(C["T4D_ALPHA_MIN"]), f32(C["T4D_T_STOP"])
    for t, lst in enumerate(lists):
        if len(lst) == 0:
            continue
        ty, tx = divmod(t, gx)
        y0, x0 = ty * TILE, tx * TILE
        y1, x1 = min(y0 + TILE, H), min(x0 + TILE, W)
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        pxf = xs.reshape(-1).to(dtype)
        pyf = ys.reshape(-1).to(dtype)
        idx = torch.from_numpy(lst)
        gxy = pre["xy"][idx]
        con = pre["conic"][idx]
        op = pre["opacity"][idx]
        dx = gxy[None, :, 0] - pxf[:, None]
        dy = gxy[None, :, 1] - pyf[:, None]
        power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
        ok = power <= 0
        G = torch.exp(torch.clamp(power, max=0.0))
        a_raw = op[None, :] * G
        a = a_raw + (torch.clamp(a_raw, max=a_max) - a_raw).detach()   # straight-through clamp
        ok = ok & (a >= a_min)
        a_eff = torch.where(ok, a, torch.zeros_like(a))
        T_incl = torch.cumprod(1.0 - a_eff, dim=1)
        stop = ok & (T_incl < t_stop)
        stopped = torch.cumsum(stop.to(torch.int32), dim=1) > 0
        contrib = ok & ~stopped
        a_c = torch.where(contrib, a, torch.zeros_like(a))
        T_after = torch.cumprod(1.0 - a_c, dim=1)
        T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
        w = a_c * T_before
        rgb = pre["rgb"][idx]
        dep = pre["depth"][idx]
        Cpix = w @ rgb                      # [npix,3]
        Dpix = w @ dep
        Apix = w.sum(dim=1)
        Tfin = T_after[:, -1]
        ar = torch.arange(1, len(lst) + 1, dtype=torch.int32)
        ncon = (contrib.to(torch.int32) * ar[None, :]).max(dim=1).values
        hh, ww = y1 - y0, x1 - x0
        color[:, y0:y1, x0:x1] = (Cpix + Tfin[:, None] * bg[None, :]).T.reshape(3, hh, ww)
        depth[0, y0:y1, x0:x1] = Dpix.reshape(hh, ww)
        alpha[0, y0:y1, x0:x1] = Apix.reshape(hh, ww)
        final_T[y0:y1, x0:x1] = Tfin.detach().reshape(hh, ww)
        n_contrib[y0:y1, x0:x1] = ncon.reshape(hh, ww)

    radii = torch.where(pre["valid"], pre["radius"], torch.zeros_like(pre["radius"])).to(torch.int32)
    if return_aux:
        aux = dict(pre=pre, lists=lists, final_T=final_T, n_contrib=n_contrib)
        return color, radii, depth, alpha, aux
    return color, radii, depth, alpha


def rasterize_with_grads(view: View, means3D, opacities, scales, rotations, colors_precomp=None,
                         shs=None, cov3D_precomp=None, dL_dcolor=None, dL_ddepth=None, dL_dalpha=None,
                         dtype=torch.float64):
    """Convenience: forward + backward against given output cotangents.  Returns (outputs, grads)."""
    def leaf(t):
        return None if t is None else t.detach().to(dtype).clone().requires_grad_(True)
    m3, op, sc, ro, cp, sh, cv = map(leaf, (means3D, opacities, scales, rotations, colors_precomp, shs,
                                            cov3D_precomp))
    m2 = torch.zeros(m3.shape[0], 3, dtype=dtype, requires_grad=True)
    color, radii, depth, alpha, aux = rasterize(view, m3, m2, op, sh, cp, sc, ro, cv, dtype=dtype,
                                                return_aux=True)
    loss = 0.0
    if dL_dcolor is not None:
        loss = loss + (color * dL_dcolor.to(dtype)).sum()
    if dL_ddepth is not None:
        loss = loss + (depth * dL_ddepth.to(dtype)).sum()
    if dL_dalpha is not None:
        loss = loss + (alpha * dL_dalpha.to(dtype)).sum()
    names = ["means3D", "means2D", "opacities", "scales", "rotations", "colors_precomp", "shs", "cov3D_precomp"]
    leaves = [m3, m2, op, sc, ro, cp, sh, cv]
    have = [(n, l) for n, l in zip(names, leaves) if l is not None]
    gs = torch.autograd.grad(loss, [l for _, l in have], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(l)) for (n, l), g in zip(have, gs)}
    outs = dict(color=color.detach(), radii=radii, depth=depth.detach(), alpha=alpha.detach(),
                final_T=aux["final_T"], n_contrib=aux["n_contrib"], aux=aux)
    return outs, grads
