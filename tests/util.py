"""Shared helpers for the parity tests: scenes, the HIP path through the C ABI, the oracles, state decoding."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as TO
from scaffold import reference_boundary as boundary, scene

GRAD_KEYS = ("means3D", "means2D", "opacities", "scales", "rotations", "colors_precomp")


def make_scene(n_lat, n_lon, H, W, V, opacity="B", sh_degree=None, seed=0, true_campos=False, bg=None):
    params = scene.make_gaussians(n_lat, n_lon, opacity=opacity, sh_degree=sh_degree, seed=seed)
    rv = boundary.params2rendervar(params)
    rv = {k: v.detach() for k, v in rv.items()}
    if sh_degree is not None:
        rv["shs"] = params["shs"]
        del rv["colors_precomp"]
    cams = scene.camera_rig(H, W, n_views=V, true_campos=true_campos or sh_degree is not None)
    if sh_degree is not None:
        cams = [c._replace(sh_degree=sh_degree) for c in cams]
    if bg is not None:
        cams = [c._replace(bg=torch.tensor(bg, dtype=torch.float32)) for c in cams]
    return rv, cams


def to_device(cams, dev):
    out = []
    for c in cams:
        out.append(c._replace(bg=c.bg.to(dev), viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev),
                              campos=c.campos.to(dev)))
    return out


def hip_render(cams, rv, dc=None, dd=None, da=None, dev="cuda"):
    """Forward (+backward when dc is given) through ViewBatch == through the C ABI.  Returns numpy dicts."""
    from topo4d_amd import ViewBatch, pack_views
    from topo4d_amd.rasterizer import _check_common
    H, W, smod, deg = _check_common(cams)
    dcams = to_device(cams, dev)
    views = pack_views(dcams, torch.device(dev))
    batch = ViewBatch(views, H, W, smod, deg)
    d = lambda k: rv[k].to(dev) if k in rv and rv[k] is not None else None
    color, radii, depth, alpha = batch.forward(d("means3D"), d("opacities"), d("scales"), d("rotations"),
                                               d("colors_precomp"), d("shs"), d("cov3D_precomp"))
    out = dict(color=color.cpu().numpy(), radii=radii.cpu().numpy(), depth=depth.cpu().numpy(),
               alpha=alpha.cpu().numpy())
    grads = None
    if dc is not None:
        g = batch.backward(dc.to(dev), None if dd is None else dd.to(dev), None if da is None else da.to(dev))
        grads = {k: (v.cpu().numpy() if v is not None else None) for k, v in g.items()}
    return out, grads, batch


def c_oracle_render(cam, rv, dc=None, dd=None, da=None):
    r = CO.OracleRender(cam, rv["means3D"], rv["opacities"], rv.get("scales"), rv.get("rotations"),
                        rv.get("colors_precomp"), rv.get("shs"), rv.get("cov3D_precomp"))
    g = None
    if dc is not None:
        g = r.backward(dc, dd, da)
    return r, g


def c_oracle_render_many(cams, rv, dcs=None, dds=None, das=None, threads=None):
    """c_oracle_render for several views at once: the C oracle is called through ctypes, which releases the GIL, so a thread per
    view uses the host's cores.  Yields (OracleRender, grads) in view order."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = len(cams)
    pick = lambda xs, i: None if xs is None else xs[i]
    with ThreadPoolExecutor(max_workers=threads or min(n, max(1, (os.cpu_count() or 2) // 2))) as ex:
        futs = [ex.submit(c_oracle_render, cams[i], rv, pick(dcs, i), pick(dds, i), pick(das, i)) for i in range(n)]
        for f in futs:
            yield f.result()


def torch_oracle_render(cam, rv, dc=None, dd=None, da=None, dtype=torch.float64):
    view = TO.View(*cam)
    outs, grads = TO.rasterize_with_grads(view, rv["means3D"], rv["opacities"], rv.get("scales"), rv.get("rotations"),
                                          colors_precomp=rv.get("colors_precomp"), shs=rv.get("shs"),
                                          cov3D_precomp=rv.get("cov3D_precomp"), dL_dcolor=dc, dL_ddepth=dd,
                                          dL_dalpha=da, dtype=dtype)
    return outs, grads


def decode_state(batch):
    """Integer/float state of a forward, sliced out of the opaque buffer via t4d_debug_state_layout."""
    from topo4d_amd import _lib
    lib = _lib.load()
    prob = batch.prob
    offs = (C.c_uint64 * 16)()
    has_sh = batch.prob.sh_coeffs > 0
    rc = lib.t4d_debug_state_layout(C.byref(prob), int(has_sh), offs, 16)
    assert rc == 0
    names = ["status", "view_total", "view_cursor", "tile_count", "bucket_fill", "tile_off", "xy", "depth",
             "conic_opacity", "rgb", "clamped", "pair_off", "keys", "final_T", "n_contrib", "total"]
    o = dict(zip(names, [int(x) for x in offs]))
    raw = batch.state.cpu().numpy()
    V, P, H, W = prob.n_views, prob.P, prob.H, prob.W
    T = ((W + 15) // 16) * ((H + 15) // 16)
    cap = prob.pair_capacity
    f = lambda name, n, dt: raw[o[name]: o[name] + n * np.dtype(dt).itemsize].view(dt)
    tile_count = f("tile_count", V * T, np.uint32).reshape(V, T)
    # The forward writes no replay state (final_T, n_contrib) for EMPTY tiles - the backward never visits them - so those
    # pixels hold whatever the allocation held.  They are presented here with the values the state means there: T = 1, no
    # contributor (what the oracle holds for a background pixel).
    gx, gy = (W + 15) // 16, (H + 15) // 16
    empty = np.repeat(np.repeat((tile_count == 0).reshape(V, gy, gx), 16, axis=1), 16, axis=2)[:, :H, :W]
    final_T = np.where(empty, np.float32(1.0), f("final_T", V * H * W, np.float32).reshape(V, H, W))
    n_contrib = np.where(empty, np.uint32(0), f("n_contrib", V * H * W, np.uint32).reshape(V, H, W))
    return dict(
        status=f("status", 4, np.uint32), view_total=f("view_total", V, np.uint32),
        tile_count=tile_count, bucket_fill=f("bucket_fill", 24, np.uint32),
        tile_off=f("tile_off", V * T, np.uint32).reshape(V, T),
        xy=f("xy", V * P * 2, np.float32).reshape(V, P, 2), depth=f("depth", V * P, np.float32).reshape(V, P),
        conic_opacity=f("conic_opacity", V * P * 4, np.float32).reshape(V, P, 4),
        pair_off=f("pair_off", V * P, np.uint32).reshape(V, P),
        keys=f("keys", V * cap, np.uint64).reshape(V, cap),
        final_T=final_T, n_contrib=n_contrib, T=T, cap=cap)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
