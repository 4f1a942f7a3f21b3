"""
Texture bake on the GPU — host-side mirror of the reference interface for BASELINE config 5.

Mirrors, name for name:
  process_uv(uv_coords, uv_h, uv_w)                       helpers.py:945-950
  render_colors(vertices, triangles, colors, h, w, c, BG) face3d/mesh/render.py:52-86   (-> _render_colors_core)
  write_texture(path, uvs, colors, faces, res)            helpers.py:953-960
over `t4d_texture_render_colors` (include/topo4d_raster.h).  Results are bit-identical to the reference's CPU code
(tests/test_gpu_texture.py compares with the reference's own source compiled into oracle/_ref).
There is no CPU path here either.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import T4D_ERR_PAIR_OVERFLOW, T4D_OK

_CAP = {}


def process_uv(uv_coords, uv_h: int = 256, uv_w: int = 256):
    """helpers.py:945-950: u*(w-1), flip v, append z = 0.  (The reference mutates its argument in place and then
    returns a new hstack'ed array; this mirror leaves the argument untouched.)"""
    uv = np.array(uv_coords, dtype=np.float64, copy=True)
    uv[:, 0] = uv[:, 0] * (uv_w - 1)
    uv[:, 1] = uv[:, 1] * (uv_h - 1)
    uv[:, 1] = uv_h - uv[:, 1] - 1
    return np.hstack((uv, np.zeros((uv.shape[0], 1))))


def _dev(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(device).contiguous()


def render_colors(vertices, triangles, colors, h: int, w: int, c: int = 3, BG=None, rows: Optional[Tuple[int, int]] = None,
                  device="cuda", return_depth: bool = False):
    """face3d/mesh/render.py:52-86 on the GPU.  vertices [nver,3] (pixel x, pixel y, depth), triangles [ntri,3],
    colors [nver,c]; returns the image [h,w,c] (float32, on `device`).  `rows=(begin,end)` bakes only that band
    (everything else keeps BG / zeros) — the unit a multi-GPU bake shards."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("topo4d_amd has no CPU path: the texture bake needs a HIP device")
    lib = _lib.load()
    v = _dev(vertices, torch.float32, device)
    t = _dev(triangles, torch.int32, device)
    col = _dev(colors, torch.float32, device)
    if v.dim() != 2 or v.shape[1] != 3 or t.dim() != 2 or t.shape[1] != 3 or col.shape != (v.shape[0], c):
        raise ValueError("vertices [nver,3], triangles [ntri,3], colors [nver,c] expected")
    r0, r1 = (0, h) if rows is None else (int(rows[0]), int(rows[1]))
    bg = None if BG is None else _dev(BG, torch.float32, device)
    if bg is not None:
        assert bg.shape == (h, w, c)
    if (r0, r1) == (0, h):
        # the whole image: the kernel writes every texel (winner or background) - nothing to fill first
        image = torch.empty(h, w, c, dtype=torch.float32, device=device)
        depth = torch.empty(h, w, dtype=torch.float32, device=device)
    else:
        # a band: the rows outside it keep the background / the initial depth (render.py:72)
        image = torch.zeros(h, w, c, dtype=torch.float32, device=device) if bg is None else bg.clone()
        depth = torch.full((h, w), -999999.0, dtype=torch.float32, device=device)
    key = (device.index, int(t.shape[0]), h, w)
    cap = _CAP.get(key, max(65536, 4 * int(t.shape[0])))
    need = C.c_int64(0)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    for _ in range(4):
        nbytes = lib.t4d_texture_bake_scratch_bytes(h, w, cap)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
        rc = lib.t4d_texture_render_colors(C.c_void_p(v.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(col.data_ptr()),
                                           None if bg is None else C.c_void_p(bg.data_ptr()),
                                           int(v.shape[0]), int(t.shape[0]), h, w, c, r0, r1, C.c_void_p(image.data_ptr()),
                                           C.c_void_p(depth.data_ptr()), C.c_void_p(scratch.data_ptr()), nbytes, cap,
                                           C.byref(need), stream)
        if rc == T4D_OK:
            break
        if rc == T4D_ERR_PAIR_OVERFLOW:
            cap = int(need.value * 1.25) + 1024
            continue
        raise RuntimeError(f"t4d_texture_render_colors failed (code {rc}): {_lib.last_error()}")
    else:
        raise RuntimeError("texture bake: pair capacity kept overflowing")
    _CAP[key] = cap
    return (image, depth) if return_depth else image


def bake_texture(uvs, colors, faces, res: int = 1024, device="cuda") -> np.ndarray:
    """helpers.py:953-959 without the file write: uint8 [res,res,3], byte-identical to what the reference saves."""
    uv_coords = process_uv(uvs, res, res)
    tex = render_colors(uv_coords, faces, colors, res, res, c=3, device=device)
    return (tex.cpu().numpy() * 255).astype(np.uint8)          # same numpy cast as helpers.py:959


def bake_texture_sharded(uvs, colors, faces, res: int = 1024, device="cuda") -> np.ndarray:
    """bake_texture with the rows of the UV image split over the ranks of the default process group (one process per GPU):
    every rank bakes its own band (`render_colors(rows=...)`, the triangles that cannot touch the band are rejected by
    the binning kernel), the bands are all-gathered, every rank returns the full uint8 image.  Byte-identical to
    bake_texture: the per-texel result does not depend on the band it is computed in."""
    import torch.distributed as dist
    from . import dist as t4d_dist
    if not dist.is_available() or not dist.is_initialized():
        return bake_texture(uvs, colors, faces, res, device)
    r0, r1 = t4d_dist.band_bounds(res, dist.get_rank(), dist.get_world_size())
    uv_coords = process_uv(uvs, res, res)
    tex = render_colors(uv_coords, faces, colors, res, res, c=3, rows=(r0, r1), device=device)
    full = t4d_dist.gather_bands(tex[r0:r1].contiguous(), res)
    return (full.cpu().numpy() * 255).astype(np.uint8)


def write_texture(path, uvs, colors, faces, res: int = 1024, device="cuda") -> None:
    """helpers.py:953-960 (`io.imsave` replaced by PIL, which this image has)."""
    from PIL import Image
    Image.fromarray(np.squeeze(bake_texture(uvs, colors, faces, res, device))).save(path)


def compute_vertex_attribute_by_weight(variables, attribute: torch.Tensor) -> torch.Tensor:
    """helpers.py:237-253 `compute_vertex_attribute_by_weight_2` on the device: `attribute` [n_coarse, d] (float32, GPU) ->
    [n_dense_total, d] float32, without the per-frame device->host->device round trip of train.py:504-506.
    `variables` holds the same keys the reference uses: 'dense_vertex_father' [n_dense(,1)], 'dense_vertex_weight'
    [n_dense,4], 'dense_quad_faces' [n_quads,4], 'dense_vertex' (only its length is used).  The index/weight arrays are
    uploaded once and cached on the dict."""
    if not attribute.is_cuda:
        raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
    lib = _lib.load()
    dev = attribute.device
    cache = variables.setdefault("_t4d_dense_cache", {})
    if cache.get("device") != dev:
        cache["father"] = torch.as_tensor(np.asarray(variables["dense_vertex_father"]).reshape(-1), dtype=torch.int32).to(dev)
        cache["weight"] = torch.as_tensor(np.asarray(variables["dense_vertex_weight"], dtype=np.float64)).to(dev).contiguous()
        cache["quads"] = torch.as_tensor(np.asarray(variables["dense_quad_faces"]), dtype=torch.int32).to(dev).contiguous()
        cache["device"] = dev
    attr = attribute.detach().float().contiguous()
    n_coarse, width = int(attr.shape[0]), int(attr.shape[1])
    n_total = int(np.asarray(variables["dense_vertex"]).shape[0])
    n_dense = n_total - n_coarse
    assert n_dense == cache["father"].numel() == cache["weight"].shape[0]
    out = torch.empty(n_total, width, dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.t4d_dense_interpolate(p(attr), p(cache["quads"]), p(cache["father"]), p(cache["weight"]), n_coarse, n_dense, width,
                                   p(out), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != T4D_OK:
        raise RuntimeError(f"t4d_dense_interpolate failed (code {rc}): {_lib.last_error()}")
    return out
