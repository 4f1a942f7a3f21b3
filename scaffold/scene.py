"""
Seeded synthetic scenes for the rasterizer hot path (BASELINE.json configs 1-4).

There is no dataset on the GPU box, so bench.py, smoke() and the parity tests render a head-sized
ellipsoid of vertex-bound Gaussians from a 24-camera rig.  The recipe mirrors how Topo4D initialises
its Gaussians (reference train.py:132-146): isotropic scale = half the nearest-neighbour distance,
rotation = quaternion turning +x onto the vertex normal (external.py:45-61), logit opacity 1000,
free RGB colours.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

from .reference_boundary import setup_camera

# BASELINE.json configs -> (n_lat, n_lon, H, W, sh_degree)
CONFIGS = {
    "C1": dict(n_lat=50, n_lon=100, H=256, W=256, n_views=1, sh_degree=None),
    "C2": dict(n_lat=150, n_lon=200, H=512, W=512, n_views=24, sh_degree=None),
    "C4": dict(n_lat=300, n_lon=400, H=2048, W=2048, n_views=24, sh_degree=3),
}

SEMI_AXES = (0.09, 0.12, 0.10)   # metres: head-sized (cf. eye scales 0.0025-0.01 in train.py:625-628)
SH_C0 = 0.28209479177387814       # helpers.py:836


def _nearest_neighbour_sq_dist(pts: np.ndarray) -> np.ndarray:
    from scipy.spatial import cKDTree
    d, _ = cKDTree(pts).query(pts, k=2)
    return d[:, 1] ** 2


def quaternion_from_normals(normals: np.ndarray) -> np.ndarray:
    """Restates external.py:45-61 `build_quaterion`: (cos(a/2), cross(+x, n)*sin(a/2)), NOT normalised."""
    n = normals / np.linalg.norm(normals, axis=1, keepdims=True)
    xaxis = np.zeros_like(n)
    xaxis[:, 0] = 1.0
    axes = np.cross(xaxis, n)
    ang = np.arccos(np.clip((xaxis * n).sum(1), -1.0, 1.0))
    return np.concatenate([np.cos(ang / 2)[:, None], axes * np.sin(ang / 2)[:, None]], axis=1)


def make_gaussians(n_lat: int, n_lon: int, opacity: str = "A", sh_degree=None, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Optimiser-style parameter dict (same keys as train.py:138-146) on the CPU, float32.

    opacity "A": logit 1000 everywhere (train.py:142 -> sigmoid == 1.0, alpha saturates at 0.99);
    opacity "B": opacities uniform(0.05, 0.95) (exercises unsaturated blending).
    """
    rng = np.random.default_rng(seed)
    ax, ay, az = SEMI_AXES
    # latitudes strictly inside (-pi/2, pi/2): no coincident pole vertices
    lat = (np.arange(n_lat) + 0.5) / n_lat * math.pi - math.pi / 2
    lon = np.arange(n_lon) / n_lon * 2 * math.pi
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    unit = np.stack([np.cos(la) * np.sin(lo), np.sin(la), np.cos(la) * np.cos(lo)], axis=-1).reshape(-1, 3)
    pts = unit * np.array([ax, ay, az])
    spacing = math.pi * ay / n_lat
    pts = pts + rng.normal(0.0, 0.1 * spacing, size=pts.shape)
    normals = unit / np.array([ax, ay, az])
    P = pts.shape[0]

    sq = np.clip(_nearest_neighbour_sq_dist(pts), 1e-7, None)        # train.py:133-134
    log_scales = np.tile(np.log(np.sqrt(sq) / 2)[:, None], (1, 3))    # train.py:143
    rot = quaternion_from_normals(normals)                            # train.py:136-137
    if opacity == "A":
        logit = np.ones((P, 1)) * 1000.0                              # train.py:142
    elif opacity == "B":
        o = rng.uniform(0.05, 0.95, size=(P, 1))
        logit = np.log(o / (1 - o))
    else:
        raise ValueError("opacity scenario must be 'A' or 'B'")
    rgb = rng.uniform(0.0, 1.0, size=(P, 3))
    params = {
        "means3D": pts, "rgb_colors": rgb, "unnorm_rotations": rot,
        "logit_opacities": logit, "log_scales": log_scales,
    }
    if sh_degree is not None:
        M = (sh_degree + 1) ** 2
        sh = rng.normal(0.0, 0.05, size=(P, M, 3))
        sh[:, 0, :] = (rgb - 0.5) / SH_C0
        params["shs"] = sh
    return {k: torch.tensor(v).float().contiguous() for k, v in params.items()}


def camera_rig(H: int, W: int, n_views: int = 24, device="cpu", true_campos=False, distance: float = 0.9):
    """24 = 3 elevations (-20,0,+20 deg) x 8 azimuths in [-70,70] deg, looking at the origin.

    fx = fy chosen so the 0.24 m tall head spans ~70 % of the image height.  Returns a list of
    GaussianRasterizationSettings built by the G1-pinned setup_camera mirror.
    """
    elevs = [-20.0, 0.0, 20.0]
    n_az = max(1, (n_views + len(elevs) - 1) // len(elevs))
    azims = np.linspace(-70.0, 70.0, n_az) if n_az > 1 else np.array([0.0])
    f = 0.7 * H * distance / (2 * SEMI_AXES[1])
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    cams = []
    for el in elevs:
        for az in azims:
            if len(cams) == n_views:
                break
            e, a = math.radians(el), math.radians(az)
            c = distance * np.array([math.sin(a) * math.cos(e), math.sin(e), math.cos(a) * math.cos(e)])
            fwd = -c / np.linalg.norm(c)
            right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
            right /= np.linalg.norm(right)
            down = np.cross(fwd, right)
            R = np.stack([right, down, fwd], axis=0)
            w2c = np.eye(4)
            w2c[:3, :3] = R
            w2c[:3, 3] = -R @ c
            cams.append(setup_camera(W, H, K, w2c.astype(np.float32), near=0.01, far=100, device=device,
                                     true_campos=true_campos))
    if n_views == 1:
        return cams[len(cams) // 2: len(cams) // 2 + 1] if len(cams) > 1 else cams
    return cams


def output_cotangents(n_views: int, H: int, W: int, seed: int = 0, depth_alpha: bool = False):
    """dL/dcolor = seeded N(0,1)/(3HW) per view (the 'pure rasterizer' metric of SURVEY §8d)."""
    g = torch.Generator().manual_seed(seed)
    dc = torch.randn(n_views, 3, H, W, generator=g) / (3 * H * W)
    if not depth_alpha:
        return dc, None, None
    dd = torch.randn(n_views, 1, H, W, generator=g) / (H * W)
    da = torch.randn(n_views, 1, H, W, generator=g) / (H * W)
    return dc, dd, da


def frame_displacement(means3D: torch.Tensor, t: int, n_frames: int = 64, seed: int = 0) -> torch.Tensor:
    """BASELINE config 3: per-frame vertex motion 0.002*sin(2*pi*t/64 + phi_v)."""
    g = torch.Generator().manual_seed(seed + 12345)
    phi = torch.rand(means3D.shape, generator=g) * (2 * math.pi)
    return means3D + 0.002 * torch.sin(2 * math.pi * t / n_frames + phi)


# ---- face-parsing label images (the `mask` entry of get_dataset, train.py:84-92) ------------------------------------
PARSING_LABELS = ("background", "skin", "l_eyebrow", "r_eyebrow", "l_eye", "r_eye", "nose", "upper_lip", "inner_mouth",
                  "lower_lip", "hair", "l_ear", "r_ear", "glasses")          # train.py:50-55 `cmap_index`, by index


def parsing_colormap_bgr(n_label: int = 14) -> np.ndarray:
    """uint8 [n_label,3]: the colour of every parsing label in the channel order of the mask images - what helpers.py:806
    builds (`label_colormap(14)[:, [2, 1, 0]]`: the pascal-VOC bit-interleaved colormap, helpers.py:783-797, with its columns
    reversed).  Bits 0/1/2 of (id >> 3j) go to bit 7-j of r/g/b.  Pinned by golden G9's `label_colors`."""
    cmap = np.zeros((n_label, 3), dtype=np.uint8)
    for label in range(n_label):
        rgb = [0, 0, 0]
        for j in range(8):
            chunk = label >> (3 * j)
            for c in range(3):
                rgb[c] |= ((chunk >> c) & 1) << (7 - j)
        cmap[label] = rgb[::-1]
    return cmap


def make_label_image(H: int, W: int, seed: int = 0) -> torch.Tensor:
    """A synthetic face-parsing image [3,H,W] float32 = label colours / 255 (as get_dataset loads a mask PNG): skin ellipse, lips
    and an inner-mouth ellipse, eyes, hair band."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    u, v = (xx - W / 2) / (W / 2), (yy - H / 2) / (H / 2)
    jx, jy = rng.uniform(-0.05, 0.05, 2)
    lab = np.zeros((H, W), np.int64)
    lab[(u / 0.7) ** 2 + (v / 0.85) ** 2 < 1] = 1
    lab[(v < -0.55) & ((u / 0.75) ** 2 + (v / 0.9) ** 2 < 1)] = 10
    lab[((u + 0.3 - jx) / 0.12) ** 2 + ((v + 0.2) / 0.06) ** 2 < 1] = 4
    lab[((u - 0.3 - jx) / 0.12) ** 2 + ((v + 0.2) / 0.06) ** 2 < 1] = 5
    lab[((u - jx) / 0.1) ** 2 + ((v - 0.05) / 0.15) ** 2 < 1] = 6
    mouth = ((u - jx) / 0.3) ** 2 + ((v - 0.45 - jy) / 0.12) ** 2
    lab[(mouth < 1) & (v < 0.45 + jy)] = 7
    lab[(mouth < 1) & (v >= 0.45 + jy)] = 9
    lab[((u - jx) / 0.22) ** 2 + ((v - 0.45 - jy) / 0.06) ** 2 < 1] = 8
    cmap = parsing_colormap_bgr(14)
    img = cmap[lab]                                                         # [H,W,3] uint8
    return torch.tensor(img / 255.0).float().permute(2, 0, 1).contiguous()


def make_dense_params(params: Dict[str, torch.Tensor], per_vertex: int = 4, seed: int = 0):
    """A dense (texture-pass) Gaussian set around a coarse one, shaped like train.py:240-266: `dense_means3D` (not trainable)
    = the coarse vertices plus `per_vertex - 1` jittered copies each, `dense_rgb_colors` interpolated colours,
    opacity 0.9999, isotropic scales from the dense spacing, identity rotations.  Returns (dense params, dense_init_colors)."""
    g = torch.Generator().manual_seed(seed + 77)
    m, c, s = params["means3D"], params["rgb_colors"], params["log_scales"]
    P = m.shape[0]
    reps = [m] + [m + torch.randn(P, 3, generator=g) * torch.exp(s).mean(1, keepdim=True) * 0.7 for _ in range(per_vertex - 1)]
    cols = [c] + [(c + torch.randn(P, 3, generator=g) * 0.05).clamp(0, 1) for _ in range(per_vertex - 1)]
    n = P * per_vertex
    dense = {
        "dense_means3D": torch.cat(reps).contiguous(),
        "dense_rgb_colors": torch.cat(cols).contiguous(),
        "dense_logit_opacities": torch.full((n, 1), math.log(0.9999 / (1 - 0.9999))),
        "dense_log_scales": (torch.cat([s] * per_vertex) - math.log(per_vertex) / 2).contiguous(),
        "dense_unnorm_rotations": torch.tensor([1.0, 0.0, 0.0, 0.0]).repeat(n, 1),
    }
    return dense, dense["dense_rgb_colors"].clone()


# ---- UV-space mesh of the texture bake (config 5) ---------------------------------------------------------------------
def uv_mesh(n, h, w, seed=0, with_depth=False):
    """A jittered n x n UV grid as a triangle soup in pixel space, triangles shuffled (BASELINE config 5: the synthetic mesh of the
    texture bake - tests, bench.py and tools/bench_bake.py share it).  Returns (vertices [n*n,3] float32 (x, y, depth), triangles
    [2(n-1)^2,3] int32, colours [n*n,3] float32)."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.linspace(0.01, 0.99, n), np.linspace(0.01, 0.99, n), indexing="xy")
    uv = np.stack([u.ravel(), v.ravel()], 1) + rng.normal(0, 0.2 / n, size=(n * n, 2))
    idx = np.arange(n * n).reshape(n, n)
    a, b, c_, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([a, b, c_], 1), np.stack([b, d, c_], 1)]).astype(np.int32)
    rng.shuffle(tris)
    z = rng.normal(0, 1, n * n) if with_depth else np.zeros(n * n)
    verts = np.stack([uv[:, 0] * (w - 1), h - uv[:, 1] * (h - 1) - 1, z], 1).astype(np.float32)
    colors = rng.uniform(0, 1, size=(n * n, 3)).astype(np.float32)
    return verts, tris, colors
