"""CPU: the two oracles against each other, against finite differences, and against the committed G6 fixture.

oracle/raster_oracle.c (fp32, explicit backward formulas) must agree with torch.autograd over oracle/torch_oracle.py
(float64) — that is what validates every backward formula the HIP kernels re-use."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as TO
from tests import util
from scaffold import scene

G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("means3D", "means2D", "opacities", "scales", "rotations", "colors_precomp")


def _cmp_grads(gc, gt, keys, rel=1e-4):
    for k in keys:
        a = np.asarray(gc[k], np.float64)
        b = gt[k].numpy().reshape(a.shape)
        scale = max(np.abs(b).max(), 1e-30)
        assert np.abs(a - b).max() <= rel * scale + 1e-10, (k, np.abs(a - b).max(), scale)


@pytest.mark.parametrize("opacity", ["A", "B"])
def test_c_oracle_matches_autograd_f64(opacity):
    H = W = 64
    rv, cams = util.make_scene(12, 20, H, W, 3, opacity=opacity, seed=1)
    dc, dd, da = scene.output_cotangents(1, H, W, seed=3, depth_alpha=True)
    for cam in cams[:2]:
        r, g = util.c_oracle_render(cam, rv, dc[0], dd[0], da[0])
        outs, grads = util.torch_oracle_render(cam, rv, dc[0], dd[0], da[0])
        assert np.abs(outs["color"].numpy() - r.color).max() < 5e-6
        assert np.abs(outs["depth"].numpy() - r.depth).max() < 5e-6
        assert np.abs(outs["alpha"].numpy() - r.alpha).max() < 5e-6
        np.testing.assert_array_equal(outs["radii"].numpy(), r.radii)
        assert (outs["n_contrib"].numpy() == r.state()["n_contrib"]).mean() > 0.999
        _cmp_grads(g, grads, KEYS)


def test_c_oracle_sh_and_cov3d_paths():
    H = W = 48
    rv, cams = util.make_scene(10, 16, H, W, 2, opacity="B", sh_degree=3, seed=2)
    rv["shs"][::5, 0, :] = -3.0
    dc, dd, da = scene.output_cotangents(1, H, W, seed=4, depth_alpha=True)
    r, g = util.c_oracle_render(cams[0], rv, dc[0], dd[0], da[0])
    outs, grads = util.torch_oracle_render(cams[0], rv, dc[0], dd[0], da[0])
    assert np.abs(outs["color"].numpy() - r.color).max() < 5e-6
    _cmp_grads(g, grads, ("means3D", "means2D", "opacities", "scales", "rotations", "shs"))
    # precomputed 3D covariance instead of scale/rotation
    rv2, cams2 = util.make_scene(10, 16, H, W, 2, opacity="B", seed=5)
    R = TO.quat_to_rot(rv2["rotations"].double())
    RS = R * rv2["scales"].double()[:, None, :]
    S = RS @ RS.transpose(1, 2)
    rv2["cov3D_precomp"] = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).float()
    del rv2["scales"], rv2["rotations"]
    r, g = util.c_oracle_render(cams2[0], rv2, dc[0])
    outs, grads = util.torch_oracle_render(cams2[0], rv2, dc[0])
    assert np.abs(outs["color"].numpy() - r.color).max() < 5e-6
    _cmp_grads(g, grads, ("means3D", "means2D", "opacities", "colors_precomp", "cov3D_precomp"))


def test_autograd_oracle_against_finite_differences():
    """Central differences of the float64 forward on a few coordinates of every input (smooth region: opacity B)."""
    H = W = 32
    rv, cams = util.make_scene(6, 10, H, W, 1, opacity="B", seed=8)
    cam = cams[0]
    dc, dd, da = scene.output_cotangents(1, H, W, seed=9, depth_alpha=True)
    view = TO.View(*cam)

    def loss_of(over):
        args = {k: v.double() for k, v in rv.items() if k != "means2D"}
        args.update(over)
        c, _, d, a = TO.rasterize(view, args["means3D"], None, args["opacities"], None, args["colors_precomp"],
                                  args["scales"], args["rotations"], None, dtype=torch.float64)
        return float((c * dc[0].double()).sum() + (d * dd[0].double()).sum() + (a * da[0].double()).sum())

    _, grads = util.torch_oracle_render(cam, rv, dc[0], dd[0], da[0])
    rng = np.random.default_rng(0)
    for k, eps in (("means3D", 1e-6), ("opacities", 1e-6), ("scales", 1e-7), ("rotations", 1e-6), ("colors_precomp", 1e-6)):
        base = rv[k].double()
        g = grads[k].reshape(base.shape)
        order = torch.argsort(g.abs().flatten(), descending=True)[:3].tolist()
        for flat in order:
            idx = np.unravel_index(flat, base.shape)
            p = base.clone(); p[idx] += eps
            m = base.clone(); m[idx] -= eps
            fd = (loss_of({k: p}) - loss_of({k: m})) / (2 * eps)
            assert abs(fd - g[idx].item()) <= 2e-3 * max(abs(fd), abs(g[idx].item())) + 1e-9, (k, idx, fd, g[idx].item())


def test_g6_fixture_pins_both_oracles():
    g = np.load(os.path.join(G, "g6_self_oracle_f64.npz"))
    from scaffold import reference_boundary as boundary
    p = scene.make_gaussians(10, 20, opacity="B", seed=6)
    rv = {k: v.detach() for k, v in boundary.params2rendervar(p).items()}
    cam = scene.camera_rig(64, 64, n_views=3)[1]
    dc, dd, da = scene.output_cotangents(1, 64, 64, seed=7, depth_alpha=True)
    outs, grads = util.torch_oracle_render(cam, rv, dc[0], dd[0], da[0])
    np.testing.assert_allclose(outs["color"].numpy(), g["color"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(outs["radii"].numpy(), g["radii"])
    for k in KEYS:
        np.testing.assert_allclose(grads[k].numpy(), g[f"grad_{k}"], rtol=1e-9, atol=1e-14)
    r, gc = util.c_oracle_render(cam, rv, dc[0], dd[0], da[0])
    assert np.abs(r.color - g["color"]).max() < 5e-6
    np.testing.assert_array_equal(r.radii, g["radii"])
    for k in KEYS:
        b = g[f"grad_{k}"]
        assert np.abs(np.asarray(gc[k], np.float64).reshape(b.shape) - b).max() <= 1e-4 * np.abs(b).max() + 1e-10


def test_mark_visible_and_empty_scene():
    rv, cams = util.make_scene(6, 10, 32, 32, 1, seed=1)
    vm = cams[0].viewmatrix.numpy().reshape(16)
    vis = CO.mark_visible(rv["means3D"].numpy(), vm)
    assert vis.all()
    far = rv["means3D"].numpy() + np.array([0, 0, 50.0], np.float32)
    assert not CO.mark_visible(far, vm).any()
    rv2 = dict(rv)
    rv2["means3D"] = torch.tensor(far)
    r, _ = util.c_oracle_render(cams[0], rv2)
    assert r.num_rendered == 0 and (r.radii == 0).all() and np.abs(r.color).max() == 0
