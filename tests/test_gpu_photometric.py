"""GPU: fused photometric loss (t4d_photometric_loss) vs the torch restatement of the reference (topo4d_amd/loss.py,
pinned by golden G3) and vs G3 itself.  Tolerance: loss 2e-6 absolute, gradients 2e-5 of their max."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle
from topo4d_amd import loss

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "g3_photometric.npz")


def test_matches_reference_golden_g3():
    g = np.load(G)
    im = torch.tensor(np.stack([g[f"im{i}"] for i in range(3)])).cuda().requires_grad_(True)
    gt = torch.tensor(np.stack([g[f"gt{i}"] for i in range(3)])).cuda()
    l = loss.photometric_loss(im, gt)
    l.sum().backward()
    for i in range(3):
        assert abs(l[i].item() - float(g[f"loss_{i}"])) < 2e-6
        ref = g[f"grad{i}"]
        assert np.abs(im.grad[i].cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 75, 100), (24, 512, 512)])
def test_matches_torch_restatement_with_camera_affine(shape):
    V, H, W = shape
    g = torch.Generator().manual_seed(V * H)
    im = torch.rand(V, 3, H, W, generator=g)
    gt = (im + torch.randn(V, 3, H, W, generator=g) * 0.1).clamp(0, 1)
    cm = torch.randn(V, 3, generator=g) * 0.1
    cc = torch.randn(V, 3, generator=g) * 0.05
    wv = torch.rand(V, generator=g) + 0.5
    a = [t.cuda().requires_grad_(True) for t in (im, cm, cc)]
    l = loss.photometric_loss(a[0], gt.cuda(), a[1], a[2])
    (l * wv.cuda()).sum().backward()
    if V <= 3:
        dev, dt = "cpu", torch.float64
    else:
        dev, dt = "cuda", torch.float32          # full-size case: compare on the GPU against the torch ops
    b = [t.to(dev, dt).requires_grad_(True) for t in (im, cm, cc)]
    lref = torch.stack([loss_oracle.photometric_loss_torch(b[0][v], gt[v].to(dev, dt), b[1][v], b[2][v]) for v in range(V)])
    (lref * wv.to(dev, dt)).sum().backward()
    assert torch.allclose(l.double().cpu(), lref.double().cpu(), atol=3e-6, rtol=0)
    l1_step = 2 * 0.8 * float(wv.max()) / (3 * H * W)        # |d/dx 0.8*mean|x-y|| jumps by this where x' == gt to the last bit
    for x, y in zip(a, b):
        gx, gy = x.grad.double().cpu(), y.grad.double().cpu()
        err = (gx - gy).abs()
        tol = 5e-5 * gy.abs().max() + 1e-12
        if gx.dim() == 4:
            # sign(x'-gt) is discontinuous: among ~2e7 fp32 values a handful tie to the last bit and may resolve differently
            assert (err > tol).float().mean() <= 1e-6 and err.max() <= l1_step * 1.01
        else:
            assert err.max() <= tol + l1_step


def test_feeds_the_rasterizer_backward():
    """loss(render) end to end: fused loss -> dL/dcolor -> t4d_rasterize_backward, vs the torch loss on the same render."""
    from tests import util
    from topo4d_amd import rasterize_views
    H = W = 96
    rv, cams = util.make_scene(20, 32, H, W, 3, opacity="B", seed=2)
    dcams = util.to_device(cams, "cuda")
    g = torch.Generator().manual_seed(1)
    gt = torch.rand(3, 3, H, W, generator=g).cuda()
    res = []
    for fused in (True, False):
        leaves = {k: v.cuda().clone().requires_grad_(True) for k, v in rv.items() if k != "means2D"}
        color, radii, depth, alpha = rasterize_views(dcams, leaves["means3D"], None, leaves["opacities"], None,
                                                     leaves["colors_precomp"], leaves["scales"], leaves["rotations"])
        if fused:
            l = loss.photometric_loss(color, gt).sum()
        else:
            l = sum(loss_oracle.photometric_loss_torch(color[v], gt[v]) for v in range(3))
        l.backward()
        res.append((l.item(), {k: v.grad.clone() for k, v in leaves.items()}))
    assert abs(res[0][0] - res[1][0]) < 1e-5
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        assert (a - b).abs().max() <= 1e-4 * b.abs().max() + 1e-10, k


def test_masked_l1_matches_reference_golden_g8_and_torch():
    """t4d_masked_l1_loss: the dense pass's masked L1 (train.py:394-405).  G8 holds the numbers the real get_loss_dense returned."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_masked_l1.npz"))
    im = torch.tensor(g["im"])[None].cuda().requires_grad_(True)
    l = loss.masked_l1_loss(im, torch.tensor(g["gt"])[None].cuda(), torch.tensor(g["filtered_mask"])[None].cuda())
    l.sum().backward()
    assert abs(l[0].item() - float(g["loss_im"])) < 2e-7
    np.testing.assert_allclose(im.grad[0].cpu().numpy(), g["grad_im"], rtol=1e-6, atol=0)
    # a batch of full-resolution dense-pass views against the torch restatement, with per-view weights
    V, H, W = 3, 1504, 2048
    gen = torch.Generator().manual_seed(3)
    imb = torch.rand(V, 3, H, W, generator=gen)
    gtb = torch.rand(V, 3, H, W, generator=gen)
    plane = (torch.rand(V, 1, H // 8, W // 8, generator=gen) > 0.4).float().repeat_interleave(8, 2).repeat_interleave(8, 3)
    mask = plane.expand(V, 3, H, W).contiguous()
    mask[2] = 0                                                   # an empty mask: NaN loss, zero gradient, like the reference
    wv = torch.tensor([1.0, 0.5, 2.0])
    a = imb.cuda().requires_grad_(True)
    lb = loss.masked_l1_loss(a, gtb.cuda(), mask.cuda())
    (lb[:2] * wv[:2].cuda()).sum().backward()
    for v in range(2):
        ref_in = imb[v].double().requires_grad_(True)
        ref = loss_oracle.masked_l1_loss_torch(ref_in, gtb[v].double(), mask[v].double())
        (ref * wv[v].double()).backward()
        assert abs(lb[v].item() - ref.item()) < 2e-6 * ref.item()
        np.testing.assert_allclose(a.grad[v].cpu().numpy(), ref_in.grad.numpy(), rtol=1e-5, atol=1e-12)
    assert torch.isnan(lb[2]) and not a.grad[2].any()
    assert torch.equal(loss.masked_l1_loss(a, gtb.cuda(), mask.cuda())[:2], lb[:2])            # deterministic


@pytest.mark.parametrize("shape", [(1, 512, 375), (2, 97, 210), (24, 128, 128)])
def test_gradient_does_not_depend_on_the_strip_partition(shape, monkeypatch):
    """k_photo_stream cuts a plane into strips of kFT - 20 columns and segments of rows (T4D_PH_THREADS / T4D_PH_ROWS force
    them); small launches take k_photo_tile (tiles of 64 x 44 or 32 x 44 pixels, the four filter passes as phases over LDS;
    T4D_PH_TILE forces either kernel and the small tile shape).  A pixel's arithmetic and its order do not depend on the kernel, strip, segment or tile it falls in: dL/dim
    must be bit-identical under every partition; the loss and the camera gradients are sums over another fixed partition and
    agree to rounding."""
    V, H, W = shape
    g = torch.Generator().manual_seed(5)
    im = torch.rand(V, 3, H, W, generator=g).cuda()
    gt = (im.cpu() + torch.randn(V, 3, H, W, generator=g) * 0.1).clamp(0, 1).cuda()
    cm = (torch.randn(V, 3, generator=g) * 0.1).cuda()
    cc = (torch.randn(V, 3, generator=g) * 0.05).cuda()
    out = []
    keys = ("T4D_PH_TILE", "T4D_PH_THREADS", "T4D_PH_ROWS")
    for env in ({}, {"T4D_PH_TILE": "1"}, {"T4D_PH_TILE": "32"}, {"T4D_PH_TILE": "0"}, {"T4D_PH_TILE": "0", "T4D_PH_THREADS": "64", "T4D_PH_ROWS": "16"},
                {"T4D_PH_TILE": "0", "T4D_PH_THREADS": "128", "T4D_PH_ROWS": "37"},
                {"T4D_PH_TILE": "0", "T4D_PH_THREADS": "192", "T4D_PH_ROWS": "64"},
                {"T4D_PH_TILE": "0", "T4D_PH_THREADS": "256", "T4D_PH_ROWS": "512"}):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        a = [t.clone().requires_grad_(True) for t in (im, cm, cc)]
        l = loss.photometric_loss(a[0], gt, a[1], a[2])
        l.sum().backward()
        out.append([l.detach().cpu().numpy()] + [t.grad.cpu().numpy() for t in a])
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    for o in out[1:]:
        assert np.array_equal(o[1], out[0][1])
        assert np.abs(o[0] - out[0][0]).max() < 1e-6
        for x, y in zip(o[2:], out[0][2:]):
            assert np.abs(x - y).max() <= 1e-5 * np.abs(y).max()
    assert np.isfinite(out[0][0]).all() and np.abs(out[0][1]).max() > 0


def test_random_shapes_against_the_float64_restatement():
    """Forty random batches - one to four views, images from 1 x 1 to 90 x 330 (narrower than a window, narrower than a strip's
    halo, one row, widths that leave a last strip of one column ...) - against the float64 torch restatement of the reference:
    loss 3e-6 absolute, dL/dim and the camera gradients 5e-5 of their largest entry (plus the L1 term's jump where x' == gt to
    the last bit)."""
    rng = np.random.default_rng(7)
    shapes = [(1, 1, 1), (1, 1, 40), (2, 40, 1), (1, 11, 20), (1, 5, 21), (3, 12, 44), (1, 64, 45), (2, 17, 172), (1, 33, 173)]
    while len(shapes) < 40:
        shapes.append((int(rng.integers(1, 5)), int(rng.integers(1, 91)), int(rng.integers(1, 331))))
    for n, (V, H, W) in enumerate(shapes):
        g = torch.Generator().manual_seed(100 + n)
        im = torch.rand(V, 3, H, W, generator=g)
        gt = (im + torch.randn(V, 3, H, W, generator=g) * 0.1).clamp(0, 1)
        cm = torch.randn(V, 3, generator=g) * 0.1
        cc = torch.randn(V, 3, generator=g) * 0.05
        wv = torch.rand(V, generator=g) + 0.5
        a = [t.cuda().requires_grad_(True) for t in (im, cm, cc)]
        l = loss.photometric_loss(a[0], gt.cuda(), a[1], a[2])
        (l * wv.cuda()).sum().backward()
        b = [t.double().requires_grad_(True) for t in (im, cm, cc)]
        lref = torch.stack([loss_oracle.photometric_loss_torch(b[0][v], gt[v].double(), b[1][v], b[2][v]) for v in range(V)])
        (lref * wv.double()).sum().backward()
        assert torch.allclose(l.double().cpu(), lref, atol=3e-6, rtol=0), (V, H, W)
        l1_step = 2 * 0.8 * float(wv.max()) / (3 * H * W)
        for x, y in zip(a, b):
            gx, gy = x.grad.double().cpu(), y.grad
            err = (gx - gy).abs()
            tol = 5e-5 * gy.abs().max() + 1e-12
            if gx.dim() == 4:
                assert err.max() <= tol + l1_step * 1.01 and (err > tol).float().mean() <= 1e-3, (V, H, W, float(err.max()))
            else:
                assert err.max() <= tol + l1_step * H * W, (V, H, W)
