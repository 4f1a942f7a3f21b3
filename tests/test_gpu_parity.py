"""
GPU parity tests proper: the HIP path, called THROUGH THE C ABI (ViewBatch -> ctypes -> t4d_rasterize_*),
against the oracles on identical seeded inputs.

Tolerances (fp32 kernels; north_star asks for gradient max-abs-error < 1e-4):
  * integer state (radii, tile bins, per-tile order, n_contrib): bit-exact against the C oracle;
  * colour / depth / alpha: |err| <= OUT_TOL = 2e-5 for EVERY pixel of the seeded scenes (max_flips = 0).  Only the
    randomised / adversarial cases pass an explicit max_flips > 0: there a discrete test (alpha >= 1/255, T < 1e-4,
    power > 0) can sit within an ulp of its threshold, `expf` is not bit-identical between glibc and the device, and
    a flipped pixel may then differ by up to FLIP_MAX;
  * gradients: max-abs-err <= GRAD_REL * max|reference gradient| (and < 1e-4 absolute), per tensor.
"""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

OUT_TOL = 2e-5
FLIP_MAX = 2e-2
GRAD_REL = 2e-4
GRAD_ABS = 1e-4


def check_outputs(hip, ref_color, ref_depth, ref_alpha, v, max_flips=0):
    """Every pixel within OUT_TOL; `max_flips` pixels (0 unless a test says otherwise) may be off by up to FLIP_MAX."""
    for name, a, b in (("color", hip["color"][v], ref_color), ("depth", hip["depth"][v], ref_depth),
                       ("alpha", hip["alpha"][v], ref_alpha)):
        err = np.abs(a.astype(np.float64) - np.asarray(b, np.float64).reshape(a.shape))
        bad = (err > OUT_TOL).reshape(err.shape[0], -1).any(axis=0)                  # per pixel, any channel
        assert bad.sum() <= max_flips, f"{name}[view {v}]: {int(bad.sum())} pixels off by > {OUT_TOL} (max {err.max():.3e})"
        assert err.max() <= FLIP_MAX, f"{name}[view {v}]: max err {err.max():.3e}"


def flipped_pixels(hip, v, r, n_contrib_mine):
    """(y, x) of the pixels where a discrete decision fell the other way than in the C oracle `r`."""
    bad = n_contrib_mine != r.state()["n_contrib"]
    for name, ref in (("color", r.color), ("depth", r.depth), ("alpha", r.alpha)):
        a = hip[name][v]
        bad |= (np.abs(a.astype(np.float64) - np.asarray(ref, np.float64).reshape(a.shape)) > OUT_TOL).any(axis=0)
    return np.argwhere(bad)


FLIP_GRAD_REL = 2e-2          # a Gaussian with a flipped pixel in reach: at most this share of the tensor's largest entry ...
FLIP_GRAD_ABS = 1e-3          # ... and never more than this (a large localised error must still fail; soaks saw <= 1.4e-4)


def check_grads_modulo_flips(hip_g, ref_g, v, flips, xy, radii, keys=util.GRAD_KEYS, rel=GRAD_REL, truth=None):
    """check_grads where a handful of threshold decisions (`flips`, from flipped_pixels) differ from the oracle.  Every Gaussian
    is held to the plain tolerance (rel * max|reference| per tensor, and GRAD_ABS absolute) EXCEPT those with a flipped pixel in
    one of the 16x16 tiles of their 3-sigma rectangle - the pixels they are evaluated on: alpha >= 1/255 reaches 3.33 sigma at
    opacity 1, beyond the 3-sigma radius - where the flip moves that pixel's transmittance for every splat behind the flipped
    one and the "colour behind" for every splat in front of it.  Such a Gaussian's gradient differs by that pixel's whole
    contribution (soak seed 962: 1.4e-4 absolute, 0.25 % of the tensor's largest); it is held to FLIP_GRAD_REL of the tensor's
    largest entry and FLIP_GRAD_ABS absolute - sanity bounds eight times the worst of 6,000 soak runs, not a precision claim.
    `truth` (optional): a callable returning the float64 autograd oracle's gradients of this view.  The C oracle computes in
    fp32 like the kernels and has rounding of its own (soak seed 2635: a cancelling scale gradient of a x8 anisotropic splat,
    C oracle 8e-8 from the float64 value, HIP 7e-9); a Gaussian without a flipped pixel in reach that misses the tolerance
    against the C oracle passes if it meets the SAME tolerance against the float64 oracle - evaluated only then."""
    tile = 16.0
    truth_g = None
    ftx, fty = (flips[:, 1] // 16, flips[:, 0] // 16) if len(flips) else (np.zeros(0), np.zeros(0))
    for k in keys:
        if k not in ref_g or ref_g[k] is None or hip_g.get(k) is None:
            continue
        a = hip_g[k][v].astype(np.float64)
        b = np.asarray(ref_g[k], np.float64).reshape(a.shape)
        scale = max(np.abs(b).max(), 1e-30)
        err = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
        for i in np.nonzero((err > rel * scale + 1e-9) | (err >= GRAD_ABS))[0]:
            r = float(radii[i])
            x0, x1 = max(0, int((xy[i, 0] - r) / tile)), int((xy[i, 0] + r + tile - 1) / tile)      # tile_rect of t4d_raster.hip
            y0, y1 = max(0, int((xy[i, 1] - r) / tile)), int((xy[i, 1] + r + tile - 1) / tile)
            near = (ftx >= x0) & (ftx < x1) & (fty >= y0) & (fty < y1)
            if not near.any() and truth is not None and err[i] < GRAD_ABS:
                if truth_g is None:
                    truth_g = truth()
                t = np.asarray(truth_g[k], np.float64).reshape(a.shape)
                if np.abs(a[i] - t[i]).max() <= rel * scale + 1e-9:
                    continue                              # the fp32 reference is the one that is off here
            assert near.any(), f"grad {k}[view {v}], Gaussian {i}: err {err[i]:.3e} vs scale {scale:.3e} with no flipped pixel in reach"
            assert err[i] <= min(FLIP_GRAD_REL * scale, FLIP_GRAD_ABS), \
                f"grad {k}[view {v}], Gaussian {i}: err {err[i]:.3e} vs scale {scale:.3e} is more than one pixel's share"


def check_view_modulo_flips(hip, hip_g, v, r, ref_g, st, max_flips, keys=util.GRAD_KEYS, truth=None):
    """One view of a full-size launch against its C-oracle render `r`: integer state exact; at most `max_flips` THRESHOLD pixels
    (a discrete decision within round-off of its threshold, `v_exp_f32` vs glibc's expf: HISTORY.md section 2 - about one scene in forty
    has one), everything else within the plain tolerances.  Returns the number of threshold pixels."""
    np.testing.assert_array_equal(hip["radii"][v], r.radii)
    assert int(st["view_total"][v]) == r.num_rendered
    os_ = r.state()
    np.testing.assert_array_equal(st["tile_count"][v], os_["ranges"][:, 1] - os_["ranges"][:, 0])
    flips = flipped_pixels(hip, v, r, st["n_contrib"][v])
    assert len(flips) <= max_flips, f"view {v}: {len(flips)} pixels took a discrete decision the other way"
    check_n_contrib(st["n_contrib"][v], os_["n_contrib"], max_flips=len(flips))
    check_outputs(hip, r.color, r.depth, r.alpha, v, max_flips=len(flips))
    if len(flips) == 0:
        check_grads(hip_g, ref_g, v, keys=keys)
    else:
        check_grads_modulo_flips(hip_g, ref_g, v, flips, st["xy"][v], hip["radii"][v], keys=keys, truth=truth)
    return len(flips)


def check_n_contrib(mine, ref, max_flips=0):
    diff = int((np.asarray(mine) != np.asarray(ref)).sum())
    assert diff <= max_flips, f"n_contrib differs on {diff} pixels"


def check_grads(hip_g, ref_g, v, keys=util.GRAD_KEYS, rel=GRAD_REL, max_bad_rows=0):
    """max-abs-err <= rel * max|reference| per tensor.  `max_bad_rows` (0 unless an adversarial test says otherwise) Gaussians
    may exceed that - a pixel whose alpha >= 1/255 decision flipped moves one splat's gradient by a whole contribution -
    but never the absolute bound GRAD_ABS."""
    for k in keys:
        if k not in ref_g or ref_g[k] is None or hip_g.get(k) is None:
            continue
        a = hip_g[k][v].astype(np.float64)
        b = np.asarray(ref_g[k], np.float64).reshape(a.shape)
        scale = max(np.abs(b).max(), 1e-30)
        err = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
        # 1e-9 floor: a gradient that cancels to exactly 0 in one summation order is ~1e-11 in another
        bad = int((err > rel * scale + 1e-9).sum())
        assert bad <= max_bad_rows, f"grad {k}[view {v}]: {bad} Gaussians off, max-abs-err {err.max():.3e} vs scale {scale:.3e}"
        assert err.max() < GRAD_ABS, f"grad {k}[view {v}]: abs err {err.max():.3e}"


@pytest.mark.parametrize("opacity", ["A", "B"])
def test_forward_backward_vs_c_oracle(opacity, render_build):
    H = W = 128
    V = 4
    rv, cams = util.make_scene(30, 50, H, W, V, opacity=opacity, seed=1)
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(V, H, W, seed=2, depth_alpha=True)
    hip, hg, batch = util.hip_render(cams, rv, dc, dd, da)
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v], dd[v], da[v])
        # integer state: bit exact
        np.testing.assert_array_equal(hip["radii"][v], r.radii)
        os_ = r.state()
        counts = os_["ranges"][:, 1] - os_["ranges"][:, 0]
        np.testing.assert_array_equal(st["tile_count"][v], counts)
        assert int(st["view_total"][v]) == r.num_rendered
        for t in np.nonzero(counts)[0]:
            off = int(st["tile_off"][v, t])
            mine = (st["keys"][v, off: off + counts[t]] & np.uint64(0xffffffff)).astype(np.uint32)
            ref = os_["point_list"][os_["ranges"][t, 0]: os_["ranges"][t, 1]]
            np.testing.assert_array_equal(mine, ref)
        vis = r.radii > 0
        np.testing.assert_array_equal(st["xy"][v][vis], os_["xy"][vis])
        np.testing.assert_array_equal(st["depth"][v][vis], os_["depth"][vis])
        np.testing.assert_array_equal(st["conic_opacity"][v][vis], os_["conic_opacity"][vis])
        check_n_contrib(st["n_contrib"][v], os_["n_contrib"])
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v)


@pytest.mark.parametrize("opacity", ["A", "B"])
def test_gradients_vs_autograd_f64(opacity, render_build):
    """Ground truth: torch.autograd over the float64 restatement (oracle/torch_oracle.py)."""
    H = W = 96
    V = 2
    rv, cams = util.make_scene(20, 32, H, W, V, opacity=opacity, seed=5)
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(V, H, W, seed=6, depth_alpha=True)
    hip, hg, _ = util.hip_render(cams, rv, dc, dd, da)
    for v in range(V):
        outs, grads = util.torch_oracle_render(cams[v], rv, dc[v], dd[v], da[v])
        check_outputs(hip, outs["color"].numpy(), outs["depth"].numpy(), outs["alpha"].numpy(), v)
        check_grads(hg, {k: g.numpy() for k, g in grads.items()}, v)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_path(deg):
    H = W = 80
    V = 2
    rv, cams = util.make_scene(16, 24, H, W, V, opacity="B", sh_degree=deg, seed=7)
    if deg < 3:  # more coefficients stored than the active degree uses
        pad = torch.randn(rv["shs"].shape[0], 3, 3) * 0.05
        rv["shs"] = torch.cat([rv["shs"], pad], dim=1).contiguous()
    rv["shs"][::7, 0, :] = -3.0     # force negative colours -> exercises the clamp flags
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(V, H, W, seed=8, depth_alpha=True)
    hip, hg, batch = util.hip_render(cams, rv, dc, dd, da)
    # the SH entry point of the per-Gaussian backward carries the per-view <outputs, cotangents> reduction as well
    dot = torch.empty(V, device="cuda")
    batch.backward(dc.cuda(), dd.cuda(), da.cuda(), cotangent_dot=dot)
    terms = [dc.numpy().astype(np.float64) * hip["color"], dd.numpy().astype(np.float64) * hip["depth"],
             da.numpy().astype(np.float64) * hip["alpha"]]
    want = sum(t.sum(axis=(1, 2, 3)) for t in terms)
    scale = sum(np.abs(t).sum(axis=(1, 2, 3)) for t in terms)
    assert (np.abs(dot.double().cpu().numpy() - want) <= 2e-6 * scale).all()
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v], dd[v], da[v])
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v, keys=("means3D", "means2D", "opacities", "scales", "rotations", "shs"))
    outs, grads = util.torch_oracle_render(cams[0], rv, dc[0], dd[0], da[0])
    check_grads(hg, {k: g.numpy() for k, g in grads.items()}, 0,
                keys=("means3D", "means2D", "opacities", "scales", "rotations", "shs"))


def test_cov3d_precomp_path():
    H = W = 80
    V = 2
    rv, cams = util.make_scene(16, 24, H, W, V, opacity="B", seed=9)
    from oracle import torch_oracle as TO
    R = TO.quat_to_rot(rv["rotations"].double())
    RS = R * rv["scales"].double()[:, None, :]
    S = RS @ RS.transpose(1, 2)
    rv["cov3D_precomp"] = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).float()
    del rv["scales"], rv["rotations"]
    from scaffold import scene
    dc, _, _ = scene.output_cotangents(V, H, W, seed=10)
    hip, hg, _ = util.hip_render(cams, rv, dc)
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v])
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v, keys=("means3D", "means2D", "opacities", "colors_precomp", "cov3D_precomp"))
    outs, grads = util.torch_oracle_render(cams[0], rv, dc[0])
    check_grads(hg, {k: g.numpy() for k, g in grads.items()}, 0,
                keys=("means3D", "means2D", "opacities", "colors_precomp", "cov3D_precomp"))


def test_ragged_image_and_background(render_build):
    """512x375-style image (not a multiple of 16, helpers.py:807) and a non-zero background."""
    H, W, V = 75, 100, 3
    rv, cams = util.make_scene(16, 24, H, W, V, opacity="B", seed=11, bg=[0.2, 0.5, 0.9])
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(V, H, W, seed=12, depth_alpha=True)
    hip, hg, _ = util.hip_render(cams, rv, dc, dd, da)
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v], dd[v], da[v])
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v)


@pytest.mark.parametrize("P,hint", [(5000, "unknown"), (5000, "no-long-bins"), (20000, "unknown")])
def test_long_tile_lists_and_every_sort_path(P, hint, render_build, monkeypatch):
    """Thousands of large Gaussians on 4x4 tiles: every bin is longer than one LDS batch (256) and longer than k_sort_tiles'
    LDS buffer (2048), so the multi-batch blend/replay and the long-bin sort run: k_sort_long_chunks + k_merge_long on bins of 5,000
    keys (three chunks, one group of siblings) and of 20,000 keys (ten chunks, three groups), and - when the host wrongly hinted
    "no long bins" - the slow in-kernel fallback of k_sort_tiles."""
    from topo4d_amd import rasterizer
    H = W = 64
    V = 1
    g = torch.Generator().manual_seed(3)
    rv = dict(means3D=(torch.rand(P, 3, generator=g) - 0.5) * torch.tensor([0.2, 0.2, 0.1]),
              opacities=torch.rand(P, 1, generator=g) * (0.05 if P <= 5000 else 0.01) + 0.01,
              scales=torch.rand(P, 3, generator=g) * 0.05 + 0.08,
              rotations=torch.nn.functional.normalize(torch.randn(P, 4, generator=g)),
              colors_precomp=torch.rand(P, 3, generator=g))
    from scaffold import scene
    cams = scene.camera_rig(H, W, n_views=1)
    dc, dd, da = scene.output_cotangents(V, H, W, seed=4, depth_alpha=True)
    rasterizer._forget_scenes()
    if hint == "no-long-bins":
        rasterizer._scene(0, P, H, W).longest_bin = 100
        rasterizer._scene(0, P, H, W).capacity = 16 * P + 1024                     # known scene size: no checked retry
    hip, hg, batch = util.hip_render(cams, rv, dc, dd, da)
    if hint == "no-long-bins":
        assert batch.prob.flags & 16
    st = util.decode_state(batch)
    assert st["tile_count"].max() > (16384 if P > 5000 else 4096)
    r, gref = util.c_oracle_render(cams[0], rv, dc[0], dd[0], da[0])
    os_ = r.state()
    counts = os_["ranges"][:, 1] - os_["ranges"][:, 0]
    np.testing.assert_array_equal(st["tile_count"][0], counts)
    for t in np.nonzero(counts)[0]:
        off = int(st["tile_off"][0, t])
        mine = (st["keys"][0, off: off + counts[t]] & np.uint64(0xffffffff)).astype(np.uint32)
        np.testing.assert_array_equal(mine, os_["point_list"][os_["ranges"][t, 0]: os_["ranges"][t, 1]])
    assert batch.fetch_status().max_tile_pairs == counts.max()
    # every pixel blends thousands of splats whose opacity (0.01 .. 0.06) is a few times the 1/255 threshold: each splat has a
    # ring of pixels where alpha crosses it, so a few of the ~10^8 decisions may fall the other way than with glibc's expf
    flips = flipped_pixels(hip, 0, r, st["n_contrib"][0])
    assert len(flips) <= 3
    check_outputs(hip, r.color, r.depth, r.alpha, 0, max_flips=len(flips))
    check_grads_modulo_flips(hg, gref, 0, flips, st["xy"][0], hip["radii"][0])
    rasterizer._forget_scenes()


def test_bins_of_exactly_the_lds_sort_capacity_among_more_long_bins_than_cus():
    """The long-bin kernels (k_sort_long_chunks, k_merge_long) walk the length-ordered work items and leave bins of <= 2048 keys to k_sort_tiles.  The order is by length
    CLASS only (floor(log2 n)), so a bin of exactly 2048 keys can sit in front of longer bins of its class: with more long bins
    than workgroups (one per CU) a workgroup that met such a bin used to stop and leave its later bins unsorted.  350 one-tile
    views of 2100 stacked Gaussians; the near plane cuts 0..52 of them depending on the camera, 150 views keep exactly 2048."""
    from scaffold import reference_boundary as boundary
    P, V, H, W = 2100, 350, 16, 16
    g = torch.Generator().manual_seed(17)
    z = 0.5 + 1e-3 * torch.arange(P, dtype=torch.float32)
    perm = torch.randperm(P, generator=g)                        # index order != depth order
    means = torch.stack([(torch.rand(P, generator=g) - 0.5) * 0.01, (torch.rand(P, generator=g) - 0.5) * 0.01, z], 1)[perm]
    rv = dict(means3D=means, opacities=torch.full((P, 1), 0.02), scales=torch.full((P, 3), 1e-3),
              rotations=torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1), colors_precomp=torch.rand(P, 3, generator=g))
    K = np.array([[40.0, 0, W / 2], [0, 40.0, H / 2], [0, 0, 1]])
    cams, want = [], []
    for v in range(V):
        cut = 52 if v % 7 < 3 else int(torch.randint(0, 52, (1,), generator=g))     # Gaussians behind the near plane (z <= 0.2)
        w2c = np.eye(4, dtype=np.float32)
        w2c[2, 3] = -(0.3 + (cut - 0.5) * 1e-3)
        cams.append(boundary.setup_camera(W, H, K, w2c))
        want.append(P - cut)
    assert sum(1 for n in want if n == 2048) >= 100 and sum(1 for n in want if n > 2048) > 150
    hip, _, batch = util.hip_render(cams, rv)
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    np.testing.assert_array_equal(st["tile_count"][:, 0], np.array(want, np.uint32))
    for v in range(V):
        vis = np.nonzero(hip["radii"][v] > 0)[0]
        assert len(vis) == want[v]
        expect = np.sort((st["depth"][v][vis].view(np.uint32).astype(np.uint64) << np.uint64(32)) | vis.astype(np.uint64))
        off = int(st["tile_off"][v, 0])
        np.testing.assert_array_equal(st["keys"][v, off: off + want[v]], expect, err_msg=f"view {v}: bin of {want[v]} keys not sorted")


def test_long_bins_at_the_chunk_and_group_boundaries_of_the_ranking_merge():
    """k_sort_long_chunks + k_merge_long: bins of exactly 2, 3 and 4 chunks (4,096 / 6,144 / 8,192 keys: one LDS group of siblings),
    one key into the next chunk or group (2,049 / 4,097 / 8,193), and an odd length in the third group (17,001).  One-tile views of
    stacked Gaussians, the near plane cutting the list to the wanted length (as the test above)."""
    from scaffold import reference_boundary as boundary
    P, H, W = 17100, 16, 16
    g = torch.Generator().manual_seed(23)
    z = 0.5 + 1e-4 * torch.arange(P, dtype=torch.float32)
    perm = torch.randperm(P, generator=g)
    means = torch.stack([(torch.rand(P, generator=g) - 0.5) * 0.01, (torch.rand(P, generator=g) - 0.5) * 0.01, z], 1)[perm]
    rv = dict(means3D=means, opacities=torch.full((P, 1), 0.02), scales=torch.full((P, 3), 1e-3),
              rotations=torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1), colors_precomp=torch.rand(P, 3, generator=g))
    K = np.array([[40.0, 0, W / 2], [0, 40.0, H / 2], [0, 0, 1]])
    want = [2049, 4096, 4097, 6144, 8192, 8193, 17001, 2048, 12288]
    cams = []
    for n in want:
        cut = P - n                                              # Gaussians behind the near plane (z_view <= 0.2)
        w2c = np.eye(4, dtype=np.float32)
        w2c[2, 3] = -(0.3 + (cut - 0.5) * 1e-4)
        cams.append(boundary.setup_camera(W, H, K, w2c))
    hip, _, batch = util.hip_render(cams, rv)
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    got = [int(x) for x in st["tile_count"][:, 0]]
    assert got == want, got
    for v, n in enumerate(want):
        vis = np.nonzero(hip["radii"][v] > 0)[0]
        expect = np.sort((st["depth"][v][vis].view(np.uint32).astype(np.uint64) << np.uint64(32)) | vis.astype(np.uint64))
        off = int(st["tile_off"][v, 0])
        np.testing.assert_array_equal(st["keys"][v, off: off + n], expect, err_msg=f"view {v}: bin of {n} keys not sorted")


def test_degenerate_inputs():
    from scaffold import scene
    H = W = 48
    cams = scene.camera_rig(H, W, n_views=2)
    # (a) a single Gaussian
    rv = dict(means3D=torch.zeros(1, 3), opacities=torch.full((1, 1), 0.7), scales=torch.full((1, 3), 0.02),
              rotations=torch.tensor([[1.0, 0, 0, 0]]), colors_precomp=torch.tensor([[0.3, 0.6, 0.9]]))
    dc, _, _ = scene.output_cotangents(2, H, W, seed=1)
    hip, hg, _ = util.hip_render(cams, rv, dc)
    for v in range(2):
        r, g = util.c_oracle_render(cams[v], rv, dc[v])
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v)
    # (b) everything behind the camera / outside the frustum: empty bins, background only, zero gradients
    rv2 = dict(rv)
    rv2["means3D"] = torch.tensor([[0.0, 0.0, 5.0]])
    rv2["means3D"] = torch.cat([rv2["means3D"], torch.tensor([[50.0, 0.0, 0.0]])])
    for k in ("opacities", "scales", "rotations", "colors_precomp"):
        rv2[k] = rv[k].repeat(2, 1)
    hip, hg, batch = util.hip_render(cams, rv2, dc)
    assert (hip["radii"] == 0).all()
    for v in range(2):
        r, g = util.c_oracle_render(cams[v], rv2, dc[v])
        np.testing.assert_array_equal(hip["radii"][v], r.radii)
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v)


def test_bitwise_determinism(render_build):
    H = W = 128
    V = 3
    rv, cams = util.make_scene(30, 50, H, W, V, opacity="B", seed=21)
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(V, H, W, seed=22, depth_alpha=True)
    a, ga, _ = util.hip_render(cams, rv, dc, dd, da)
    b, gb, _ = util.hip_render(cams, rv, dc, dd, da)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    for k in ga:
        if ga[k] is not None:
            np.testing.assert_array_equal(ga[k], gb[k])


def test_views_batched_equals_views_one_by_one(render_build):
    """A view does not know which other views share its launch: forward outputs bit for bit.  The backward of a ONE-view launch
    cuts its lists into segments of 64 positions, that of a 2-8 view launch into segments of 128 (seg_positions, round 5): where
    the segmented backward runs, a view's gradients agree between the two launches to summation-order rounding (the partial sums
    of a pair meet in a different order), where whole tiles are replayed ("throughput") bit for bit.  Two views per launch on both
    sides are bit for bit in every build."""
    H = W = 96
    V = 5
    rv, cams = util.make_scene(20, 32, H, W, V, opacity="B", seed=31)
    from scaffold import scene
    dc, _, _ = scene.output_cotangents(V, H, W, seed=32)
    a, ga, _ = util.hip_render(cams, rv, dc)
    for v in range(V):
        b, gb, _ = util.hip_render(cams[v:v + 1], rv, dc[v:v + 1])
        for k in a:
            np.testing.assert_array_equal(a[k][v], b[k][0])
        for k in ga:
            if ga[k] is None:
                continue
            if render_build == "throughput":
                np.testing.assert_array_equal(ga[k][v], gb[k][0])
            else:
                x, y = ga[k][v].astype(np.float64), gb[k][0].astype(np.float64)
                assert np.abs(x - y).max() <= 2e-5 * np.abs(x).max() + 1e-12, k
    for v0 in (0, 3):                                        # the same segment length on both sides: exact
        b, gb, _ = util.hip_render(cams[v0:v0 + 2], rv, dc[v0:v0 + 2])
        for k in ga:
            if ga[k] is not None:
                np.testing.assert_array_equal(ga[k][v0:v0 + 2], gb[k])


def test_latency_and_throughput_builds_agree(monkeypatch):
    """The two builds of the render kernels share every per-pixel operation: forward outputs and state are bit-equal; the
    backward adds the same partial sums in a different fixed order (one slab per DPP row instead of one per wave)."""
    H, W, V = 100, 75, 3
    rv, cams = util.make_scene(24, 40, H, W, V, opacity="B", seed=41)
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(V, H, W, seed=42, depth_alpha=True)
    res = {}
    for name, lim in (("throughput", "0"), ("latency", "1000000000")):
        monkeypatch.setenv("T4D_LATENCY_TILES", lim)
        out, g, batch = util.hip_render(cams, rv, dc, dd, da)
        st = util.decode_state(batch)
        res[name] = (out, g, st)
    a, b = res["throughput"], res["latency"]
    for k in a[0]:
        np.testing.assert_array_equal(a[0][k], b[0][k])
    np.testing.assert_array_equal(a[2]["n_contrib"], b[2]["n_contrib"])
    np.testing.assert_array_equal(a[2]["final_T"], b[2]["final_T"])
    for k in a[1]:
        if a[1][k] is not None:
            scale = np.abs(a[1][k]).max()
            assert np.abs(a[1][k].astype(np.float64) - b[1][k]).max() <= 2e-6 * scale + 1e-12, k


@pytest.mark.parametrize("with_depth_alpha", [False, True])
def test_segmented_backward_agrees_with_the_whole_tile_replay(with_depth_alpha, monkeypatch):
    """Small launches cut the backward along depth (kSeg = 128 list positions per work item) and start every segment from the
    forward's snapshot of the blend state instead of from the replay of everything behind it.  Lists of up to ~1,500 pairs,
    unsaturated opacities (pixels read deep into them), a background: the segmented backward of both builds against the
    whole-tile replay (same decisions, sums rounded differently) and against the C oracle."""
    H, W, V = 64, 48, 2
    rv, cams = util.make_scene(60, 100, H, W, V, opacity="B", seed=43)
    rv["opacities"] = rv["opacities"] * 0.15                        # thin splats: the replay runs through hundreds of them
    rv["scales"] = rv["scales"] * 2.0
    cams = [c._replace(bg=torch.tensor([0.3, 0.5, 0.2])) for c in cams]
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(V, H, W, seed=44, depth_alpha=True)
    if not with_depth_alpha:
        dd = da = None
    res = {}
    for name, lat, noseg in (("whole", "0", "1"), ("segments", "0", None), ("latency-segments", "1000000000", None)):
        monkeypatch.setenv("T4D_LATENCY_TILES", lat)
        if noseg:
            monkeypatch.setenv("T4D_NO_SEGMENTS", noseg)
        else:
            monkeypatch.delenv("T4D_NO_SEGMENTS", raising=False)
        out, g, batch = util.hip_render(cams, rv, dc, dd, da)
        res[name] = (out, g, util.decode_state(batch))
    st = res["whole"][2]
    assert st["tile_count"].max() > 4 * 128                         # several segments per tile ...
    assert (st["n_contrib"] > 3 * 128).any()                        # ... and pixels that read beyond the third boundary
    a = res["whole"]
    for name in ("segments", "latency-segments"):
        b = res[name]
        for k in a[0]:
            np.testing.assert_array_equal(a[0][k], b[0][k])         # the forward is the same program per pixel
        for k in a[1]:
            if a[1][k] is not None:
                scale = np.abs(a[1][k]).max()
                assert np.abs(a[1][k].astype(np.float64) - b[1][k]).max() <= 2e-5 * scale + 1e-12, (name, k)
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v], None if dd is None else dd[v], None if da is None else da[v])
        for name in ("segments", "latency-segments"):
            check_grads(res[name][1], g, v)


@pytest.mark.parametrize("with_depth_alpha", [False, True])
def test_long_tiles_of_a_big_one_view_launch_are_segmented(with_depth_alpha, monkeypatch):
    """The texture pass's call shape - ONE view of more than 8,192 tiles - fills the chip with whole tiles, but a few lists of
    thousands of pairs would keep their workgroups walking long after the rest has finished: tiles of at least 2,048 pairs are cut
    into depth segments there (their own launch behind the whole-tile one, which skips them).  A 1472 x 1472 view (8,464 tiles) of a head plus a
    cluster of 4,000 thin splats over a handful of tiles: against the whole-tile replay of everything (T4D_NO_SEGMENTS) and the C
    oracle."""
    H = W = 1472
    rv, cams = util.make_scene(60, 100, H, W, 1, opacity="B", seed=61)
    g = torch.Generator().manual_seed(62)
    n = 4000
    centre = rv["means3D"][rv["means3D"][:, 2].argmax()]            # the point of the head nearest to camera 0's side
    extra = {
        "means3D": centre[None] + torch.randn(n, 3, generator=g) * torch.tensor([0.0015, 0.0015, 0.004]),
        "colors_precomp": torch.rand(n, 3, generator=g),
        "rotations": torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1),
        "opacities": torch.rand(n, 1, generator=g) * 0.03 + 0.005,  # thin: pixels read through a thousand of them
        "scales": torch.rand(n, 3, generator=g) * 0.0008 + 0.0004,
    }
    rv = {k: torch.cat([v, extra[k]]).contiguous() if k in extra else v for k, v in rv.items()}
    cams = [c._replace(bg=torch.tensor([0.2, 0.1, 0.4])) for c in cams]
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(1, H, W, seed=63, depth_alpha=True)
    if not with_depth_alpha:
        dd = da = None
    res = {}
    for name, noseg in (("whole", "1"), ("hybrid", None)):
        if noseg:
            monkeypatch.setenv("T4D_NO_SEGMENTS", noseg)
        else:
            monkeypatch.delenv("T4D_NO_SEGMENTS", raising=False)
        out, gr, batch = util.hip_render(cams, rv, dc, dd, da)
        res[name] = (out, gr, util.decode_state(batch))
    st = res["whole"][2]
    assert ((W + 15) // 16) * ((H + 15) // 16) > 8192
    assert (st["tile_count"] >= 2048).sum() >= 2                    # some tiles are cut into segments ...
    assert ((st["tile_count"] > 0) & (st["tile_count"] < 2048)).sum() > 2000       # ... most are not ...
    assert (st["n_contrib"] > 1024).any()                           # ... and pixels read beyond the eighth boundary of a long one
    a, b = res["whole"], res["hybrid"]
    for k in a[0]:
        np.testing.assert_array_equal(a[0][k], b[0][k])             # (the forward does not look at T4D_NO_SEGMENTS)
    for k in a[1]:
        if a[1][k] is not None:
            scale = np.abs(a[1][k]).max()
            assert np.abs(a[1][k].astype(np.float64) - b[1][k]).max() <= 2e-5 * scale + 1e-12, k
    r, go = util.c_oracle_render(cams[0], rv, dc[0], None if dd is None else dd[0], None if da is None else da[0])
    # (thousands of thin splats over a few pixels: one threshold decision that differs from the oracle's expf moves one splat's gradient)
    check_outputs(b[0], r.color, r.depth, r.alpha, 0, max_flips=4)
    check_grads(b[1], go, 0, max_bad_rows=2)
    # The long tiles' FORWARD runs parallel along depth too (t4d_raster_render_fwd_long.h: segments blended from T = 1, a prefix pass,
    # the segments in which a pixel stops walked again); T4D_NO_LONG_FWD sends them through the one-pass forward: same decisions but
    # for pixels within rounding of a threshold, sums associated differently
    monkeypatch.setenv("T4D_NO_LONG_FWD", "1")
    o1, g1, b1 = util.hip_render(cams, rv, dc, dd, da)
    monkeypatch.delenv("T4D_NO_LONG_FWD")
    s1 = util.decode_state(b1)
    moved = s1["n_contrib"] != res["hybrid"][2]["n_contrib"]
    assert moved.sum() <= 4, int(moved.sum())
    for k in ("color", "depth", "alpha"):
        err = np.abs(o1[k].astype(np.float64) - b[0][k])
        assert (err > 5e-6).reshape(err.shape[0], -1, err.shape[-2], err.shape[-1]).any(axis=(0, 1))[~moved[0]].sum() == 0, (k, err.max())
    assert np.abs(s1["final_T"] - res["hybrid"][2]["final_T"])[~moved].max() <= 1e-6
    for k in g1:
        if g1[k] is not None:
            scale = np.abs(g1[k]).max()
            assert np.abs(g1[k].astype(np.float64) - b[1][k]).max() <= (2e-5 if moved.sum() == 0 else 2e-2) * scale + 1e-12, k
    # `debug=True` of the settings tuple (synchronise + check after every kernel - eight more launches than a small view has): same bits
    o_d, g_d, _ = util.hip_render([c._replace(debug=True) for c in cams], rv, dc, dd, da)
    for k in b[0]:
        np.testing.assert_array_equal(o_d[k], b[0][k])
    for k in g_d:
        if g_d[k] is not None:
            np.testing.assert_array_equal(g_d[k], b[1][k])
    # <outputs, cotangents> from the replay: the long tiles' share comes from their first segments, the others' from the whole-tile launch
    from topo4d_amd import ViewBatch, pack_views
    dev = torch.device("cuda")
    batch = ViewBatch(pack_views(util.to_device(cams, dev), dev), H, W, 1.0, 0)
    d = lambda k: rv[k].to(dev)
    color, _, depth, alpha = batch.forward(d("means3D"), d("opacities"), d("scales"), d("rotations"), d("colors_precomp"))
    cot = [t if t is None else t.to(dev) for t in (dc, dd, da)]
    dot = torch.full((1,), float("nan"), device=dev)
    g_with = batch.backward(*cot, cotangent_dot=dot)
    want = (color.double() * cot[0].double()).sum()
    scale = (color.double() * cot[0].double()).abs().sum()
    if with_depth_alpha:
        want = want + (depth.double() * cot[1].double()).sum() + (alpha.double() * cot[2].double()).sum()
        scale = scale + (depth.double() * cot[1].double()).abs().sum() + (alpha.double() * cot[2].double()).abs().sum()
    assert abs(float(dot[0]) - float(want)) <= 2e-6 * float(scale), (dot, want, scale)
    for k in g_with:
        if g_with[k] is not None:
            assert np.array_equal(g_with[k].cpu().numpy().reshape(b[1][k].shape), b[1][k]), k


def test_long_tiles_with_sh_colours(monkeypatch):
    """The long tiles' kernels read the colours the SH evaluation of k_preprocess left behind (per view), not `colors_precomp`: a
    1472 x 1472 view with SH degree 1 and a cluster of thin splats, against the C oracle, the one-pass forward and the whole-tile replay."""
    H = W = 1472
    rv, cams = util.make_scene(50, 80, H, W, 1, opacity="B", sh_degree=1, seed=81)
    g = torch.Generator().manual_seed(82)
    n = 4000
    centre = rv["means3D"][rv["means3D"][:, 2].argmax()]
    extra = {
        "means3D": centre[None] + torch.randn(n, 3, generator=g) * torch.tensor([0.0015, 0.0015, 0.004]),
        "shs": torch.randn(n, rv["shs"].shape[1], 3, generator=g) * 0.3,
        "rotations": torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1),
        "opacities": torch.rand(n, 1, generator=g) * 0.03 + 0.005,
        "scales": torch.rand(n, 3, generator=g) * 0.0008 + 0.0004,
    }
    rv = {k: torch.cat([v, extra[k]]).contiguous() if k in extra else v for k, v in rv.items()}
    from scaffold import scene
    dc, dd, da = scene.output_cotangents(1, H, W, seed=83, depth_alpha=True)
    keys = ("means3D", "means2D", "opacities", "scales", "rotations", "shs")
    out, gr, batch = util.hip_render(cams, rv, dc, dd, da)
    st = util.decode_state(batch)
    assert (st["tile_count"] >= 2048).sum() >= 2
    r, go = util.c_oracle_render(cams[0], rv, dc[0], dd[0], da[0])
    check_outputs(out, r.color, r.depth, r.alpha, 0, max_flips=4)
    check_grads(gr, go, 0, keys=keys, max_bad_rows=2)
    for env in ("T4D_NO_LONG_FWD", "T4D_NO_SEGMENTS"):
        monkeypatch.setenv(env, "1")
        o1, g1, b1 = util.hip_render(cams, rv, dc, dd, da)
        monkeypatch.delenv(env)
        moved = util.decode_state(b1)["n_contrib"] != st["n_contrib"]
        assert moved.sum() <= 4
        for k in ("color", "depth", "alpha"):
            err = np.abs(o1[k].astype(np.float64) - out[k])
            assert (err > 5e-6).reshape(err.shape[0], -1, H, W).any(axis=(0, 1))[~moved[0]].sum() == 0, (env, k, err.max())
        for k in keys:
            scale = np.abs(g1[k]).max()
            assert np.abs(g1[k].astype(np.float64) - gr[k]).max() <= (2e-5 if moved.sum() == 0 else 2e-2) * scale + 1e-12, (env, k)


def test_shuffled_gaussian_order_takes_the_global_atomic_binning_path():
    """Mesh order is what makes the LDS tile histogram of k_preprocess effective; a random permutation at 1024^2 makes
    every workgroup's tile bounding box larger than the histogram, so the per-pair global-atomic fallback runs."""
    H = W = 1024
    V = 2
    rv, cams = util.make_scene(60, 80, H, W, V, opacity="B", seed=51)
    perm = torch.randperm(rv["means3D"].shape[0], generator=torch.Generator().manual_seed(1))
    rv = {k: v[perm].contiguous() for k, v in rv.items()}
    from scaffold import scene
    dc, _, _ = scene.output_cotangents(V, H, W, seed=52)
    hip, hg, batch = util.hip_render(cams, rv, dc)
    st = util.decode_state(batch)
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v])
        np.testing.assert_array_equal(hip["radii"][v], r.radii)
        os_ = r.state()
        counts = os_["ranges"][:, 1] - os_["ranges"][:, 0]
        np.testing.assert_array_equal(st["tile_count"][v], counts)
        for t in np.nonzero(counts)[0][::7]:
            off = int(st["tile_off"][v, t])
            mine = (st["keys"][v, off: off + counts[t]] & np.uint64(0xffffffff)).astype(np.uint32)
            np.testing.assert_array_equal(mine, os_["point_list"][os_["ranges"][t, 0]: os_["ranges"][t, 1]])
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v)


def test_debug_flag_of_the_settings_tuple(render_build):
    """`debug=True` (reference setup_camera passes False; upstream's flag dumps a snapshot on failure) maps to
    T4D_FLAG_DEBUG_SYNC: synchronise and check after every kernel.  Same kernels, so results are bit-equal, forward and
    backward, through the drop-in class."""
    from diff_gaussian_rasterization import GaussianRasterizer
    H, W = 120, 90
    rv, cams = util.make_scene(24, 40, H, W, 1, opacity="B", seed=61)
    cam = util.to_device(cams, "cuda")[0]
    res = []
    for debug in (False, True):
        leaves = {k: v.cuda().requires_grad_(True) for k, v in rv.items()}
        leaves["means2D"] = torch.zeros_like(leaves["means3D"], requires_grad=True)
        color, radii, depth, alpha = GaussianRasterizer(raster_settings=cam._replace(debug=debug))(**leaves)
        (color.square().sum() + depth.sum() + 0.5 * alpha.sum()).backward()
        res.append([t.detach().cpu().numpy() for t in (color, radii, depth, alpha)]
                   + [leaves[k].grad.cpu().numpy() for k in sorted(leaves)])
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("with_depth_alpha", [False, True])
@pytest.mark.parametrize("bg", [None, (0.3, 0.6, 0.1)])
def test_cotangent_dot_from_the_backward(with_depth_alpha, bg, render_build):
    """T4DBackwardIO.cotangent_dot: <color, dL_dcolor> (+ <depth, dL_ddepth> + <alpha, dL_dalpha>) per view, emitted by the
    replay (its suffix sum at the eye) and by the empty tiles (background at T = 1).  Checked against the float64 inner
    product of the forward's own outputs; ragged image (100 x 75: partial tiles), with and without a background, and the
    gradients must not depend on whether the dot was asked for."""
    from topo4d_amd import ViewBatch, pack_views
    from scaffold import scene
    H, W, V = 100, 75, 3
    rv, cams = util.make_scene(24, 40, H, W, V, opacity="B", seed=71, bg=bg)
    dc, dd, da = scene.output_cotangents(V, H, W, seed=72, depth_alpha=True)
    if not with_depth_alpha:
        dd = da = None
    dev = torch.device("cuda")
    batch = ViewBatch(pack_views(util.to_device(cams, dev), dev), H, W, 1.0, 0)
    d = lambda k: rv[k].to(dev)
    color, radii, depth, alpha = batch.forward(d("means3D"), d("opacities"), d("scales"), d("rotations"), d("colors_precomp"))
    cot = [t if t is None else t.to(dev) for t in (dc, dd, da)]
    dot = torch.full((V,), float("nan"), device=dev)
    g_with = batch.backward(*cot, cotangent_dot=dot)
    g_without = batch.backward(*cot)
    for k in g_with:
        if g_with[k] is not None:
            assert torch.equal(g_with[k], g_without[k]), k
    want = (color.double() * cot[0].double()).sum(dim=(1, 2, 3))
    scale = (color.double() * cot[0].double()).abs().sum(dim=(1, 2, 3))
    if with_depth_alpha:
        want = want + (depth.double() * cot[1].double()).sum(dim=(1, 2, 3)) + (alpha.double() * cot[2].double()).sum(dim=(1, 2, 3))
        scale = scale + (depth.double() * cot[1].double()).abs().sum(dim=(1, 2, 3)) + (alpha.double() * cot[2].double()).abs().sum(dim=(1, 2, 3))
    err = (dot.double() - want).abs()
    assert torch.all(err <= 2e-6 * scale), (dot, want, scale)
    dot2 = torch.empty_like(dot)
    batch.backward(*cot, cotangent_dot=dot2)
    assert torch.equal(dot, dot2)                                   # fixed summation order


def test_empty_tiles_written_by_row_fill_workgroups(monkeypatch):
    """Round 3: the pixels of empty tiles (background colour, zero depth and alpha) are written by workgroups of k_render_fwd
    that walk a ROW of tiles in image order, 16 bytes per lane when the planes allow it, 4 bytes otherwise.  Both paths must
    leave exactly the image the oracle renders: a sparse scene on a non-black background (most tiles empty), widths that are
    and are not multiples of four and of the tile size, and both builds of the render kernels."""
    for H, W in ((96, 128), (70, 100), (33, 90), (16, 4)):
        rv, cams = util.make_scene(6, 8, H, W, 3, opacity="B", seed=5, bg=(0.25, 0.5, 0.75))
        ref = [util.c_oracle_render(c, rv)[0] for c in cams]
        for tiles in ("0", "1000000000"):                    # throughput build, latency build
            monkeypatch.setenv("T4D_LATENCY_TILES", tiles)
            outs = []
            for scalar in (False, True):
                if scalar:
                    monkeypatch.setenv("T4D_FILL_SCALAR", "1")
                else:
                    monkeypatch.delenv("T4D_FILL_SCALAR", raising=False)
                # the outputs come from torch.empty: poison the blocks the caching allocator is about to hand out again, so that a
                # pixel nobody writes shows up as NaN instead of as the previous (correct) image
                poison = [torch.full((len(cams), c, H, W), float("nan"), device="cuda") for c in (3, 1, 1)]
                torch.cuda.synchronize()
                del poison
                hip, _, batch = util.hip_render(cams, rv)
                outs.append(hip)
                for v, r in enumerate(ref):
                    check_outputs(hip, r.color, r.depth, r.alpha, v, max_flips=2)       # (threshold pixels: see flipped_pixels)
                # the pixels of empty tiles are EXACTLY background / zero, and every one of them is written
                empty = util.decode_state(batch)["tile_count"].reshape(len(cams), (H + 15) // 16, (W + 15) // 16) == 0
                assert H * W <= 256 or (empty.any() and not empty.all())       # the scene really has both kinds of tiles
                pix_empty = np.repeat(np.repeat(empty, 16, axis=1), 16, axis=2)[:, :H, :W]
                for ch, b in enumerate((0.25, 0.5, 0.75)):
                    assert (hip["color"][:, ch][pix_empty] == np.float32(b)).all()
                assert (hip["depth"][:, 0][pix_empty] == 0).all() and (hip["alpha"][:, 0][pix_empty] == 0).all()
            for k in ("color", "depth", "alpha"):
                assert np.array_equal(outs[0][k], outs[1][k]), k
    monkeypatch.delenv("T4D_FILL_SCALAR", raising=False)


def test_sh_backward_degree3_kernel_equals_the_general_one(monkeypatch):
    """Degree 3 with 16 stored coefficients takes k_sh_bwd16 (coefficient rows staged once per workgroup and EIGHT views, launch
    index decoded into (block of 256 Gaussians, group of views)); any other layout takes k_sh_bwd.  Both do the same arithmetic
    in the same order: on 11 views (a full group + one of three) of 384 Gaussians (a full block + half a block) every
    gradient must be bit-identical, and the general kernel is held to the oracle by test_sh_colour_path."""
    H = W = 48
    V = 11
    rv, cams = util.make_scene(16, 24, H, W, V, opacity="B", sh_degree=3, seed=11)
    rv["shs"][::5, 0, :] = -3.0
    from scaffold import scene
    dc, _, _ = scene.output_cotangents(V, H, W, seed=12)
    monkeypatch.delenv("T4D_SH_BWD_PLAIN", raising=False)
    _, g16, _ = util.hip_render(cams, rv, dc)
    monkeypatch.setenv("T4D_SH_BWD_PLAIN", "1")
    _, gpl, _ = util.hip_render(cams, rv, dc)
    monkeypatch.delenv("T4D_SH_BWD_PLAIN", raising=False)
    assert np.abs(g16["shs"]).max() > 0
    for k in ("shs", "means2D", "opacities", "scales", "rotations"):
        assert np.array_equal(g16[k], gpl[k]), k
    # the view-direction term of dL/dmeans3D: the degree is a compile-time constant in one kernel and a run-time value in the
    # other, so the basis gradient is contracted differently - last bits only
    assert np.abs(g16["means3D"] - gpl["means3D"]).max() <= 1e-6 * np.abs(gpl["means3D"]).max()
    r, g = util.c_oracle_render(cams[9], rv, dc[9])
    check_grads(g16, g, 9, keys=("means3D", "shs"))


def test_one_view_scan_and_scatter_in_one_launch(monkeypatch):
    """One view of at most 1,024 tiles (Topo4D's own call shape) runs its binning front end as ONE launch - k_front_small:
    preprocess, then scan and scatter behind a grid-wide barrier - or, with T4D_NO_FRONT_FUSION, as k_preprocess ->
    k_scan_scatter_small, instead of the three kernels of a multi-view launch (T4D_NO_SMALL_VIEW).
    Everything the paths leave behind must be identical: tile offsets, the view's total, the status block, the sorted keys
    of every tile, the outputs and the gradients; the work items may come in another order inside a length class (both orders
    are arbitrary), so they are compared as sets per class.  Sizes: 512x375 with Topo4D's 8,280 Gaussians, a 1,024-tile
    image, and a tiny one."""
    from scaffold import scene
    switches = ("T4D_NO_FRONT_FUSION", "T4D_NO_SMALL_VIEW")
    for (n_lat, n_lon, H, W) in ((69, 120, 512, 375), (40, 60, 512, 512), (6, 8, 33, 90)):
        rv, cams = util.make_scene(n_lat, n_lon, H, W, 24 if H > 100 else 3, opacity="A", seed=3)
        cams = cams[len(cams) // 2: len(cams) // 2 + 1]
        dc, _, _ = scene.output_cotangents(1, H, W, seed=4)
        res = []
        for env in ({}, {"T4D_NO_FRONT_FUSION": "1"}, {"T4D_NO_SMALL_VIEW": "1"}):
            for k in switches:
                monkeypatch.delenv(k, raising=False)
            for k, val in env.items():
                monkeypatch.setenv(k, val)
            hip, hg, batch = util.hip_render(cams, rv, dc)
            st = util.decode_state(batch)
            res.append((hip, hg, st, batch.last_status))
        for k in switches:
            monkeypatch.delenv(k, raising=False)
        h1, g1, s1, t1 = res[2]
        for h0, g0, s0, t0 in res[:2]:
            assert (t0.max_pairs_per_view, t0.total_pairs, t0.overflow, t0.max_tile_pairs) == \
                   (t1.max_pairs_per_view, t1.total_pairs, t1.overflow, t1.max_tile_pairs)
            for k in ("tile_count", "tile_off", "view_total", "bucket_fill"):
                assert np.array_equal(s0[k], s1[k]), k
            tc, to = s0["tile_count"][0], s0["tile_off"][0]
            for t in np.nonzero(tc)[0]:
                assert np.array_equal(s0["keys"][0][to[t]:to[t] + tc[t]], s1["keys"][0][to[t]:to[t] + tc[t]]), int(t)
            for k in ("color", "depth", "alpha", "radii"):
                assert np.array_equal(h0[k], h1[k]), k
            for k in util.GRAD_KEYS:
                assert np.array_equal(g0[k], g1[k]), k
        h0 = res[0][0]
        r, g = util.c_oracle_render(cams[0], rv, dc[0])
        check_outputs(h0, r.color, r.depth, r.alpha, 0, max_flips=2)