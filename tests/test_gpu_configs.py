"""
GPU parity at the sizes BASELINE.json names, beyond config 2 (tests/test_gpu_fullsize.py):

  * config 4 at FULL size — 24 views x 2048x2048, P = 120,000, SH degree 3 (reference shape: the texture iterations at full
    resolution, train.py:729-741; SH basis helpers.py:836-922): size-independent invariants on all 24 views + two sampled
    views against the C oracle, both opacity scenarios;
  * the real dense-pass envelope — P = 1,000,000 at 4096x3008, ONE view per call, through the drop-in module with
    params2rendervar_dense-shaped kwargs (train.py:385-388, helpers.py:102-112): radii, tile counts, per-tile order (bins
    beyond the LDS sort buffer included), image and gradients against the C oracle;
  * config 2, opacity scenario B, at full size;
  * config 5 at 8192x8192 against the reference's own compiled rasterizer (oracle/_ref);
  * randomised and adversarial scenes (culling edge cases) — the former tools/stress_parity.py.
"""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_parity import (check_grads, check_grads_modulo_flips, check_n_contrib, check_outputs, check_view_modulo_flips,
                                    flipped_pixels)

pytestmark = pytest.mark.gpu


def _bins_ok(st, v, P, radii_v, gx, gy, sample_tiles=None):
    """Bins tile the arena, every list strictly ascending in (depth bits, index), keys carry the Gaussian's depth, and
    every Gaussian sits once in every tile of its 3-sigma rectangle."""
    offs, cnts = st["tile_off"][v].astype(np.int64), st["tile_count"][v].astype(np.int64)
    assert int(cnts.sum()) == int(st["view_total"][v])
    order = np.argsort(offs, kind="stable")
    nz = order[cnts[order] > 0]
    assert (offs[nz][1:] == offs[nz][:-1] + cnts[nz][:-1]).all()
    keys = st["keys"][v]
    seen = np.zeros(P, np.int64)
    tiles = np.nonzero(cnts)[0] if sample_tiles is None else sample_tiles
    for t in tiles:
        k = keys[offs[t]: offs[t] + cnts[t]]
        assert (k[1:] > k[:-1]).all(), f"tile {t}: list not strictly ascending"
        idx = (k & np.uint64(0xffffffff)).astype(np.int64)
        dep = (k >> np.uint64(32)).astype(np.uint32).view(np.float32)
        np.testing.assert_array_equal(dep, st["depth"][v][idx])
        seen[idx] += 1
    if sample_tiles is None:
        xy, r = st["xy"][v], radii_v
        x0 = np.clip(np.trunc((xy[:, 0] - r) / 16), 0, gx); x1 = np.clip(np.trunc((xy[:, 0] + r + 15) / 16), 0, gx)
        y0 = np.clip(np.trunc((xy[:, 1] - r) / 16), 0, gy); y1 = np.clip(np.trunc((xy[:, 1] + r + 15) / 16), 0, gy)
        want = np.where(r > 0, (x1 - x0) * (y1 - y0), 0).astype(np.int64)
        np.testing.assert_array_equal(seen, want)


# ------------------------------------------------------------------------------------------------------------------
# config 1: one 256x256 view, 5k fixed-topology Gaussians, forward colour + depth (+ alpha)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opacity", ["A", "B"])
def test_c1_single_view_forward_colour_depth_alpha(opacity, render_build):
    """BASELINE config 1's exact workload (scene.CONFIGS["C1"]): the plumbing case - one camera, P = 5,000, forward only,
    through the drop-in module (the call of train.py:307), every pixel of colour, depth and alpha against the C oracle."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from scaffold import scene
    cfg = scene.CONFIGS["C1"]
    H, W = cfg["H"], cfg["W"]
    rv, cams = util.make_scene(cfg["n_lat"], cfg["n_lon"], H, W, cfg["n_views"], opacity=opacity, seed=0)
    assert (rv["means3D"].shape[0], H, W, len(cams)) == (5000, 256, 256, 1)
    cam = util.to_device(cams, "cuda")[0]
    with torch.no_grad():
        im, radius, depth, alpha = GaussianRasterizer(raster_settings=cam)(
            means3D=rv["means3D"].cuda(), means2D=torch.zeros_like(rv["means3D"]).cuda(), opacities=rv["opacities"].cuda(),
            colors_precomp=rv["colors_precomp"].cuda(), scales=rv["scales"].cuda(), rotations=rv["rotations"].cuda())
    assert im.shape == (3, H, W) and depth.shape == (1, H, W) and alpha.shape == (1, H, W) and radius.shape == (5000,)
    r, _ = util.c_oracle_render(cams[0], rv)
    np.testing.assert_array_equal(radius.cpu().numpy(), r.radii)
    hip = dict(color=im[None].cpu().numpy(), depth=depth[None].cpu().numpy(), alpha=alpha[None].cpu().numpy())
    check_outputs(hip, r.color, r.depth, r.alpha, 0)          # max_flips = 0: every pixel within 2e-5
    assert (alpha > 0.5).float().mean() > 0.2                 # the head really is in the picture


# ------------------------------------------------------------------------------------------------------------------
# config 4: 24 x 2048^2, P = 120k, SH degree 3
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opacity", ["A", "B"])
def test_c4_full_size_sh3(opacity):
    from scaffold import scene
    cfg = scene.CONFIGS["C4"]
    H, W, V = cfg["H"], cfg["W"], cfg["n_views"]
    rv, cams = util.make_scene(cfg["n_lat"], cfg["n_lon"], H, W, V, opacity=opacity, sh_degree=cfg["sh_degree"], seed=0)
    P = rv["means3D"].shape[0]
    assert (P, H, W, V) == (120000, 2048, 2048, 24)
    rv["shs"][::11, 0, :] = -3.0                       # some colours go negative: the clamp flags are exercised at size
    dc, _, _ = scene.output_cotangents(V, H, W, seed=0)
    out, g, batch = util.hip_render(cams, rv, dc)
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    radii = out["radii"]
    assert (radii > 0).sum() == V * P                  # the whole head is inside every frustum
    for v in (0, 11, 23):
        _bins_ok(st, v, P, radii[v], W // 16, H // 16)
    # image identities on all views
    assert np.abs(out["alpha"][:, 0] + st["final_T"] - 1.0).max() < 5e-6
    assert (st["final_T"] >= 1e-4 * 0.999).all() and (st["final_T"] <= 1.0).all()
    assert (out["alpha"] >= 0).all() and (out["depth"] >= 0).all() and np.isfinite(out["color"]).all()
    assert np.abs(g["means2D"][..., 2]).max() == 0
    for k in ("means3D", "opacities", "scales", "rotations", "shs"):
        assert np.isfinite(g[k]).all(), k
    # linearity of the backward in the cotangent, bit for bit (4 views are enough at this size)
    sub = [0, 7, 13, 23]
    _, g1, _ = util.hip_render([cams[v] for v in sub], rv, dc[sub])
    _, g2, _ = util.hip_render([cams[v] for v in sub], rv, 2 * dc[sub])
    for k in g1:
        if g1[k] is not None:
            np.testing.assert_array_equal(g1[k], g[k][sub])
            np.testing.assert_array_equal(g2[k], 2 * g1[k])
    # SIX of the 24 views against the C oracle (the oracle calls release the GIL: six threads).  4.2 M pixels per view: a
    # T < 1e-4 / alpha >= 1/255 decision within an ulp of its threshold can fall the other way (device exp vs glibc expf) on a
    # handful of them - at most 8 per view here (scenario B blends ten times more splats per pixel than A: 5 on one view, 0-2 on
    # the others), never a systematic difference
    six = (1, 5, 9, 14, 18, 22)
    total = 0
    for v, (r, gref) in zip(six, util.c_oracle_render_many([cams[v] for v in six], rv, [dc[v] for v in six])):
        total += check_view_modulo_flips(out, g, v, r, gref, st, max_flips=8,
                                         keys=("means3D", "means2D", "opacities", "scales", "rotations", "shs"))
    assert total <= 16, f"{total} threshold pixels in 6 views"
    print(f"config 4-{opacity}, 6 views against the C oracle: {total} threshold pixels")


# ------------------------------------------------------------------------------------------------------------------
# config 2, scenario B, full size
# ------------------------------------------------------------------------------------------------------------------
def test_c2_full_size_scenario_b():
    from scaffold import scene
    cfg = scene.CONFIGS["C2"]
    H, W, V = cfg["H"], cfg["W"], cfg["n_views"]
    rv, cams = util.make_scene(cfg["n_lat"], cfg["n_lon"], H, W, V, opacity="B", seed=0)
    dc, dd, da = scene.output_cotangents(V, H, W, seed=0, depth_alpha=True)
    out, g, batch = util.hip_render(cams, rv, dc, dd, da)
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    for v in (0, 12, 23):
        _bins_ok(st, v, 30000, out["radii"][v], W // 16, H // 16)
    assert np.abs(out["alpha"][:, 0] + st["final_T"] - 1.0).max() < 5e-6
    # unsaturated opacities: pixels blend far more splats than in scenario A
    assert st["n_contrib"].max() > 20
    _, gc, _ = util.hip_render(cams, rv, dc)
    rgb = rv["colors_precomp"].numpy().astype(np.float64)
    lhs = (gc["colors_precomp"].astype(np.float64) * rgb[None]).sum(axis=(1, 2))
    rhs = (dc.numpy().astype(np.float64) * out["color"]).sum(axis=(1, 2, 3))
    np.testing.assert_allclose(lhs, rhs, rtol=2e-4, atol=1e-9)
    # ALL 24 views against the C oracle (threads: the oracle calls release the GIL); threshold pixels as in the scenario-A test
    total = 0
    for v, (r, gref) in enumerate(util.c_oracle_render_many(cams, rv, dc, dd, da)):
        total += check_view_modulo_flips(out, g, v, r, gref, st, max_flips=2)
    assert total <= 8, f"{total} threshold pixels in 24 views"
    print(f"config 2-B, 24 views against the C oracle: {total} threshold pixels")


# ------------------------------------------------------------------------------------------------------------------
# the dense-pass envelope: P = 1M, 4096x3008, one view per call, drop-in module, params2rendervar_dense kwargs
# ------------------------------------------------------------------------------------------------------------------
def test_dense_envelope_one_million_gaussians_through_the_drop_in():
    """train.py:385-388: `rendervar = params2rendervar_dense(params, variables); im, radius, _, _ = Renderer(raster_settings=
    curr_data['cam'])(**rendervar)` at the texture pass's size (helpers.py:608-609: --density 30 => order 10^6 Gaussians,
    full-resolution ~4K images).  dense_means3D is a plain tensor (re-interpolated per frame, train.py:259-261), opacities
    0.9999 and scales log(nn_dist) are Parameters with LR 0 (train.py:257,262,283-284)."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from scaffold import reference_boundary as boundary, scene
    from topo4d_amd import rasterizer
    H, W = 3008, 4096
    p = scene.make_gaussians(1000, 1000, opacity="A", seed=0)
    P = p["means3D"].shape[0]
    assert P == 1_000_000
    dense = {
        "dense_means3D": p["means3D"].cuda(),                                                   # not a Parameter
        "dense_rgb_colors": torch.nn.Parameter(p["rgb_colors"].cuda()),
        "dense_unnorm_rotations": torch.nn.Parameter(p["unnorm_rotations"].cuda()),
        "dense_logit_opacities": torch.nn.Parameter(torch.full((P, 1), float(np.log(0.9999 / 0.0001))).cuda()),
        "dense_log_scales": torch.nn.Parameter((p["log_scales"] + float(np.log(2.0))).cuda()),   # log(nn_dist), not half of it
    }
    cam_cpu = scene.camera_rig(H, W, n_views=24)[9]
    cam = util.to_device([cam_cpu], "cuda")[0]
    rendervar = boundary.params2rendervar_dense(dense)
    assert set(rendervar) == {"means3D", "colors_precomp", "rotations", "opacities", "scales", "means2D"}
    rendervar["means2D"].retain_grad()
    rasterizer._BATCH_LOG = []
    try:
        im, radius, depth, alpha = Renderer(raster_settings=cam)(**rendervar)      # train.py:388 discards the last two
        batch = rasterizer._BATCH_LOG[-1]
    finally:
        rasterizer._BATCH_LOG = None
    assert im.shape == (3, H, W) and radius.shape == (P,) and radius.dtype == torch.int32
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    dc, _, _ = scene.output_cotangents(1, H, W, seed=3)
    dc = dc * 100.0                                     # 12.3 M pixels: keep the gradients well above the comparison floor
    (im * dc[0].cuda()).sum().backward()
    seen = radius > 0                                                                         # train.py:411-414 bookkeeping
    assert seen.all()

    rv_cpu = {k: v.detach().cpu() for k, v in rendervar.items() if k != "means2D"}
    r, gref = util.c_oracle_render(cam_cpu, rv_cpu, dc[0])
    np.testing.assert_array_equal(radius.cpu().numpy(), r.radii)
    assert int(st["view_total"][0]) == r.num_rendered
    os_ = r.state()
    counts = (os_["ranges"][:, 1] - os_["ranges"][:, 0]).astype(np.int64)
    np.testing.assert_array_equal(st["tile_count"][0], counts)
    # the kernels' three sort regimes all occur at this size
    assert (counts > 8192).any() and ((counts > 2048) & (counts <= 8192)).any() and ((counts > 512) & (counts <= 2048)).any()
    rng = np.random.default_rng(0)
    nz = np.nonzero(counts)[0]
    sample = np.unique(np.concatenate([np.argsort(counts)[-40:], rng.choice(nz, 600, replace=False),
                                       np.nonzero((counts > 2048) & (counts <= 8192))[0][:40]]))
    for t in sample:                                                   # per-tile order bit-exact vs the oracle's stable sort
        off = int(st["tile_off"][0, t])
        mine = (st["keys"][0, off: off + counts[t]] & np.uint64(0xffffffff)).astype(np.uint32)
        np.testing.assert_array_equal(mine, os_["point_list"][os_["ranges"][t, 0]: os_["ranges"][t, 1]])
    _bins_ok(st, 0, P, r.radii, W // 16, H // 16, sample_tiles=sample)
    vis = r.radii > 0
    np.testing.assert_array_equal(st["xy"][0][vis], os_["xy"][vis])
    np.testing.assert_array_equal(st["conic_opacity"][0][vis], os_["conic_opacity"][vis])
    check_n_contrib(st["n_contrib"][0], os_["n_contrib"], max_flips=8)          # 12.3 M pixels (see the config-4 test)
    hip = dict(color=im.detach().cpu().numpy()[None], depth=depth.detach().cpu().numpy()[None], alpha=alpha.detach().cpu().numpy()[None])
    check_outputs(hip, r.color, r.depth, r.alpha, 0, max_flips=8)
    mine = {"means2D": rendervar["means2D"].grad.cpu().numpy()[None], "colors_precomp": dense["dense_rgb_colors"].grad.cpu().numpy()[None]}
    flips = flipped_pixels(hip, 0, r, st["n_contrib"][0])
    check_grads_modulo_flips(mine, gref, 0, flips, os_["xy"], r.radii, keys=("colors_precomp", "means2D"))
    assert dense["dense_means3D"].grad is None                          # not a leaf that requires grad
    for k in ("dense_unnorm_rotations", "dense_logit_opacities", "dense_log_scales"):
        assert torch.isfinite(dense[k].grad).all(), k


# ------------------------------------------------------------------------------------------------------------------
# config 5 at 8192^2 against the reference's own code
# ------------------------------------------------------------------------------------------------------------------
def test_c5_texture_bake_8192_bit_identical_to_reference_code():
    from oracle import texture_oracle as TX
    from scaffold.scene import uv_mesh
    from topo4d_amd import texture
    res, n = 8192, 1025                                   # 1,050,625 vertices, 2,097,152 triangles
    verts, tris, colors = uv_mesh(n, res, res, seed=0)
    ref, dref = TX.render_colors_cpu(verts, tris, colors, res, res, return_depth=True)      # oracle/_ref when present
    img, dep = texture.render_colors(verts, tris, colors, res, res, return_depth=True)
    np.testing.assert_array_equal(img.cpu().numpy(), ref)
    np.testing.assert_array_equal(dep.cpu().numpy(), dref)
    assert (ref.reshape(-1, 3).any(axis=1)).mean() > 0.5


# ------------------------------------------------------------------------------------------------------------------
# randomised parity (formerly tools/stress_parity.py)
# ------------------------------------------------------------------------------------------------------------------
def _randomised_trial(trial, strict=False):
    """One seeded random scene (size, scale, anisotropy, orientation, opacity scenario, with/without depth+alpha cotangents)
    against the C oracle.  At most two pixels per view may take a discrete decision (alpha >= 1/255, T < 1e-4) the other way
    than glibc's expf does; a Gaussian may miss the gradient tolerance ONLY with such a pixel inside its 3-sigma square
    (check_grads_modulo_flips).  strict=True (tools/soak_parity.py --strict): no such allowance for the gradients."""
    from scaffold import scene
    rng = np.random.default_rng(1000 + trial)
    H, W = int(rng.integers(40, 150)), int(rng.integers(40, 150))
    V = 2
    opacity = "AB"[trial % 2]
    rv, cams = util.make_scene(int(rng.integers(8, 30)), int(rng.integers(10, 40)), H, W, V, opacity=opacity, seed=100 + trial)
    sc = float(rng.choice([0.3, 0.7, 1.0, 2.0, 4.0, 8.0]))
    aniso = torch.tensor(rng.uniform(0.3, 3.0, size=(rv["scales"].shape[0], 3)), dtype=torch.float32)
    rv["scales"] = rv["scales"] * sc * aniso
    q = torch.tensor(rng.normal(size=(rv["rotations"].shape[0], 4)), dtype=torch.float32)
    rv["rotations"] = torch.nn.functional.normalize(q)
    dc, dd, da = scene.output_cotangents(V, H, W, seed=trial, depth_alpha=True)
    use_da = trial % 3 != 0
    hip, hg, batch = util.hip_render(cams, rv, dc, dd if use_da else None, da if use_da else None)
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    n_flips = 0
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v], dd[v] if use_da else None, da[v] if use_da else None)
        np.testing.assert_array_equal(hip["radii"][v], r.radii)
        flips = flipped_pixels(hip, v, r, st["n_contrib"][v])
        assert len(flips) <= 2, f"view {v}: {len(flips)} pixels took a discrete decision the other way"
        check_outputs(hip, r.color, r.depth, r.alpha, v, max_flips=len(flips))
        if strict:
            check_grads(hg, g, v)
        else:
            f64 = lambda: {k: t.numpy() for k, t in util.torch_oracle_render(cams[v], rv, dc[v], dd[v] if use_da else None,
                                                                              da[v] if use_da else None)[1].items()}
            check_grads_modulo_flips(hg, g, v, flips, st["xy"][v], hip["radii"][v], truth=f64)
        n_flips += len(flips)
    return n_flips


@pytest.mark.parametrize("trial", range(12))
def test_randomised_scenes_against_c_oracle(trial):
    _randomised_trial(trial)


def test_soak_seed_2635_the_fp32_reference_is_the_noisy_one(render_build):
    """Named regression (round 3's 2,000 further soak seeds on the final kernels): no flipped pixel, yet one Gaussian's dL/dscales
    is 2.9e-4 of the tensor's largest entry (1.3e-7 absolute) away from the C oracle.  tools/diag_trial.py 2635: the C oracle is
    8.2e-8 from the float64 autograd value, the kernels 6.9e-9 - a cancelling sum of a x8-scaled anisotropic splat rounded in
    fp32 by the reference (by how much depends on the host: the C oracle is built -march=native).  The check accepts a Gaussian
    that meets the tolerance against the float64 oracle."""
    _randomised_trial(2635)


def test_soak_seed_171_one_threshold_pixel(render_build):
    """Named regression: round 2's soak (tools/soak_parity.py --strict) fails on this seed in both builds - ONE pixel where a
    splat's alpha is 1.00000009/255, `v_exp_f32` and glibc's expf land on different sides of 1/255, and one Gaussian's dL/dmeans3D
    moves by 3.4e-4 of the scene's largest gradient.  The flip-aware check explains it (exactly one flipped pixel, in reach)."""
    assert _randomised_trial(171) >= 1
    with pytest.raises(AssertionError):
        _randomised_trial(171, strict=True)


def test_soak_seed_962_flipped_pixel_beyond_the_absolute_bound(render_build):
    """Named regression from round 3's soak over 1,000 seeds (tools/soak_parity.py 12 1000: 2,000 runs, 22 threshold pixels, this
    seed the only one the flip-aware check of the time rejected): the flipped pixel's contribution moves one Gaussian's dL/dmeans3D
    by 1.4e-4 - above the 1e-4 absolute bound that holds for every Gaussian WITHOUT a flipped pixel in reach."""
    assert _randomised_trial(962) >= 1
    with pytest.raises(AssertionError):
        _randomised_trial(962, strict=True)


BUILDS = (("throughput", {"T4D_LATENCY_TILES": "0", "T4D_NO_SEGMENTS": "1"}),          # what a 24-view launch runs
          ("latency + segments", {"T4D_LATENCY_TILES": "1000000000"}),                 # what a one-view call runs
          ("throughput + segments", {"T4D_LATENCY_TILES": "0"}))                       # what a 2-8 view launch runs


def test_randomised_soak_all_builds(monkeypatch):
    """A FIXED slice of tools/soak_parity.py inside the driver-run suite: seeds 12..211 - 200 randomised
    scenes against the C oracle with the flip-aware check - each under one of the three builds of the render kernels (seed mod 3),
    the first 20 under all three: 240 runs.  Round 2's soak of 400 runs had one scene miss the plain gradient tolerance because
    of ONE threshold pixel; with the flip-aware check every run must be green."""
    import time
    t0 = time.time()
    runs = flips = 0
    for trial in range(12, 212):
        for b, (name, env) in enumerate(BUILDS):
            if trial >= 32 and trial % 3 != b:
                continue
            monkeypatch.delenv("T4D_NO_SEGMENTS", raising=False)
            for k, val in env.items():
                monkeypatch.setenv(k, val)
            try:
                flips += _randomised_trial(trial)
            except AssertionError as e:
                raise AssertionError(f"trial {trial} ({name}): {e}") from e
            runs += 1
    assert runs == 240
    print(f"soak: 200 seeds, {runs} runs over {len(BUILDS)} builds, {flips} threshold pixels in total, {time.time() - t0:.0f} s")


# ------------------------------------------------------------------------------------------------------------------
# adversarial cases for the sub-block culling (cutoff_radius2 / subblock_touch_mask use fast intrinsics + a margin)
# ------------------------------------------------------------------------------------------------------------------
def _unproject(cam, px, py, z):
    """World point that the camera `cam` (setup_camera layout) projects to pixel centre coordinates (px, py) at view depth z."""
    H, W = cam.image_height, cam.image_width
    ndc_x = (2.0 * px + 1.0) / W - 1.0
    ndc_y = (2.0 * py + 1.0) / H - 1.0
    pv = torch.stack([ndc_x * cam.tanfovx * z, ndc_y * cam.tanfovy * z, z, torch.ones_like(z)], dim=1).double()
    w2c = cam.viewmatrix.reshape(4, 4).t().double()                     # the record holds the transpose (helpers.py:67)
    return (torch.linalg.inv(w2c) @ pv.t()).t()[:, :3].float()


def test_culling_edges_opacity_at_one_over_255():
    """Opacities within a few ulp of 1/255 (alpha can reach the threshold only at the very centre), and just below it
    (never drawn - Topo4D's eye interior is 1e-6, train.py:626)."""
    from scaffold import scene
    H = W = 96
    cam = scene.camera_rig(H, W, n_views=3)[1]
    g = torch.Generator().manual_seed(5)
    n = 600
    px = torch.randint(8, W - 8, (n,), generator=g).float()
    py = torch.randint(8, H - 8, (n,), generator=g).float()
    z = 0.8 + 0.2 * torch.rand(n, generator=g)
    means = _unproject(cam, px, py, z)
    thr = np.float32(1.0 / 255.0)
    steps = np.array([-64, -8, -3, -2, -1, 0, 1, 2, 3, 8, 64, 4096], np.int64)
    op = (thr.view(np.int32) + steps[np.arange(n) % len(steps)].astype(np.int32)).view(np.float32)
    op[::29] = 1e-6
    rv = dict(means3D=means, opacities=torch.tensor(op)[:, None], scales=torch.full((n, 3), 0.004),
              rotations=torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1), colors_precomp=torch.rand(n, 3, generator=g))
    dc, dd, da = scene.output_cotangents(1, H, W, seed=6, depth_alpha=True)
    hip, hg, batch = util.hip_render([cam], rv, dc * 1e3, dd, da)
    r, gref = util.c_oracle_render(cam, rv, dc[0] * 1e3, dd[0], da[0])
    np.testing.assert_array_equal(hip["radii"][0], r.radii)
    st = util.decode_state(batch)
    flips = flipped_pixels(hip, 0, r, st["n_contrib"][0])
    assert len(flips) <= 4
    check_outputs(hip, r.color, r.depth, r.alpha, 0, max_flips=len(flips))
    check_grads_modulo_flips(hg, gref, 0, flips, st["xy"][0], hip["radii"][0])


def test_culling_edges_needles_and_sub_block_corners():
    """Needle-shaped splats (conic eigenvalue ratio >> 1) in every orientation, centred ON the corners and edges of the 4x4
    sub-blocks the render kernels cull by, plus fat splats whose cut-off circle grazes sub-block corners."""
    from scaffold import scene
    H = W = 128
    cams = scene.camera_rig(H, W, n_views=3)[:2]
    g = torch.Generator().manual_seed(9)
    n = 900
    # centres on the sub-block lattice: pixel-centre coordinate k*4 - 0.5 is the corner shared by four sub-blocks
    kx = torch.randint(3, W // 4 - 3, (n,), generator=g).float()
    ky = torch.randint(3, H // 4 - 3, (n,), generator=g).float()
    px = kx * 4 - 0.5
    py = ky * 4 - 0.5
    third = n // 3
    px[third: 2 * third] += 2.0                                             # on a sub-block edge, mid-way
    px[2 * third:] += torch.rand(n - 2 * third, generator=g) * 4            # anywhere
    py[2 * third:] += torch.rand(n - 2 * third, generator=g) * 4
    z = 0.75 + 0.3 * torch.rand(n, generator=g)
    means = _unproject(cams[0], px, py, z)
    scales = torch.full((n, 3), 0.0004)
    scales[:, 0] = torch.where(torch.arange(n) % 2 == 0, torch.tensor(0.02), torch.tensor(0.003))     # needles / short needles
    fat = torch.arange(n) % 7 == 0
    scales[fat] = 0.006 + 0.004 * torch.rand(int(fat.sum()), 1, generator=g)
    rot = torch.nn.functional.normalize(torch.randn(n, 4, generator=g))
    op = torch.rand(n, 1, generator=g) * 0.9 + 0.05
    op[::5] = 1.0
    rv = dict(means3D=means, opacities=op, scales=scales, rotations=rot, colors_precomp=torch.rand(n, 3, generator=g))
    dc, dd, da = scene.output_cotangents(2, H, W, seed=10, depth_alpha=True)
    hip, hg, batch = util.hip_render(cams, rv, dc, dd, da)
    st = util.decode_state(batch)
    co = st["conic_opacity"][0]
    mid = 0.5 * (co[:, 0] + co[:, 2]); det = co[:, 0] * co[:, 2] - co[:, 1] ** 2
    ratio = (mid + np.sqrt(np.maximum(mid * mid - det, 0))) ** 2 / np.maximum(det, 1e-30)
    assert np.nanmax(ratio[hip["radii"][0] > 0]) > 50.0                     # needles really are needles on screen
    for v in range(2):
        r, gref = util.c_oracle_render(cams[v], rv, dc[v], dd[v], da[v])
        np.testing.assert_array_equal(hip["radii"][v], r.radii)
        flips = flipped_pixels(hip, v, r, st["n_contrib"][v])
        assert len(flips) <= 2
        check_outputs(hip, r.color, r.depth, r.alpha, v, max_flips=len(flips))
        check_grads_modulo_flips(hg, gref, v, flips, st["xy"][v], hip["radii"][v])
