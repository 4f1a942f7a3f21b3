"""
GPU: the drop-in module end to end, the capacity/overflow policy, and BASELINE config 2 at FULL size
(24 views x 512x512, P = 30,000) through size-independent properties plus a sampled comparison with the C oracle.
"""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_parity import check_grads, check_n_contrib, check_outputs

pytestmark = pytest.mark.gpu


def test_drop_in_module_autograd_through_topo4d_activations():
    """Renderer(raster_settings=cam)(**params2rendervar(params)) + the reference photometric loss, gradients on the
    LEAF Parameters (train.py:303-315,667) vs float64 autograd through the same activations on the CPU oracle."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from oracle import torch_oracle as TO
    from scaffold import reference_boundary as boundary, scene
    from oracle import loss_oracle
    from topo4d_amd import loss
    H = W = 64
    p_cpu = scene.make_gaussians(12, 20, opacity="B", seed=3)
    cams_cpu = scene.camera_rig(H, W, n_views=3)
    g = torch.Generator().manual_seed(0)
    p_cpu["log_scales"] = p_cpu["log_scales"] + torch.randn(240, 3, generator=g) * 0.3   # anisotropic: rotations matter
    gt = torch.rand(3, H, W, generator=g)
    cam_m, cam_c = torch.randn(3, generator=g) * 0.1, torch.randn(3, generator=g) * 0.05

    params = {k: torch.nn.Parameter(v.cuda()) for k, v in p_cpu.items()}
    cam = util.to_device(cams_cpu, "cuda")[1]
    rv = boundary.params2rendervar(params)
    rv["means2D"].retain_grad()
    im, radius, depth, alpha = Renderer(raster_settings=cam)(**rv)
    assert im.shape == (3, H, W) and depth.shape == (1, H, W) and alpha.shape == (1, H, W)
    assert radius.shape == (240,) and radius.dtype == torch.int32
    l = loss_oracle.photometric_loss_torch(im, gt.cuda(), cam_m.cuda(), cam_c.cuda())
    l.backward()
    seen = radius > 0                                                     # train.py:373-375 usage
    assert seen.any() and torch.max(radius[seen], torch.zeros_like(radius[seen]).float()).dtype == torch.float32

    pd = {k: v.double().clone().requires_grad_(True) for k, v in p_cpu.items()}
    rvd = boundary.params2rendervar(pd)
    view = TO.View(*cams_cpu[1])
    m2 = torch.zeros(240, 3, dtype=torch.float64, requires_grad=True)
    c, _, _, _ = TO.rasterize(view, rvd["means3D"], m2, rvd["opacities"], None, rvd["colors_precomp"], rvd["scales"],
                              rvd["rotations"], None, dtype=torch.float64)
    lref = loss_oracle.photometric_loss_torch(c, gt.double(), cam_m.double(), cam_c.double())
    lref.backward()
    assert abs(l.item() - lref.item()) < 1e-5
    leaf = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")
    floor = 1e-6 * max(float(pd[k].grad.abs().max()) for k in leaf)
    for k in leaf:
        a = params[k].grad.cpu().double().numpy()
        b = pd[k].grad.numpy()
        assert np.abs(a - b).max() <= 3e-4 * np.abs(b).max() + floor, (k, np.abs(a - b).max(), np.abs(b).max())
    a = rv["means2D"].grad.cpu().double().numpy()
    assert np.abs(a - m2.grad.numpy()).max() <= 3e-4 * np.abs(m2.grad.numpy()).max() + 1e-9
    assert np.abs(a[:, 2]).max() == 0


def test_checked_mode_grows_the_pair_arena_and_lazy_mode_is_memory_safe(monkeypatch):
    import topo4d_amd
    from scaffold import scene
    from topo4d_amd import rasterizer
    H = W = 96
    rv, cams = util.make_scene(20, 32, H, W, 3, opacity="B", seed=4)
    dc, _, _ = scene.output_cotangents(3, H, W, seed=5)
    ref, gref, _ = util.hip_render(cams, rv, dc)
    dev_index = torch.device("cuda").index or 0
    # (a) checked: start far too small -> T4D_ERR_PAIR_OVERFLOW -> retried with a larger arena, same results
    monkeypatch.setattr(rasterizer, "_initial_capacity", lambda P: 256)
    rasterizer._forget_scenes()
    out, g, batch = util.hip_render(cams, rv, dc)
    assert batch.prob.pair_capacity > 256 and batch.last_status.overflow == 0
    for k in ref:
        np.testing.assert_array_equal(out[k], ref[k])
    for k in gref:
        if gref[k] is not None:
            np.testing.assert_array_equal(g[k], gref[k])
    # (b) lazy with a wrong learned capacity: lists are truncated, nothing is written out of bounds, status says so
    rasterizer._scene(0, 640, H, W).capacity = 512
    rasterizer._scene(dev_index, 640, H, W).capacity = 512
    topo4d_amd.set_sync_mode("lazy")
    try:
        out2, g2, batch2 = util.hip_render(cams, rv, dc)
        st = batch2.fetch_status()
        assert st.overflow == 1 and st.max_pairs_per_view > 512
        assert np.isfinite(out2["color"]).all()
        # a truncated forward's backward is benign: every gradient is exactly zero (never sums over unwritten scratch)
        for k, v in g2.items():
            if v is not None:
                assert not v.any(), k
        dot = torch.full((3,), float("nan"), device="cuda")
        batch2.backward(dc.cuda(), cotangent_dot=dot)
        assert not dot.any()                            # ... and so is the per-view <outputs, cotangents>
    finally:
        topo4d_amd.set_sync_mode("checked")
        rasterizer._forget_scenes()


def test_view_dot_and_mark_visible():
    from topo4d_amd.rasterizer import view_dot
    import topo4d_amd
    g = torch.Generator().manual_seed(1)
    a = torch.randn(5, 3, 37, 53, generator=g).cuda()
    b = torch.randn(5, 3, 37, 53, generator=g).cuda()
    ref = (a.double() * b.double()).sum(dim=(1, 2, 3))
    out = view_dot(a, b)
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-3)
    a4 = torch.randn(3, 3, 64, 64, generator=g).cuda(); b4 = torch.randn(3, 3, 64, 64, generator=g).cuda()
    assert torch.allclose(view_dot(a4, b4).double(), (a4.double() * b4.double()).sum(dim=(1, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.equal(view_dot(a4, b4), view_dot(a4, b4))
    rv, cams = util.make_scene(6, 10, 32, 32, 1, seed=1)
    cam = util.to_device(cams, "cuda")[0]
    pts = torch.cat([rv["means3D"], rv["means3D"] + torch.tensor([0, 0, 50.0])]).cuda()
    vis = topo4d_amd.GaussianRasterizer(cam).markVisible(pts)
    assert vis[:60].all() and not vis[60:].any()


@pytest.fixture(scope="module")
def c2():
    """BASELINE config 2, both output cotangents; rendered once for the tests below."""
    from scaffold import scene
    cfg = scene.CONFIGS["C2"]
    rv, cams = util.make_scene(cfg["n_lat"], cfg["n_lon"], cfg["H"], cfg["W"], cfg["n_views"], opacity="A", seed=0)
    dc, dd, da = scene.output_cotangents(cfg["n_views"], cfg["H"], cfg["W"], seed=0, depth_alpha=True)
    out, g, batch = util.hip_render(cams, rv, dc, dd, da)
    return dict(rv=rv, cams=cams, dc=dc, dd=dd, da=da, out=out, g=g, batch=batch, st=util.decode_state(batch))


def test_c2_full_size_binning_invariants(c2):
    st = c2["st"]
    assert st["status"][0] == 0
    V = 24
    np.testing.assert_array_equal(st["tile_count"].sum(axis=1), st["view_total"])
    radii = c2["out"]["radii"]
    assert (radii >= 0).all() and (radii > 0).sum() == 24 * 30000      # the whole head is inside every frustum
    for v in (0, 7, 23):
        offs, cnts = st["tile_off"][v], st["tile_count"][v]
        # bins tile the arena without gaps or overlap
        order = np.argsort(offs, kind="stable")
        nz = order[cnts[order] > 0]
        assert (offs[nz][1:] == offs[nz][:-1] + cnts[nz][:-1]).all()
        keys = st["keys"][v]
        seen = np.zeros(30000, np.int64)
        for t in np.nonzero(cnts)[0]:
            k = keys[offs[t]: offs[t] + cnts[t]]
            assert (k[1:] > k[:-1]).all(), "per-tile list strictly ascending in (depth bits, index)"
            idx = (k & np.uint64(0xffffffff)).astype(np.int64)
            dep = (k >> np.uint64(32)).astype(np.uint32).view(np.float32)
            np.testing.assert_array_equal(dep, st["depth"][v][idx])
            seen[idx] += 1
        # every Gaussian appears once per tile of its 3-sigma rectangle
        xy, r = st["xy"][v], radii[v]
        x0 = np.clip(np.trunc((xy[:, 0] - r) / 16), 0, 32); x1 = np.clip(np.trunc((xy[:, 0] + r + 15) / 16), 0, 32)
        y0 = np.clip(np.trunc((xy[:, 1] - r) / 16), 0, 32); y1 = np.clip(np.trunc((xy[:, 1] + r + 15) / 16), 0, 32)
        np.testing.assert_array_equal(seen, ((x1 - x0) * (y1 - y0)).astype(np.int64))


def test_c2_full_size_image_identities(c2):
    out, st = c2["out"], c2["st"]
    # alpha = sum_i alpha_i T_i = 1 - prod(1 - alpha_i) = 1 - final_T
    assert np.abs(out["alpha"][:, 0] + st["final_T"] - 1.0).max() < 5e-6
    assert (st["final_T"] >= 1e-4 * 0.999).all() and (st["final_T"] <= 1.0).all()
    assert (out["alpha"] >= 0).all() and (out["depth"] >= 0).all()
    # background enters as T_final * bg: render again with bg = 1 and compare
    cams1 = [c._replace(bg=torch.ones(3)) for c in c2["cams"][:4]]
    out1, _, _ = util.hip_render(cams1, c2["rv"])
    np.testing.assert_allclose(out1["color"] - out["color"][:4], np.repeat(st["final_T"][:4, None], 3, axis=1), rtol=0, atol=2e-6)
    np.testing.assert_array_equal(out1["alpha"], out["alpha"][:4])


def test_c2_full_size_gradient_identities(c2):
    g, out, st = c2["g"], c2["out"], c2["st"]
    # (a) the backward is linear in the cotangents: doubling them doubles every gradient bit for bit
    _, g2, _ = util.hip_render(c2["cams"], c2["rv"], 2 * c2["dc"], 2 * c2["dd"], 2 * c2["da"])
    for k in g:
        if g[k] is not None:
            np.testing.assert_array_equal(g2[k], 2 * g[k])
    # (b) checksum of checksums linking forward and backward: C = sum_g w_g c_g + T bg  =>
    #     sum_g c_g . dL/dc_g = sum_pix dL/dC . (C - T bg)          (bg = 0 here)
    _, gc, _ = util.hip_render(c2["cams"], c2["rv"], c2["dc"])
    rgb = c2["rv"]["colors_precomp"].numpy().astype(np.float64)
    lhs = (gc["colors_precomp"].astype(np.float64) * rgb[None]).sum(axis=(1, 2))
    rhs = (c2["dc"].numpy().astype(np.float64) * out["color"]).sum(axis=(1, 2, 3))
    np.testing.assert_allclose(lhs, rhs, rtol=2e-4, atol=1e-9)
    # (b') the same checksum from the backward itself: cotangent_dot[v] = <color, dL/dC> + <depth, dL/dD> + <alpha, dL/dA>
    dev = torch.device("cuda")
    dot = torch.empty(24, device=dev)
    c2["batch"].backward(c2["dc"].to(dev), c2["dd"].to(dev), c2["da"].to(dev), cotangent_dot=dot)
    terms = [c2["dc"].numpy().astype(np.float64) * out["color"], c2["dd"].numpy().astype(np.float64) * out["depth"],
             c2["da"].numpy().astype(np.float64) * out["alpha"]]
    want = sum(t.sum(axis=(1, 2, 3)) for t in terms)
    scale = sum(np.abs(t).sum(axis=(1, 2, 3)) for t in terms)
    assert (np.abs(dot.double().cpu().numpy() - want) <= 2e-6 * scale).all()
    # (c) screen-space gradient has no z component; culled Gaussians would have exactly zero gradient
    assert np.abs(g["means2D"][..., 2]).max() == 0
    for k in ("means3D", "opacities", "scales", "rotations", "colors_precomp"):
        assert np.isfinite(g[k]).all()


def test_c2_full_size_all_views_against_c_oracle(c2):
    """BASELINE config 2 at full size, scenario A: EVERY one of the 24 views - radii, pair counts, per-tile counts, last
    contributors, colour / depth / alpha and all gradients - against the C oracle (threads: the oracle calls release the GIL).
    6.3 M pixels: a view may hold up to two threshold pixels (HISTORY.md section 2), the launch at most eight; every Gaussian without one
    in reach meets the plain tolerance."""
    from tests.test_gpu_parity import check_view_modulo_flips
    total = 0
    for v, (r, gref) in enumerate(util.c_oracle_render_many(c2["cams"], c2["rv"], c2["dc"], c2["dd"], c2["da"])):
        total += check_view_modulo_flips(c2["out"], c2["g"], v, r, gref, c2["st"], max_flips=2)
    assert total <= 8, f"{total} threshold pixels in 24 views"
    print(f"config 2-A, 24 views against the C oracle: {total} threshold pixels")


@pytest.mark.parametrize("views", [(0, 8, 16), (13,)])
def test_c2_scene_in_the_view_sharded_rank_shape(c2, views):
    """What one rank of BASELINE config 3 launches: three views (or one) of the full C2 scene.  That launch takes
    the other build of every stage - one-launch front end (1 view) or the small scan, the 1024-thread sort, the latency forward
    with snapshots, the depth-segmented backward - than the 24-view launch of the fixture does.  A view does not know which other
    views share its launch: binning state and images are bit-equal to the 24-view launch's, the gradients agree with it to
    summation-order rounding and with the C oracle to the parity tolerance."""
    idx = list(views)
    cams = [c2["cams"][v] for v in idx]
    dc, dd, da = c2["dc"][idx], c2["dd"][idx], c2["da"][idx]
    out, g, batch = util.hip_render(cams, c2["rv"], dc, dd, da)
    st = util.decode_state(batch)
    assert st["status"][0] == 0
    assert batch.prob.n_views == len(idx)
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(out[k], c2["out"][k][idx])
    for k in ("tile_count", "view_total", "n_contrib", "final_T"):
        np.testing.assert_array_equal(st[k], c2["st"][k][idx])
    for i, v in enumerate(idx):                                      # sorted lists: same keys in the same order
        for t in np.nonzero(st["tile_count"][i])[0][::37]:
            n = st["tile_count"][i][t]
            np.testing.assert_array_equal(st["keys"][i][st["tile_off"][i][t]:][:n],
                                          c2["st"]["keys"][v][c2["st"]["tile_off"][v][t]:][:n])
    assert st["tile_count"].max() > 128                              # lists long enough to be cut into several segments
    for k in g:
        if g[k] is None:
            continue
        a, b = c2["g"][k][idx].astype(np.float64), g[k].astype(np.float64)
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max() + 1e-12, k
    r, gref = util.c_oracle_render(cams[0], c2["rv"], dc[0], dd[0], da[0])
    check_outputs(out, r.color, r.depth, r.alpha, 0)
    check_grads(g, gref, 0)


def test_module_accepts_what_upstream_accepts():
    """Non-contiguous / float64 inputs, [P] vs [P,1] opacities, SH and precomputed-covariance paths, no_grad renders
    (train.py:463,484), all through the drop-in module."""
    import topo4d_amd
    from oracle import torch_oracle as TO
    from scaffold import scene
    H = W = 64
    rv, cams = util.make_scene(10, 16, H, W, 2, opacity="B", seed=8)
    cam = util.to_device(cams, "cuda")[0]
    R = topo4d_amd.GaussianRasterizer(cam)
    base = {k: v.cuda() for k, v in rv.items()}
    ref = [t.clone() for t in R(base["means3D"], None, base["opacities"], colors_precomp=base["colors_precomp"],
                                scales=base["scales"], rotations=base["rotations"])]
    # non-contiguous means (every other row of a bigger tensor), float64 scales, flat opacities
    big = torch.zeros(320, 3, device="cuda"); big[::2] = base["means3D"]
    out = R(big[::2], torch.zeros(160, 3, device="cuda"), base["opacities"].reshape(-1), colors_precomp=base["colors_precomp"],
            scales=base["scales"].double(), rotations=base["rotations"])
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    with torch.no_grad():                                    # report_progress-style render
        out = R(**{**base, "means2D": torch.zeros(160, 3, device="cuda")})
    assert torch.equal(out[0], ref[0]) and not out[0].requires_grad
    # SH colours through the module: degree 0 with DC = (rgb - 0.5)/C0 reproduces the precomputed colours
    sh = ((base["colors_precomp"] - 0.5) / 0.28209479177387814)[:, None, :].contiguous()
    out = topo4d_amd.GaussianRasterizer(cam._replace(sh_degree=0))(base["means3D"], None, base["opacities"], shs=sh,
                                                                   scales=base["scales"], rotations=base["rotations"])
    assert (out[0] - ref[0]).abs().max() < 1e-5
    # precomputed covariance through the module
    Rm = TO.quat_to_rot(rv["rotations"].double()); RS = Rm * rv["scales"].double()[:, None, :]; S = RS @ RS.transpose(1, 2)
    cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).float().cuda().requires_grad_(True)
    out = R(base["means3D"], None, base["opacities"], colors_precomp=base["colors_precomp"], cov3D_precomp=cov)
    assert (out[0] - ref[0]).abs().max() < 1e-5
    out[0].sum().backward()
    assert cov.grad is not None and torch.isfinite(cov.grad).all() and cov.grad.abs().max() > 0
    assert R.markVisible(base["means3D"]).all()


def test_auto_sync_mode_tracks_capacity_without_syncing():
    import topo4d_amd
    from scaffold import scene
    from topo4d_amd import rasterizer
    H = W = 96
    rv, cams = util.make_scene(20, 32, H, W, 2, opacity="B", seed=4)
    dcams = util.to_device(cams, "cuda")
    d = {k: v.cuda() for k, v in rv.items()}
    R = [topo4d_amd.GaussianRasterizer(c) for c in dcams]
    call = lambda r, sc=1.0: r(d["means3D"], None, d["opacities"], colors_precomp=d["colors_precomp"], scales=d["scales"] * sc,
                               rotations=d["rotations"])
    ref = [[t.clone() for t in call(r)] for r in R]
    rasterizer._forget_scenes()
    topo4d_amd.set_sync_mode("auto")
    try:
        for it in range(4):                                   # first call per camera is checked, the rest are not
            for k, r in enumerate(R):
                out = call(r)
                for a, b in zip(out, ref[k]):
                    assert torch.equal(a, b)
        tracks = [t for sc in rasterizer._SCENES.values() for t in sc.tracks.values()]
        assert len(tracks) == 2 and all(t.need > 0 for t in tracks)
        # a slowly growing scene (scales +8 % per iteration): capacity follows, nothing is ever truncated
        sc = 1.0
        for it in range(12):
            sc *= 1.08
            out = call(R[0], sc)
            topo4d_amd.set_sync_mode("checked"); chk = call(R[0], sc); topo4d_amd.set_sync_mode("auto")
            for a, b in zip(out, chk):
                assert torch.equal(a, b)
        # same with the host running ahead of the GPU (no synchronisation at all for 30 forwards)
        for it in range(30):
            sc *= 1.03
            out = call(R[1], sc)
        topo4d_amd.set_sync_mode("checked"); chk = call(R[1], sc); topo4d_amd.set_sync_mode("auto")
        for a, b in zip(out, chk):
            assert torch.equal(a, b)
        # an abrupt jump (scales x6 from one call to the next) overflows once: the truncated render's OWN backward raises, before
        # any gradient exists and before an optimiser could step - whenever its binning status has landed by then (it was copied
        # out right behind the binning kernels; here the device is drained first, in a loop the loss sits in between)
        rasterizer._forget_scenes()
        call(R[0]); call(R[0])
        leaf = d["means3D"].clone().requires_grad_(True)
        out_t = R[0](leaf, None, d["opacities"], colors_precomp=d["colors_precomp"], scales=d["scales"] * 6.0, rotations=d["rotations"])
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="truncated"):
            out_t[0].sum().backward()
        assert leaf.grad is None
        assert not rasterizer._PENDING                         # ... and the books are clean: nothing left to report later
        topo4d_amd.poll_truncation()
        # ... and when the status has NOT landed yet (a host running ahead of the device; forced here): the backward does not wait -
        # it launches, the library writes ZERO gradients for the truncated forward, and the truncation raises at the next look:
        # before the fused optimiser's step, or in poll_truncation() / the next forward.  Never a non-zero gradient, never a wait.
        from topo4d_amd.optim import FusedAdamPins
        rasterizer._forget_scenes()
        call(R[0]); call(R[0])
        leaf = torch.nn.Parameter(d["means3D"].clone())
        opt = FusedAdamPins([{'params': [leaf], 'name': 'means3D', 'lr': 1e-3}])
        before = leaf.detach().clone()
        out_t = R[0](leaf, None, d["opacities"], colors_precomp=d["colors_precomp"], scales=d["scales"] * 6.0, rotations=d["rotations"])
        real_landed = rasterizer._Pending.landed
        rasterizer._Pending.landed = lambda self: False
        try:
            out_t[0].sum().backward()                          # no exception, no wait
        finally:
            rasterizer._Pending.landed = real_landed
        assert leaf.grad is not None and not leaf.grad.any()   # zeros, written by the library itself
        assert len(rasterizer._PENDING) == 1
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="truncated"):
            opt.step()                                         # the look before the step
        assert torch.equal(leaf.detach(), before) and not rasterizer._PENDING
        out = call(R[0], 6.0)                                  # arena was enlarged: now complete again
        topo4d_amd.set_sync_mode("checked")
        for a, b in zip(out, call(R[0], 6.0)):
            assert torch.equal(a, b)
    finally:
        topo4d_amd.set_sync_mode("checked")
        rasterizer._forget_scenes()


def test_auto_mode_status_ring_never_hands_out_a_slot_that_is_still_owned():
    """statuses harvested out of order (a backward waiting for ITS forward) lower the in-flight count while an older
    slot of the ring is still owned; claim() must not hand that slot out again before its owner is settled."""
    from topo4d_amd import rasterizer
    rasterizer._PENDING.clear()
    try:
        track = rasterizer._AutoTrack(rasterizer._Scene((torch.cuda.current_device(), 1, 1, 1)))
        landed = lambda e, need=7: (track.host.__setitem__((e.slot, 0), need << 32), track.host.__setitem__((e.slot, 1), 0))
        e0 = track.claim(100)                               # an old forward whose status never lands (stands for: still in flight)
        later = []
        for _ in range(track.RING - 1):                     # slots 1..7, each harvested out of order by "its own backward"
            e = track.claim(100)
            landed(e)
            rasterizer.poll_truncation(wait_for=e)
            assert e.done and not e0.done
            later.append(e)
        assert track.count == 1 and track.head == e0.slot   # the count says "one in flight", the head points at e0's slot
        e8 = track.claim(100)                               # must settle e0 first (poll, then synchronise and forget it)
        assert e0.done and e8.slot == e0.slot and track.live[e8.slot] is e8
        assert [e for e in rasterizer._PENDING] == [e8]
        landed(e8, need=9)
        rasterizer.poll_truncation(wait_for=e8)
        assert track.need == 9 and track.count == 0 and not rasterizer._PENDING
    finally:
        rasterizer._PENDING.clear()


def test_view_summed_gradients_in_one_launch():
    """t4d_sum_views: the view-summed gradients rasterize_views returns (one launch instead of six torch reductions), v ascending."""
    from topo4d_amd import rasterizer
    g = torch.Generator().manual_seed(5)
    V = 7
    per_view = {"means3D": torch.randn(V, 1001, 3, generator=g).cuda(), "means2D": torch.randn(V, 1001, 3, generator=g).cuda(),
                "opacities": torch.randn(V, 1001, 1, generator=g).cuda(), "colors_precomp": None, "shs": torch.randn(V, 1001, 16, 3, generator=g).cuda(),
                "scales": torch.randn(V, 1001, 3, generator=g).cuda(), "rotations": torch.randn(V, 1001, 4, generator=g).cuda(), "cov3D_precomp": None}
    out = rasterizer._sum_views(per_view, V, need_means2D=False)
    assert out["colors_precomp"] is None and out["cov3D_precomp"] is None and out["means2D"] is None
    for k, t in per_view.items():
        if t is None or k == "means2D":
            continue
        ref = t[0].clone()
        for v in range(1, V):                       # the kernel's order: v ascending, one rounding per addition
            ref += t[v]
        assert out[k].shape == t.shape[1:] and torch.equal(out[k], ref), k
    out2 = rasterizer._sum_views(per_view, V, need_means2D=True)
    assert torch.equal(out2["means2D"], sum(per_view["means2D"][v] for v in range(V)))
