"""GPU: the loss branches train.py really executes (train.py:631-632: use_mask = True, use_mask_dense = False) on the fused
pieces, against goldens captured from the REAL train.get_loss / train.get_loss_dense (G9, G10; oracle/gen_golden.py) and against
the same loops built from torch ops:
  * t4d_label_mask_target = helpers.get_mask + masked_gt (helpers.py:811-823, train.py:320-326), bit for bit;
  * the later-frame geometry loss (masked target, camera affine) and the texture pass's loss (no affine, + 0.02 soft colour);
  * loop.optimise_views(use_mask=...) and loop.optimise_dense_views (train.py:729-741), hand-chained vs autograd vs torch pieces;
  * loop.GraphedViews on those branches, and load_frame()."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def _groups(params, lrs):
    return [{'params': [v], 'name': k, 'lr': lrs[k]} for k, v in params.items()]


def test_label_mask_target_is_the_reference_get_mask_bit_for_bit():
    from scaffold import scene
    from topo4d_amd import loss
    g = np.load(os.path.join(G, "g9_get_loss_masked.npz"))
    colors = g["label_colors"]
    assert np.array_equal(scene.parsing_colormap_bgr(14), colors)
    inner = colors[[int(g["inner_mouth_index"])]]
    gt = torch.tensor(g["gt"]).cuda()
    for key, out in (("mask_image", "filtered_mask"), ("mask_image_soft", "filtered_mask_soft")):
        filtered, target = loss.label_mask_target(torch.tensor(g[key]).cuda(), inner, gt)
        assert np.array_equal(filtered.cpu().numpy(), g[out]), key
        want = g["gt"].copy()
        want[g[out] == 1] *= np.float32(0.1)
        assert np.array_equal(target.cpu().numpy(), want), key
    filtered, target = loss.label_mask_target(torch.tensor(g["mask_image"]).cuda(), inner, gt)
    assert np.array_equal(target.cpu().numpy(), g["later_target"])          # what the reference handed to l1_loss_v1
    f3, _ = loss.label_mask_target(torch.tensor(g["mask_image_soft"]).cuda(), colors[[7, 8, 9]], gt)
    assert np.array_equal(f3.cpu().numpy(), g["filtered_mask_soft_3_labels"])
    # a batch of cameras in one launch equals the cameras one by one; mask only / target only
    masks = torch.stack([torch.tensor(g["mask_image"]), torch.tensor(g["mask_image_soft"]), torch.tensor(g["mask_image"])]).cuda()
    gts = torch.stack([gt, gt * 0.5, gt + 1])
    fb, tb = loss.label_mask_target(masks, inner, gts)
    for v in range(3):
        f1, t1 = loss.label_mask_target(masks[v], inner, gts[v])
        assert torch.equal(fb[v], f1) and torch.equal(tb[v], t1)
    assert loss.label_mask_target(masks, inner, None)[1] is None
    assert torch.equal(loss.label_mask_target(masks, inner, gts, want_mask=False)[1], tb)
    none, t0 = loss.label_mask_target(masks, np.zeros((0, 3)), gts)          # no label selected: nothing masked
    assert none.sum() == 0 and torch.equal(t0, gts)
    # the reference's own call shape (helpers.get_mask: label names, cmap_index, per-label colour tiles)
    cmap_index = {name: i for i, name in enumerate(scene.PARSING_LABELS)}
    tiles = [torch.tile(torch.tensor(colors[i]).reshape(3, 1, 1), (1, 48, 40)).cuda() for i in range(14)]      # train.py:635
    got = loss.get_mask(["upper_lip", "inner_mouth", "lower_lip"], torch.tensor(g["mask_image_soft"]).cuda(), cmap_index, tiles)
    assert np.array_equal(got.cpu().numpy(), g["filtered_mask_soft_3_labels"])
    with pytest.raises(ValueError):
        loss.label_mask_target(masks, np.zeros((17, 3)), gts)
    with pytest.raises(ValueError):
        loss.label_mask_target(masks, inner, gts[:2])


def test_geometry_loss_of_both_frames_matches_the_reference_get_loss():
    """G9: losses['im'] and the gradients w.r.t. the render, cam_m and cam_c of the REAL train.get_loss(use_mask=True), first frame
    and later frames, from t4d_label_mask_target + t4d_photometric_loss (strip AND tile kernels)."""
    from topo4d_amd import loop, loss
    g = np.load(os.path.join(G, "g9_get_loss_masked.npz"))
    cid = int(g["cam_id"])
    inner = g["label_colors"][[int(g["inner_mouth_index"])]]
    entry = {"im": torch.tensor(g["gt"]).cuda(), "mask": torch.tensor(g["mask_image"]).cuda(), "id": cid}
    for tile in ("0", "1"):
        os.environ["T4D_PH_TILE"] = tile
        try:
            for tag, initial in (("first", True), ("later", False)):
                target = loop.target_image(entry, use_mask=True, is_initial_timestep=initial, label_colors=inner)
                assert np.array_equal(target.cpu().numpy(), g[f"{tag}_target"])
                im = torch.tensor(g[f"{tag}_im"]).cuda().requires_grad_(True)
                cm = torch.tensor(g[f"{tag}_cam_m"]).cuda().requires_grad_(True)
                cc = torch.tensor(g[f"{tag}_cam_c"]).cuda().requires_grad_(True)
                l = loss.photometric_loss(im, target, cm[cid], cc[cid])
                l.backward()
                assert abs(l.item() - float(g[f"{tag}_loss_im"])) < 2e-6
                np.testing.assert_allclose(im.grad.cpu().numpy(), g[f"{tag}_grad_im"], rtol=2e-4, atol=2e-9)
                np.testing.assert_allclose(cm.grad.cpu().numpy(), g[f"{tag}_grad_cam_m"], rtol=2e-4, atol=1e-7)
                np.testing.assert_allclose(cc.grad.cpu().numpy(), g[f"{tag}_grad_cam_c"], rtol=2e-4, atol=1e-7)
        finally:
            del os.environ["T4D_PH_TILE"]
    assert "masked_im" in entry                        # computed once, kept in the dataset entry


def test_dense_loss_matches_the_reference_get_loss_dense():
    """G10: the REAL train.get_loss_dense(use_mask=False): total, both terms, dL/drender and dL/d dense_rgb_colors."""
    from topo4d_amd import loss
    g = np.load(os.path.join(G, "g10_get_loss_dense.npz"))
    im = torch.tensor(g["im"]).cuda().requires_grad_(True)
    rgb = torch.tensor(g["dense_rgb_colors"]).cuda().requires_grad_(True)
    init = torch.tensor(g["dense_init_colors"]).cuda()
    l_im = loss.photometric_loss(im, torch.tensor(g["gt"]).cuda())
    l_soft = loss.soft_color_loss(rgb, init)
    total = l_im + 0.02 * l_soft
    total.backward()
    assert abs(total.item() - float(g["loss"])) < 2e-6
    assert abs(l_im.item() - float(g["loss_im"])) < 2e-6
    assert abs(0.02 * l_soft.item() - float(g["loss_soft_color"])) < 1e-7
    np.testing.assert_allclose(im.grad.cpu().numpy(), g["grad_im"], rtol=2e-4, atol=2e-9)
    assert np.array_equal(rgb.grad.cpu().numpy(), g["grad_dense_rgb_colors"])          # autograd route: the reference's .grad bit for bit
    # the raw entry point with the weight folded in: bit for bit what autograd left in the reference's .grad, also when ADDED to
    # an existing gradient (the rasterizer's dL/dcolours in the hand-chained iteration)
    l2, g2 = loss.soft_color_loss_raw(rgb.detach(), init, 0.02)
    assert np.array_equal(g2.cpu().numpy(), g["grad_dense_rgb_colors"]) and torch.equal(l2, l_soft.detach())
    base = torch.randn_like(init)
    acc = base.clone()
    loss.soft_color_loss_raw(rgb.detach(), init, 0.02, grad=acc, accumulate=True)
    assert torch.equal(acc, base + g2)
    # sizes around the partial-sum grid, against float64
    for rows in (1, 255, 1024, 262145, 1000003):
        x = torch.randn(rows, 3, generator=torch.Generator().manual_seed(rows)).cuda()
        y = x + torch.randn(rows, 3, generator=torch.Generator().manual_seed(rows + 1)).cuda() * 0.1
        y[::3] = x[::3]
        l, gr = loss.soft_color_loss_raw(x, y, 0.02)
        ref = (x.double() - y.double()).abs().sum(-1).mean()
        assert abs(l.item() - ref.item()) <= 2e-6 * ref.item() + 1e-9, rows
        assert torch.equal(gr, torch.sign(x - y) * (torch.tensor(0.02, dtype=torch.float32) / torch.tensor(float(rows), dtype=torch.float32)).cuda())
        l_again, _ = loss.soft_color_loss_raw(x, y, 0.02)
        assert torch.equal(l, l_again)                  # fixed summation order


def _geometry_case(H, W, V, seed=3):
    from tests import util
    from scaffold import scene
    p0 = scene.make_gaussians(12, 20, opacity="B", seed=seed)
    # anisotropic scales: the rotation of an isotropic Gaussian has a round-off gradient, which Adam turns into +-lr steps
    p0['log_scales'] = p0['log_scales'] + torch.randn(240, 3, generator=torch.Generator().manual_seed(9)) * 0.3
    p0['cam_m'] = torch.randn(V, 3, generator=torch.Generator().manual_seed(2)) * 0.05
    p0['cam_c'] = torch.randn(V, 3, generator=torch.Generator().manual_seed(3)) * 0.05
    cams = util.to_device(scene.camera_rig(H, W, n_views=V), "cuda")
    g = torch.Generator().manual_seed(5)

    def frame(t):
        return [{'cam': cams[i], 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': i,
                 'mask': scene.make_label_image(H, W, seed=10 * t + i).cuda()} for i in range(V)]
    lrs = {'means3D': 1.6e-4, 'rgb_colors': 0.0025, 'unnorm_rotations': 0.001, 'logit_opacities': 0.05, 'log_scales': 0.001,
           'cam_m': 1e-3, 'cam_c': 1e-3}
    return p0, frame, lrs


def test_masked_geometry_loop_hand_chained_equals_autograd_and_torch_pieces():
    """train.py:661-673 with use_mask=True in a later frame: loop.optimise_views chained by hand == through autograd bit for bit,
    and both follow the same loop built from torch ops (the checker's get_mask / masked_gt / loss + torch.optim.Adam)."""
    from oracle import loss_oracle
    from scaffold import scene
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W, V = 64, 96, 3
    p0, frame, lrs = _geometry_case(H, W, V)
    inner = scene.parsing_colormap_bgr(14)[[scene.PARSING_LABELS.index("inner_mouth")]]
    base = frame(1)
    res = []
    for mode in ("explicit", "autograd", "torch"):
        dataset = [dict(e) for e in base]
        params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
        mx = torch.zeros(240, device="cuda")
        if mode == "torch":
            opt = torch.optim.Adam(_groups(params, lrs), lr=0.0, eps=1e-15)
            colors = torch.tensor(inner).cuda()

            def ref_loss(im, target, cm, cc, colors=colors):
                return loss_oracle.photometric_loss_torch(im, target, cm, cc)
            for e in dataset:                              # the checker's own target, per entry (the reference: per iteration)
                e['masked_im'] = loss_oracle.masked_target_torch(e['im'], loss_oracle.label_mask_torch(e['mask'], colors))
            losses = loop.optimise_views(params, dataset, opt, n_iters=9, seed=4, max_2D_radius=mx, loss_fn=ref_loss, use_mask=True,
                                         is_initial_timestep=False)
        else:
            opt = FusedAdamPins(_groups(params, lrs), eps=1e-15)
            losses = loop.optimise_views(params, dataset, opt, n_iters=9, seed=4, max_2D_radius=mx, explicit=(mode == "explicit"),
                                         use_mask=True, is_initial_timestep=False, label_colors=inner)
            assert all('masked_im' in e for e in dataset) and any((e['masked_im'] != e['im']).any() for e in dataset)
        res.append(({k: v.detach().clone() for k, v in params.items()}, torch.stack(losses), mx))
    (pe, le, me), (pa, la, ma), (pt, lt, mt) = res
    assert torch.equal(le, la) and torch.equal(me, ma)
    for k in pe:
        assert torch.equal(pe[k], pa[k]), (k, (pe[k] - pa[k]).abs().max())
    assert torch.allclose(le, lt, atol=2e-5), (le, lt)
    assert torch.equal(me, mt)
    for k in pe:
        moved = (pt[k] - p0[k].cuda()).abs()
        ok = ((pe[k] - pt[k]).abs() <= 0.02 * moved + 3e-5)
        assert ok.float().mean() > 0.97, (k, ok.float().mean())
    # the first frame takes the plain target: a different run
    params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
    first = loop.optimise_views(params, [dict(e) for e in base], FusedAdamPins(_groups(params, lrs), eps=1e-15), n_iters=9, seed=4,
                                use_mask=True, is_initial_timestep=True)
    assert not torch.equal(torch.stack(first), le)
    # frozen parameters (requires_grad False) get neither gradient nor step on the hand-chained path, as under autograd
    for explicit in (True, False):
        params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
        params['means3D'].requires_grad_(False); params['cam_c'].requires_grad_(False)
        opt = FusedAdamPins(_groups(params, lrs), eps=1e-15)
        loop.optimise_views(params, [dict(e) for e in base], opt, n_iters=4, seed=1, explicit=explicit)
        assert torch.equal(params['means3D'], p0['means3D'].cuda()) and torch.equal(params['cam_c'], p0['cam_c'].cuda())
        assert (params['cam_m'] != p0['cam_m'].cuda()).any() and (params['rgb_colors'] != p0['rgb_colors'].cuda()).any()
        res.append({k: v.detach().clone() for k, v in params.items()})
    for k in res[-1]:
        assert torch.equal(res[-1][k], res[-2][k]), k


def _dense_case(H, W, V):
    from tests import util
    from scaffold import scene
    coarse = scene.make_gaussians(10, 16, opacity="A", seed=4)
    dense, init = scene.make_dense_params(coarse, per_vertex=4, seed=1)
    n = dense['dense_means3D'].shape[0]
    gen = torch.Generator().manual_seed(11)
    dense['dense_log_scales'] = dense['dense_log_scales'] + torch.randn(n, 3, generator=gen) * 0.3       # (as in _geometry_case)
    dense['dense_unnorm_rotations'] = torch.nn.functional.normalize(dense['dense_unnorm_rotations'] + torch.randn(n, 4, generator=gen) * 0.3)
    cams = util.to_device(scene.camera_rig(H, W, n_views=V), "cuda")
    g = torch.Generator().manual_seed(6)
    dataset = [{'cam': cams[i], 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': i, 'mask': None} for i in range(V)]
    lrs = {'dense_means3D': 0.0, 'dense_unnorm_rotations': 0.001, 'dense_logit_opacities': 0.0, 'dense_log_scales': 0.0,
           'dense_rgb_colors': 0.0025}                     # train.py:281-285
    n = dense['dense_means3D'].shape[0]
    frozen = torch.zeros(n, dtype=torch.bool)
    frozen[::6] = True                                     # static | dynamic | mouth_inner rows (train.py:732-734)
    return dense, init, dataset, lrs, frozen.cuda()


def _dense_params(dense):
    params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in dense.items()}
    params['dense_means3D'].requires_grad_(False)          # train.py:259-261: a plain tensor, never optimised
    return params


def test_texture_loop_hand_chained_equals_autograd_and_torch_pieces():
    """train.py:729-741: pins on dense_rgb_colors BEFORE each render, get_loss_dense(use_mask=False) (no affine, + 0.02 soft
    colour), backward, Adam - loop.optimise_dense_views chained by hand == through autograd bit for bit, and both follow the same
    loop built from torch ops.  The pinned rows end with the LAST step's values (the reference pins before the render only)."""
    from oracle import loss_oracle
    from scaffold import reference_boundary as boundary
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W, V = 64, 80, 3
    dense, init, dataset, lrs, frozen = _dense_case(H, W, V)
    res = []
    for mode in ("explicit", "autograd", "torch"):
        params = _dense_params(dense)
        variables = {'dense_init_colors': init.clone().cuda()}
        mx = torch.zeros(params['dense_means3D'].shape[0], device="cuda")
        if mode == "torch":
            opt = torch.optim.Adam(_groups(params, lrs), lr=0.0, eps=1e-15)

            def pins(params=params):
                params['dense_rgb_colors'][frozen] = 0.0
            losses = loop.optimise_dense_views(params, variables, dataset, opt, n_iters=8, seed=2, max_2D_radius=mx, pre_iteration=pins,
                                               loss_fn=loss_oracle.photometric_loss_torch, soft_color_fn=loss_oracle.soft_color_torch)
        else:
            opt = FusedAdamPins(_groups(params, lrs), eps=1e-15)
            opt.set_pin('dense_rgb_colors', frozen, 0.0)
            losses = loop.optimise_dense_views(params, variables, dataset, opt, n_iters=8, seed=2, max_2D_radius=mx,
                                               explicit=(mode == "explicit"))
            assert all(p.grad is None for p in params.values())
        res.append(({k: v.detach().clone() for k, v in params.items()}, torch.stack(losses), mx))
    (pe, le, me), (pa, la, ma), (pt, lt, mt) = res
    assert torch.equal(le, la) and torch.equal(me, ma)
    for k in pe:
        assert torch.equal(pe[k], pa[k]), (k, (pe[k] - pa[k]).abs().max())
    assert torch.allclose(le, lt, atol=2e-5), (le, lt)
    assert torch.equal(me, mt) and me.max() > 0
    for k in pe:
        moved = (pt[k] - dense[k].cuda()).abs()
        ok = ((pe[k] - pt[k]).abs() <= 0.02 * moved + 3e-5)
        assert ok.float().mean() > 0.97, (k, ok.float().mean())
    assert torch.equal(pe['dense_means3D'], dense['dense_means3D'].cuda())              # not trainable
    assert torch.equal(pe['dense_logit_opacities'], dense['dense_logit_opacities'].cuda())   # lr 0
    assert (pe['dense_rgb_colors'][frozen] != 0).any() and (pt['dense_rgb_colors'][frozen] != 0).any()
    assert (pe['dense_rgb_colors'][frozen].abs() < 0.01).all()                          # one step of lr 0.0025 away from the pin
    assert (pe['dense_unnorm_rotations'] != dense['dense_unnorm_rotations'].cuda()).any()


def test_graphed_views_on_the_masked_branch_and_the_texture_loop():
    """loop.GraphedViews records the branches train.py runs: later-frame masked targets (with load_frame() for the next frame's
    images) and the texture iteration (pins before the render, soft colour).  Replays == the eager hand-chained loops bit for bit."""
    import topo4d_amd
    from scaffold import scene
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W, V = 64, 80, 3
    p0, frame, lrs = _geometry_case(H, W, V)
    inner = scene.parsing_colormap_bgr(14)[[scene.PARSING_LABELS.index("inner_mouth")]]
    frames = [frame(1), frame(2)]
    schedule = [0, 2, 1, 1, 0, 2]

    def run_geometry(graphed):
        params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
        opt = FusedAdamPins(_groups(params, lrs), eps=1e-15, capturable=graphed)
        losses = []
        data = [[dict(e) for e in f] for f in frames]
        if graphed:
            gv = loop.GraphedViews(params, data[0], opt, use_mask=True, is_initial_timestep=False, label_colors=inner)
            assert gv.explicit
        topo4d_amd.set_sync_mode("lazy")
        try:
            for t in range(2):
                if graphed and t:
                    gv.load_frame(data[t])
                if not graphed:
                    loop.prepare_masked_targets(data[t], inner)
                    cam_grads = {k: torch.zeros_like(params[k]) for k in ('cam_m', 'cam_c')}
                    opt.clear_grad = {'cam_m', 'cam_c'}
                for c in schedule:
                    if graphed:
                        losses.append(gv.step(c).clone())
                    else:
                        l, _, grads, _, _ = loop.explicit_iteration(params, data[t][c], cam_grads, target=data[t][c]['masked_im'])
                        for k, gr in grads.items():
                            params[k].grad = gr
                        opt.step(); opt.zero_grad(set_to_none=True)
                        losses.append(l.clone())
            if graphed:
                gv.check()
        finally:
            topo4d_amd.set_sync_mode("checked")
        return {k: v.detach().clone() for k, v in params.items()}, torch.stack(losses)

    pg, lg = run_geometry(True)
    pe, le = run_geometry(False)
    assert torch.equal(lg, le), (lg, le)
    for k in pg:
        assert torch.equal(pg[k], pe[k]), (k, (pg[k] - pe[k]).abs().max())
    assert not torch.equal(lg[:6], lg[6:])                 # the second frame's images really arrived

    dense, init, dataset, dlrs, frozen = _dense_case(H, W, V)

    def run_dense(graphed):
        params = _dense_params(dense)
        variables = {'dense_init_colors': init.clone().cuda()}
        opt = FusedAdamPins(_groups(params, dlrs), eps=1e-15, capturable=graphed)
        opt.set_pin('dense_rgb_colors', frozen, 0.0)
        losses = []
        if graphed:
            gv = loop.GraphedViews(params, dataset, opt, dense=True, variables=variables)
        topo4d_amd.set_sync_mode("lazy")
        try:
            for c in schedule:
                if graphed:
                    losses.append(gv.step(c).clone())
                else:
                    opt.apply_pins(('dense_rgb_colors',))
                    l, _, grads, _, _ = loop.explicit_iteration(params, dataset[c], dense=True, soft_color=(variables['dense_init_colors'], 0.02))
                    for k, gr in grads.items():
                        params[k].grad = gr
                    opt.step(pins=False); opt.zero_grad(set_to_none=True)
                    losses.append(l.clone())
            if graphed:
                gv.check()
        finally:
            topo4d_amd.set_sync_mode("checked")
        return {k: v.detach().clone() for k, v in params.items()}, torch.stack(losses)

    pg, lg = run_dense(True)
    pe, le = run_dense(False)
    assert torch.equal(lg, le), (lg, le)
    for k in pg:
        assert torch.equal(pg[k], pe[k]), (k, (pg[k] - pe[k]).abs().max())
    assert (pg['dense_rgb_colors'][frozen] != 0).any()
    with pytest.raises(ValueError):
        loop.GraphedViews(_dense_params(dense), dataset, FusedAdamPins(_groups(_dense_params(dense), dlrs), capturable=True), dense=True)


def test_graphed_texture_loop_follows_update_dense_states_into_the_next_frame():
    """the reference RE-BINDS variables['dense_init_colors'] and params['dense_means3D'] to new tensors in every later frame
    (update_dense_states, train.py:498-507) while a recorded iteration reads the storage they had at capture time.  A replay
    with a re-bound entry is refused; load_frame copies the new contents into the recorded buffers (and binds the entries back):
    two frames of graphed texture iterations == the same two frames chained by hand, bit for bit."""
    import topo4d_amd
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W, V = 96, 128, 3
    dense, init, dataset, dlrs, frozen = _dense_case(H, W, V)
    g = torch.Generator().manual_seed(77)
    dataset2 = [dict(e, im=torch.rand(3, H, W, generator=g).cuda()) for e in dataset]
    shift = (torch.randn(dense['dense_means3D'].shape, generator=g) * 0.001).cuda()
    schedule = [0, 2, 1, 1]

    def update_dense_states(params, variables):             # what train.py:498-507 does to the two dict entries: NEW tensors
        variables['dense_init_colors'] = params['dense_rgb_colors'].clone().detach()
        params['dense_means3D'] = (params['dense_means3D'].detach() + shift).clone()

    def run(graphed):
        params = _dense_params(dense)
        variables = {'dense_init_colors': init.clone().cuda()}
        opt = FusedAdamPins(_groups(params, dlrs), eps=1e-15, capturable=graphed)
        opt.set_pin('dense_rgb_colors', frozen, 0.0)
        losses = []
        if graphed:
            gv = loop.GraphedViews(params, dataset, opt, dense=True, variables=variables)
        topo4d_amd.set_sync_mode("lazy")
        try:
            for t, data in enumerate((dataset, dataset2)):
                if t:
                    update_dense_states(params, variables)
                    if graphed:
                        with pytest.raises(RuntimeError, match="load_frame"):
                            gv.step(0)
                        gv.load_frame(data)
                        assert variables['dense_init_colors'].data_ptr() == gv._rebound[0][2].data_ptr()
                for c in schedule:
                    if graphed:
                        losses.append(gv.step(c).clone())
                    else:
                        opt.apply_pins(('dense_rgb_colors',))
                        l, _, grads, _, _ = loop.explicit_iteration(params, data[c], dense=True, soft_color=(variables['dense_init_colors'], 0.02))
                        for k, gr in grads.items():
                            params[k].grad = gr
                        opt.step(pins=False); opt.zero_grad(set_to_none=True)
                        losses.append(l.clone())
            if graphed:
                gv.check()
        finally:
            topo4d_amd.set_sync_mode("checked")
        return {k: v.detach().clone() for k, v in params.items()}, torch.stack(losses), variables['dense_init_colors'].clone()

    pg, lg, ig = run(True)
    pe, le, ie = run(False)
    assert torch.equal(lg, le), (lg, le)
    assert torch.equal(ig, ie) and not torch.equal(ie, init.cuda())
    for k in pg:
        assert torch.equal(pg[k], pe[k]), (k, (pg[k] - pe[k]).abs().max())
    assert not torch.equal(lg[:4], lg[4:])


def test_texture_loop_pins_must_be_cleared_before_the_next_geometry_loop():
    """the reference writes the frozen rows of dense_rgb_colors before the texture loop's renders
    ONLY (train.py:731-734); `optimizer.step()` of the geometry loop applies EVERY registered pin.  With `clear_pin` after the texture
    loop the next frame's geometry steps leave dense_rgb_colors alone - what update_dense_states (train.py:502) then clones is what the
    reference clones; without it the rows read zero."""
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W, V = 64, 64, 2
    dense, init, dataset, dlrs, frozen = _dense_case(H, W, V)
    geo = {'means3D': torch.nn.Parameter(torch.randn(50, 3).cuda())}
    for clear in (True, False):
        params = _dense_params(dense)
        allp = dict(params, **geo)
        opt = FusedAdamPins(_groups(allp, dict(dlrs, means3D=1e-3)), eps=1e-15)
        variables = {'dense_init_colors': init.clone().cuda()}
        opt.set_pin('dense_rgb_colors', frozen, 0.0)
        loop.optimise_dense_views(params, variables, dataset, opt, n_iters=3, seed=0)
        after_texture = params['dense_rgb_colors'].detach().clone()
        assert (after_texture[frozen] != 0).any()                    # the pinned rows hold the last step's values, as in the reference
        if clear:
            opt.clear_pin('dense_rgb_colors')
        geo['means3D'].grad = torch.randn(50, 3).cuda()              # a geometry step of the next frame: only means3D has a gradient
        opt.step(); opt.zero_grad(set_to_none=True)
        if clear:
            assert torch.equal(params['dense_rgb_colors'].detach(), after_texture)
        else:
            assert not params['dense_rgb_colors'].detach()[frozen].any()


def test_texture_iteration_at_the_texture_pass_size():
    """HOT LOOP 2 at its real size: P = 10^6 dense Gaussians, one 4096 x 3008 view (helpers.py:608-609, README "4K images"): two
    iterations of loop.optimise_dense_views chained by hand == through autograd bit for bit; nothing overflows (the soft-colour
    kernel's grid, the one-view loss at 12 M pixels, the pair arena of a 48,128-tile view)."""
    from tests import util
    from scaffold import scene
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W = 3008, 4096
    p = scene.make_gaussians(1000, 1000, opacity="A", seed=0)
    cam = util.to_device(scene.camera_rig(H, W, n_views=24)[12:13], "cuda")[0]
    g = torch.Generator().manual_seed(2)
    dataset = [{'cam': cam, 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': 0, 'mask': None}]
    lrs = {'dense_means3D': 0.0, 'dense_unnorm_rotations': 0.001, 'dense_logit_opacities': 0.0, 'dense_log_scales': 0.0, 'dense_rgb_colors': 0.0025}
    frozen = torch.zeros(1000000, dtype=torch.bool, device="cuda"); frozen[::7] = True
    res = []
    for explicit in (True, False):
        params = {"dense_" + k: torch.nn.Parameter(v.clone().cuda()) for k, v in p.items()}
        params['dense_means3D'].requires_grad_(False)
        variables = {'dense_init_colors': params['dense_rgb_colors'].detach().clone()}
        opt = FusedAdamPins(_groups(params, lrs), eps=1e-15)
        opt.set_pin('dense_rgb_colors', frozen, 0.0)
        mx = torch.zeros(1000000, device="cuda")
        losses = loop.optimise_dense_views(params, variables, dataset, opt, n_iters=2, seed=0, max_2D_radius=mx, explicit=explicit)
        res.append(({k: v.detach().clone() for k, v in params.items()}, torch.stack(losses), mx))
        del params, opt
        torch.cuda.empty_cache()
    (pe, le, me), (pa, la, ma) = res
    assert torch.isfinite(le).all() and torch.equal(le, la) and torch.equal(me, ma) and me.max() > 0
    for k in pe:
        assert torch.equal(pe[k], pa[k]), (k, (pe[k] - pa[k]).abs().max())
    assert (pe['dense_rgb_colors'][frozen] != 0).any() and (pe['dense_rgb_colors'][~frozen] != p['rgb_colors'].cuda()[~frozen]).any()
    # ... and replayed from a HIP graph (the forward of this launch is five kernels since its long tiles are rendered parallel along
    # depth, the backward two: all of them must be capturable): the same parameters after the same two iterations
    params = {"dense_" + k: torch.nn.Parameter(v.clone().cuda()) for k, v in p.items()}
    params['dense_means3D'].requires_grad_(False)
    variables = {'dense_init_colors': params['dense_rgb_colors'].detach().clone()}
    opt = FusedAdamPins(_groups(params, lrs), eps=1e-15, capturable=True)
    opt.set_pin('dense_rgb_colors', frozen, 0.0)
    gv = loop.GraphedViews(params, dataset, opt, dense=True, variables=variables)
    lg = [gv.step(0).clone() for _ in range(2)]
    gv.check()
    assert torch.equal(torch.stack(lg), le)
    for k in pe:
        assert torch.equal(params[k].detach(), pe[k]), (k, (params[k].detach() - pe[k]).abs().max())
