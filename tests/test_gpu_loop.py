"""GPU: fused Adam + pins vs torch.optim.Adam, and the per-view optimisation loop (train.py:661-700 shape) with the fused
pieces vs the same loop built from torch ops."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(params, lrs):
    return [{'params': [v], 'name': k, 'lr': lrs[k]} for k, v in params.items()]


def test_fused_adam_matches_torch_adam_and_pins():
    from topo4d_amd.optim import FusedAdamPins
    g = torch.Generator().manual_seed(0)
    shapes = {'means3D': (1000, 3), 'rgb_colors': (1000, 3), 'unnorm_rotations': (1000, 4), 'logit_opacities': (1000, 1),
              'log_scales': (1000, 3), 'cam_m': (24, 3)}
    lrs = {'means3D': 1.6e-5, 'rgb_colors': 0.0025, 'unnorm_rotations': 0.001, 'logit_opacities': 0.0, 'log_scales': 0.001,
           'cam_m': 1e-4}
    init = {k: torch.randn(*s, generator=g) for k, s in shapes.items()}
    pa = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in init.items()}
    pb = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in init.items()}
    ref = torch.optim.Adam(_groups(pa, lrs), lr=0.0, eps=1e-15)
    mine = FusedAdamPins(_groups(pb, lrs), lr=0.0, eps=1e-15)
    static = torch.arange(0, 1000, 7).cuda()
    mouth = torch.zeros(1000, dtype=torch.bool).cuda(); mouth[100:140] = True
    static_verts = pa['means3D'][static].clone().detach()
    mine.set_pin('means3D', static, static_verts)
    mine.set_pin('rgb_colors', mouth, 0.0)
    mine.set_pin('log_scales', mouth, float(np.log(0.01)))
    for it in range(6):
        for k in shapes:
            if k == 'cam_m' and it % 2:
                continue                                           # a parameter without gradient this iteration
            gr = torch.randn(*shapes[k], generator=g).cuda() * (10.0 if k == 'means3D' else 1.0)
            pa[k].grad = gr.clone(); pb[k].grad = gr.clone()
        if it == 3:
            for grp in ref.param_groups + mine.param_groups:       # helpers.update_optimizer
                if grp['name'] == 'rgb_colors':
                    grp['lr'] = 0.00025
        ref.step(); ref.zero_grad(set_to_none=True)
        with torch.no_grad():                                      # train.py:676-680 style freezes
            pa['means3D'][static] = static_verts
            pa['rgb_colors'][mouth] = torch.zeros_like(pa['rgb_colors'][mouth])
            pa['log_scales'][mouth] = float(np.log(0.01))
        mine.step(); mine.zero_grad(set_to_none=True)
        for k in shapes:
            assert torch.allclose(pa[k], pb[k], rtol=2e-6, atol=1e-7), (it, k, (pa[k] - pb[k]).abs().max())
    assert torch.equal(pb['means3D'][static], static_verts)


def test_capturable_adam_follows_a_tensor_that_changes_its_size():
    """capturable mode keeps one device step counter per 256-element workgroup, laid out by the tensors' sizes.  A group's
    tensor replaced by a LARGER one (the cat_params_to_optimizer pattern of external.py) used to write counters out of bounds and to
    shift every later tensor's.  Now the array is re-laid-out (every tensor keeps its count) and the C ABI checks its length:
    three steps, a resize, three more steps == the same on the host-counted optimiser, bit for bit."""
    import ctypes as C
    from topo4d_amd import _lib
    from topo4d_amd.optim import FusedAdamPins
    g = torch.Generator().manual_seed(3)
    shapes = {'a': (700, 3), 'b': (300, 4), 'c': (1000, 1)}
    lrs = {'a': 1e-3, 'b': 2e-3, 'c': 5e-4}
    init = {k: torch.randn(*s, generator=g) for k, s in shapes.items()}
    extra = torch.randn(900, 3, generator=g)
    grads = [{k: torch.randn(*(s if not (it >= 3 and k == 'a') else (1600, 3)), generator=g) for k, s in shapes.items()} for it in range(6)]
    res = []
    for capturable in (True, False):
        p = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in init.items()}
        opt = FusedAdamPins(_groups(p, lrs), eps=1e-15, capturable=capturable)
        for it in range(6):
            if it == 3:                                            # cat_params_to_optimizer: a new, larger tensor with extended moments
                old = p['a']
                new = torch.nn.Parameter(torch.cat([old.detach(), extra.cuda()], 0))
                st = opt.state.pop(old)
                st["exp_avg"] = torch.cat([st["exp_avg"], torch.zeros(900, 3, device="cuda")], 0)
                st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"], torch.zeros(900, 3, device="cuda")], 0)
                opt.state[new] = st
                opt.param_groups[0]["params"] = [new]
                p['a'] = new
            for k in p:
                if not (k == 'c' and it == 1):                     # a tensor that skips a step keeps its own count
                    p[k].grad = grads[it][k].cuda()
            opt.step(); opt.zero_grad(set_to_none=True)
        res.append(({k: v.detach().clone() for k, v in p.items()}, opt.steps()))
    (pc, sc), (ph, sh) = res
    assert sc == sh == [6, 6, 5]
    for k in pc:
        assert torch.equal(pc[k], ph[k]), (k, (pc[k] - ph[k]).abs().max())
    # the ABI refuses a counter array of another layout outright
    lib = _lib.load()
    t = torch.zeros(512, 3, device="cuda")
    arr = (_lib.T4DAdamTensor * 1)(_lib.T4DAdamTensor(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, None, 512, 3, 1e-3, 1, 0))
    cnt = torch.zeros(6, dtype=torch.int32, device="cuda"); lr = torch.zeros(1, device="cuda")
    assert lib.t4d_adam_step_counters(arr, 1) == 6
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.t4d_adam_pin_step_graph(arr, 1, 0.9, 0.999, 1e-15, cnt.data_ptr(), 5, lr.data_ptr(), stream) == _lib.T4D_ERR_ARG
    assert lib.t4d_adam_pin_step_graph(arr, 1, 0.9, 0.999, 1e-15, cnt.data_ptr(), 6, lr.data_ptr(), stream) == 0
    torch.cuda.synchronize()


def test_per_view_loop_fused_vs_torch_pieces():
    import topo4d_amd
    from tests import util
    from scaffold import scene
    from oracle import loss_oracle
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H = W = 64
    p0 = scene.make_gaussians(12, 20, opacity="B", seed=3)
    p0['log_scales'] = p0['log_scales'] + torch.randn(240, 3, generator=torch.Generator().manual_seed(9)) * 0.3
    p0['cam_m'] = torch.zeros(3, 3); p0['cam_c'] = torch.zeros(3, 3)
    cams = util.to_device(scene.camera_rig(H, W, n_views=3), "cuda")
    g = torch.Generator().manual_seed(5)
    dataset = [{'cam': cams[i], 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': i} for i in range(3)]
    lrs = {'means3D': 1.6e-4, 'rgb_colors': 0.0025, 'unnorm_rotations': 0.001, 'logit_opacities': 0.0, 'log_scales': 0.001,
           'cam_m': 1e-4, 'cam_c': 1e-4}
    results = []
    for fused in (True, False):
        params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
        groups = _groups(params, lrs)
        opt = FusedAdamPins(groups, eps=1e-15) if fused else torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        mx = torch.zeros(240, device="cuda")
        losses = loop.optimise_views(params, dataset, opt, n_iters=7, seed=1, max_2D_radius=mx,
                                     loss_fn=None if fused else loss_oracle.photometric_loss_torch)
        results.append(({k: v.detach().clone() for k, v in params.items()}, torch.stack(losses), mx))
    (pf, lf, mf), (pt, lt, mt) = results
    assert torch.allclose(lf, lt, atol=2e-5), (lf, lt)
    assert lf[-1] < lf[0] + 1e-3
    # Adam divides by sqrt(v): an entry whose gradient is round-off noise moves by +-lr per step in either implementation,
    # so parameters are compared where the update is well-conditioned (|total change| clearly above the noise floor).
    for k in pf:
        moved = (pt[k] - p0[k].cuda()).abs()
        ok = ((pf[k] - pt[k]).abs() <= 0.02 * moved + 3e-5)
        assert ok.float().mean() > 0.97, (k, ok.float().mean(), (pf[k] - pt[k]).abs().max())
    assert torch.equal(mf, mt) and mf.max() > 0


def test_fused_activations_match_torch():
    """t4d_activate_forward/backward vs torch's normalize / sigmoid / exp and their autograd (helpers.py:95-97)."""
    from scaffold import reference_boundary as boundary, scene
    torch.manual_seed(3)
    p = scene.make_gaussians(12, 20, opacity="B", seed=5)
    p["unnorm_rotations"] = p["unnorm_rotations"] * (0.2 + 3 * torch.rand(p["unnorm_rotations"].shape[0], 1))   # not unit length
    p["unnorm_rotations"][3] = 0.0                                                                      # the eps-clamped branch
    pa = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p.items()}
    pb = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p.items()}
    from topo4d_amd.boundary import params2rendervar_fused
    ra, rb = boundary.params2rendervar(pa), params2rendervar_fused(pb)
    for k in ("rotations", "opacities", "scales"):
        assert rb[k].shape == ra[k].shape
        torch.testing.assert_close(rb[k], ra[k], rtol=2e-6, atol=1e-7)
    g = {k: torch.randn_like(ra[k]) for k in ("rotations", "opacities", "scales")}
    sum((ra[k] * g[k]).sum() for k in g).backward()
    sum((rb[k] * g[k]).sum() for k in g).backward()
    for k in ("unnorm_rotations", "logit_opacities", "log_scales"):
        torch.testing.assert_close(pb[k].grad, pa[k].grad, rtol=2e-5, atol=1e-6)
    # only some inputs need a gradient
    pc = {k: v.detach().clone().requires_grad_(k == "log_scales") for k, v in pb.items()}
    rc = params2rendervar_fused(pc)
    (rc["scales"] * g["scales"]).sum().backward()
    torch.testing.assert_close(pc["log_scales"].grad, pa["log_scales"].grad, rtol=2e-5, atol=1e-6)
    assert pc["unnorm_rotations"].grad is None


def test_graphed_views_replay_equals_the_eager_loop():
    """loop.GraphedViews (one HIP graph per camera: activations -> render -> loss -> backward -> Adam + pins) against the
    same iterations issued eagerly, same camera schedule, changing a learning rate on the way."""
    import topo4d_amd
    from tests import util
    from scaffold import scene
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H = W = 64
    p0 = scene.make_gaussians(12, 20, opacity="B", seed=3)
    p0['log_scales'] = p0['log_scales'] + torch.randn(240, 3, generator=torch.Generator().manual_seed(9)) * 0.3
    p0['cam_m'] = torch.zeros(3, 3); p0['cam_c'] = torch.zeros(3, 3)
    cams = util.to_device(scene.camera_rig(H, W, n_views=3), "cuda")
    g = torch.Generator().manual_seed(5)
    dataset = [{'cam': cams[i], 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': i} for i in range(3)]
    lrs = {'means3D': 1.6e-4, 'rgb_colors': 0.0025, 'unnorm_rotations': 0.001, 'logit_opacities': 0.0, 'log_scales': 0.001,
           'cam_m': 1e-4, 'cam_c': 1e-4}
    schedule = [0, 2, 1, 1, 0, 2, 2, 0, 1, 0]
    pins = torch.arange(0, 240, 5).cuda()

    def run(graphed):
        params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
        opt = FusedAdamPins(_groups(params, lrs), eps=1e-15, capturable=graphed)
        opt.set_pin('means3D', pins, params['means3D'][pins].detach().clone())
        losses = []
        if graphed:
            gv = loop.GraphedViews(params, dataset, opt)
        topo4d_amd.set_sync_mode("lazy")
        try:
            for it, c in enumerate(schedule):
                if it == 5:
                    for grp in opt.param_groups:                    # helpers.update_optimizer between stages
                        if grp['name'] == 'rgb_colors':
                            grp['lr'] = 0.00025
                if graphed:
                    losses.append(gv.step(c).clone())
                else:
                    l, _, _ = loop.photometric_iteration(params, dataset[c])
                    l.backward()
                    opt.step(); opt.zero_grad(set_to_none=True)
                    losses.append(l.detach().clone())
            if graphed:
                gv.check()
                assert opt.steps()[0] == len(schedule)
        finally:
            topo4d_amd.set_sync_mode("checked")
        return {k: v.detach().clone() for k, v in params.items()}, torch.stack(losses)

    # capacity for the eager run is learned by the graphed run's warm-up (same scene, same cameras)
    pg, lg = run(True)
    pe, le = run(False)
    assert torch.allclose(lg, le, rtol=1e-6, atol=1e-7), (lg, le)
    for k in pg:
        assert torch.allclose(pg[k], pe[k], rtol=2e-6, atol=1e-7), (k, (pg[k] - pe[k]).abs().max())
    assert torch.equal(pg['means3D'][pins], p0['means3D'].cuda()[pins])


def test_explicit_graphed_iteration_is_the_autograd_capture_bit_for_bit():
    """loop.GraphedViews chains the iteration by hand when it can (explicit_iteration: the same five library calls without the
    fill / copy / multiply launches autograd puts around them; the camera-affine gradients land in persistent buffers that the
    Adam step clears; the forward's status goes to pinned host memory without a copy node).  Parameters, optimiser state and
    losses after a schedule of replays - with a learning-rate change on the way - must equal the autograd capture's bit for bit."""
    import topo4d_amd
    from tests import util
    from scaffold import scene
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W = 64, 80
    p0 = scene.make_gaussians(12, 20, opacity="B", seed=3)
    p0['log_scales'] = p0['log_scales'] + torch.randn(240, 3, generator=torch.Generator().manual_seed(9)) * 0.3
    p0['cam_m'] = torch.randn(3, 3, generator=torch.Generator().manual_seed(2)) * 0.05
    p0['cam_c'] = torch.randn(3, 3, generator=torch.Generator().manual_seed(3)) * 0.05
    cams = util.to_device(scene.camera_rig(H, W, n_views=3), "cuda")
    g = torch.Generator().manual_seed(5)
    dataset = [{'cam': cams[i], 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': i} for i in range(3)]
    lrs = {'means3D': 1.6e-4, 'rgb_colors': 0.0025, 'unnorm_rotations': 0.001, 'logit_opacities': 0.05, 'log_scales': 0.001,
           'cam_m': 1e-3, 'cam_c': 1e-3}
    schedule = [0, 2, 1, 1, 0, 2, 2, 0, 1, 0, 1, 2]
    pins = torch.arange(0, 240, 5).cuda()

    def run(explicit):
        params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
        opt = FusedAdamPins(_groups(params, lrs), eps=1e-15, capturable=True)
        opt.set_pin('means3D', pins, params['means3D'][pins].detach().clone())
        gv = loop.GraphedViews(params, dataset, opt, explicit=explicit)
        assert gv.explicit == explicit and not opt.clear_grad
        losses = []
        for it, c in enumerate(schedule):
            if it == 5:
                for grp in opt.param_groups:
                    if grp['name'] == 'rgb_colors':
                        grp['lr'] = 0.00025
            losses.append(gv.step(c).clone())
        gv.check()
        assert opt.steps()[0] == len(schedule)
        state = {k: (opt.state[v]['exp_avg'].clone(), opt.state[v]['exp_avg_sq'].clone()) for k, v in params.items()}
        return {k: v.detach().clone() for k, v in params.items()}, torch.stack(losses), state, gv

    pe, le, se, gve = run(True)
    pa, la, sa, _ = run(False)
    assert torch.equal(le, la)
    for k in pe:
        assert torch.equal(pe[k], pa[k]), (k, (pe[k] - pa[k]).abs().max())
        assert torch.equal(se[k][0], sa[k][0]) and torch.equal(se[k][1], sa[k][1]), k
    assert (pe['cam_m'] != p0['cam_m'].cuda()).any() and (pe['logit_opacities'] != p0['logit_opacities'].cuda()).any()
    assert gve._status_host is not None                 # one small view per launch: the status needed no copy node
    assert len(gve.means2D_grads) == 3 and gve.means2D_grads[0].shape == (240, 3)


def test_hand_chained_iteration_composes_with_extra_loss_terms():
    """Every iteration of the real loop carries topology regularisers on the parameters (train.py:330-368; out of scope as kernels).
    The hand-chained iteration takes them as `extra_loss`: differentiated through autograd on their own, their gradients added to
    the render's - one iteration's gradients and loss equal the all-autograd iteration's to rounding, the eager loop follows the
    autograd loop, and GraphedViews records it (replays == the eager hand-chained loop)."""
    import topo4d_amd
    from tests import util
    from scaffold import scene
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W = 64, 80
    p0 = scene.make_gaussians(12, 20, opacity="B", seed=3)
    p0['log_scales'] = p0['log_scales'] + torch.randn(240, 3, generator=torch.Generator().manual_seed(9)) * 0.3
    p0['cam_m'] = torch.randn(3, 3, generator=torch.Generator().manual_seed(2)) * 0.05
    p0['cam_c'] = torch.randn(3, 3, generator=torch.Generator().manual_seed(3)) * 0.05
    cams = util.to_device(scene.camera_rig(H, W, n_views=3), "cuda")
    g = torch.Generator().manual_seed(5)
    dataset = [{'cam': cams[i], 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': i} for i in range(3)]
    lrs = {'means3D': 1.6e-4, 'rgb_colors': 0.0025, 'unnorm_rotations': 0.001, 'logit_opacities': 0.05, 'log_scales': 0.001,
           'cam_m': 1e-3, 'cam_c': 1e-3}
    nb = torch.randint(0, 240, (240, 4), generator=g).cuda()
    prev = p0['means3D'].cuda()

    def regularisers(p, rv):                                   # iso / rot / scale terms shaped like train.py:340-361
        off = rv['means3D'][nb] - rv['means3D'][:, None]
        iso = (torch.sqrt((off ** 2).sum(-1) + 1e-20) - torch.sqrt(((prev[nb] - prev[:, None]) ** 2).sum(-1) + 1e-20)).abs().mean()
        rot = ((rv['rotations'][nb] - rv['rotations'][:, None]) ** 2).sum(-1).mean()
        return 20.0 * iso + 20.0 * rot + 10.0 * rv['scales'].min(dim=1).values.sum() + 1e-3 * p['cam_m'].pow(2).sum()

    # one iteration: gradients and loss
    pa = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
    pb = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
    la, _, _ = loop.photometric_iteration(pa, dataset[1], extra_loss=regularisers)
    la.backward()
    lb, _, grads, _, _ = loop.explicit_iteration(pb, dataset[1], extra_loss=regularisers)
    assert abs(la.item() - lb.item()) <= 1e-6 * abs(la.item())
    for k in pa:
        scale = float(pa[k].grad.abs().max())
        assert scale > 0 and (grads[k] - pa[k].grad).abs().max() <= 2e-6 * scale, (k, (grads[k] - pa[k].grad).abs().max(), scale)
    # the loop, and the recorded loop
    res = []
    for mode in ("explicit", "autograd", "graphed"):
        params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
        opt = FusedAdamPins(_groups(params, lrs), eps=1e-15, capturable=(mode == "graphed"))
        if mode == "graphed":
            gv = loop.GraphedViews(params, dataset, opt, extra_loss=regularisers)
            assert gv.explicit
            topo4d_amd.set_sync_mode("lazy")
            try:
                rng, todo, losses = __import__("random").Random(4), [], []
                for _ in range(8):
                    curr, todo = loop.get_batch(todo, dataset, rng)
                    losses.append(gv.step(curr['id']).clone())
                gv.check()
            finally:
                topo4d_amd.set_sync_mode("checked")
        else:
            losses = loop.optimise_views(params, dataset, opt, n_iters=8, seed=4, extra_loss=regularisers, explicit=(mode == "explicit"))
        res.append(({k: v.detach().clone() for k, v in params.items()}, torch.stack(losses)))
    (pe, le), (pa_, la_), (pg, lg) = res
    assert torch.allclose(le, la_, rtol=2e-5, atol=1e-7), (le, la_)
    assert torch.allclose(le, lg, rtol=2e-6, atol=1e-8), (le, lg)
    for k in pe:
        moved = (pa_[k] - p0[k].cuda()).abs()
        assert ((pe[k] - pa_[k]).abs() <= 0.02 * moved + 3e-5).float().mean() > 0.97, k
        assert ((pe[k] - pg[k]).abs() <= 0.02 * moved + 3e-5).float().mean() > 0.97, k


def test_explicit_eager_loop_is_the_autograd_loop_bit_for_bit():
    """loop.optimise_views chains every iteration by hand when it can (explicit_iteration); parameters, losses and the radius
    bookkeeping after a run must equal the autograd loop's bit for bit."""
    from tests import util
    from scaffold import scene
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    H, W = 64, 96
    p0 = scene.make_gaussians(12, 20, opacity="B", seed=3)
    p0['cam_m'] = torch.randn(3, 3, generator=torch.Generator().manual_seed(2)) * 0.05
    p0['cam_c'] = torch.randn(3, 3, generator=torch.Generator().manual_seed(3)) * 0.05
    cams = util.to_device(scene.camera_rig(H, W, n_views=3), "cuda")
    g = torch.Generator().manual_seed(5)
    dataset = [{'cam': cams[i], 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': i} for i in range(3)]
    lrs = {'means3D': 1.6e-4, 'rgb_colors': 0.0025, 'unnorm_rotations': 0.001, 'logit_opacities': 0.05, 'log_scales': 0.001,
           'cam_m': 1e-3, 'cam_c': 1e-3}
    res = []
    for explicit in (True, False):
        params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
        opt = FusedAdamPins(_groups(params, lrs), eps=1e-15)
        mx = torch.zeros(240, device="cuda")
        losses = loop.optimise_views(params, dataset, opt, n_iters=9, seed=4, max_2D_radius=mx, explicit=explicit)
        assert not opt.clear_grad and all(p.grad is None for p in params.values())
        res.append(({k: v.detach().clone() for k, v in params.items()}, torch.stack(losses), mx))
    (pe, le, me), (pa, la, ma) = res
    assert torch.equal(le, la) and torch.equal(me, ma)
    for k in pe:
        assert torch.equal(pe[k], pa[k]), (k, (pe[k] - pa[k]).abs().max())
    assert (pe['cam_c'] != p0['cam_c'].cuda()).any()


def test_explicit_frame_iteration_equals_autograd_over_all_views():
    """loop.explicit_frame_iteration (all cameras of a frame in one launch set, chained by hand) against the same frame through
    autograd: params2rendervar_fused -> rasterize_views -> photometric_loss(...).sum().backward().  Losses and every gradient -
    the camera affine's included - bit for bit."""
    from tests import util
    from scaffold import scene
    from topo4d_amd import loop, loss as t4d_loss, rasterize_views
    from topo4d_amd.boundary import params2rendervar_fused
    H, W, V = 64, 96, 5
    p0 = scene.make_gaussians(12, 20, opacity="B", seed=3)
    p0['cam_m'] = torch.randn(V, 3, generator=torch.Generator().manual_seed(2)) * 0.05
    p0['cam_c'] = torch.randn(V, 3, generator=torch.Generator().manual_seed(3)) * 0.05
    cams = util.to_device(scene.camera_rig(H, W, n_views=V), "cuda")
    g = torch.Generator().manual_seed(5)
    frame = [{'cam': cams[i], 'im': torch.rand(3, H, W, generator=g).cuda(), 'id': i} for i in range(V)]
    params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in p0.items()}
    l, radii, grads, _ = loop.explicit_frame_iteration(params, frame)
    rv = params2rendervar_fused(params)
    im, radii_a, _, _ = rasterize_views(cams, rv["means3D"], rv["means2D"], rv["opacities"], colors_precomp=rv["colors_precomp"],
                                        scales=rv["scales"], rotations=rv["rotations"])
    la = t4d_loss.photometric_loss(im, torch.stack([e['im'] for e in frame]), params['cam_m'], params['cam_c'])
    la.sum().backward()
    assert torch.equal(l, la.detach()) and torch.equal(radii, radii_a)
    for k, gr in grads.items():
        assert torch.equal(gr, params[k].grad), (k, (gr - params[k].grad).abs().max())
    assert grads['cam_m'].abs().max() > 0 and grads['means3D'].abs().max() > 0
    with pytest.raises(ValueError):
        loop.explicit_frame_iteration(params, [frame[0], frame[2]])


def test_raw_parameter_mode_of_the_rasterizer_is_params2rendervar_bit_for_bit():
    """T4D_FLAG_RAW_PARAMS (ViewBatch.raw_params): the rasterizer takes un-normalised quaternions, logit opacities and log scales
    (helpers.py:95-97) and returns their gradients.  Against t4d_activate_forward -> rasterizer -> t4d_activate_backward per
    view: outputs, radii and every gradient bit for bit - invisible Gaussians (zeros) and a zero quaternion (the eps-clamped
    branch of F.normalize) included."""
    from tests import util
    from scaffold import scene
    from topo4d_amd import ViewBatch, pack_views
    from topo4d_amd.boundary import activate_backward, activate_forward
    H, W, V = 96, 80, 3
    p = scene.make_gaussians(14, 22, opacity="B", seed=4)
    P = p['means3D'].shape[0]
    g = torch.Generator().manual_seed(8)
    p['unnorm_rotations'] = p['unnorm_rotations'] * (0.3 + 2 * torch.rand(P, 1, generator=g))
    p['unnorm_rotations'][5] = 0.0
    p['log_scales'] = p['log_scales'] + torch.randn(P, 3, generator=g) * 0.3
    p['means3D'][7] = torch.tensor([0.0, 0.0, 50.0])          # far outside every frustum
    cams = util.to_device(scene.camera_rig(H, W, n_views=V), "cuda")
    dev = torch.device("cuda")
    ur, lo, ls = (p[k].cuda().contiguous() for k in ('unnorm_rotations', 'logit_opacities', 'log_scales'))
    m3, rgb = p['means3D'].cuda(), p['rgb_colors'].cuda()
    dc = torch.randn(V, 3, H, W, generator=g).cuda()
    views = pack_views(cams, dev)
    # route 1: separate activation kernels
    rot, op, sc = activate_forward(ur, lo, ls)
    b1 = ViewBatch(views, H, W)
    o1 = b1.forward(m3, op, sc, rot, colors_precomp=rgb)
    g1 = b1.backward(dc)
    # route 2: raw parameters
    b2 = ViewBatch(views, H, W)
    b2.raw_params = True
    o2 = b2.forward(m3, lo, ls, ur, colors_precomp=rgb)
    g2 = b2.backward(dc)
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    for k in ('means3D', 'means2D', 'colors_precomp'):
        assert torch.equal(g1[k], g2[k]), k
    assert (o1[1] > 0).any() and (o1[1][:, 7] == 0).all()
    for v in range(V):
        d_ur, d_lo, d_ls = activate_backward(ur, op, sc, g1['rotations'][v].contiguous(), g1['opacities'][v].contiguous(),
                                             g1['scales'][v].contiguous())
        assert torch.equal(d_ur, g2['rotations'][v]), v
        assert torch.equal(d_lo, g2['opacities'][v]), v
        assert torch.equal(d_ls, g2['scales'][v]), v
    assert torch.isfinite(g2['rotations']).all() and g2['scales'].abs().max() > 0 and (g2['opacities'][:, 7] == 0).all()
