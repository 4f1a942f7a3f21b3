"""
GPU: (1) several PARAMETER SETS (frames) per launch - T4DProblem.views_per_param_set, ViewBatch(param_sets=S): what a rank of
BASELINE config 3's view-sharded job launches (its 3 of 24 cameras for several independent frames at once); (2) `scale_modifier`
other than 1 (field 6 of GaussianRasterizationSettings, helpers.py:73-86; helpers.py:79 passes 1.0, upstream scales cov3D and
dL/dscales by it).

Both go THROUGH THE C ABI (ViewBatch -> ctypes -> t4d_rasterize_*) against the C oracle and the float64 autograd oracle, with the
tolerances of tests/test_gpu_parity.py.
"""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_parity import check_grads, check_outputs

pytestmark = pytest.mark.gpu


def _frames(n_sets, n_lat, n_lon, sh_degree=None, seed=0):
    """n_sets frames whose per-Gaussian inputs ALL differ (a real sequence moves means3D; the ABI promises every input)."""
    from scaffold import reference_boundary as boundary, scene
    rvs = []
    for s in range(n_sets):
        params = scene.make_gaussians(n_lat, n_lon, opacity="B", sh_degree=sh_degree, seed=seed + 17 * s)
        params["means3D"] = scene.frame_displacement(params["means3D"], 5 * s, 64)
        rv = {k: v.detach() for k, v in boundary.params2rendervar(params).items()}
        if sh_degree is not None:
            rv["shs"] = params["shs"]
            del rv["colors_precomp"]
        rvs.append(rv)
    return rvs


def _render_sets(cams_per_set, rvs, dc, dd=None, da=None, raw=False):
    """One launch set: len(rvs) parameter sets x len(cams_per_set) views each (set-major view order)."""
    from topo4d_amd import ViewBatch, pack_views
    from topo4d_amd.rasterizer import _check_common
    dev = torch.device("cuda")
    H, W, smod, deg = _check_common(cams_per_set)
    S = len(rvs)
    views = pack_views(util.to_device(cams_per_set, dev), dev).repeat(S, 1).contiguous()
    batch = ViewBatch(views, H, W, smod, deg, param_sets=S)
    batch.raw_params = raw
    stack = lambda k: torch.stack([rv[k] for rv in rvs], 0).to(dev).contiguous() if k in rvs[0] else None
    color, radii, depth, alpha = batch.forward(stack("means3D"), stack("opacities"), stack("scales"), stack("rotations"),
                                               stack("colors_precomp"), stack("shs"), stack("cov3D_precomp"))
    dot = torch.empty(views.shape[0], device=dev)
    g = batch.backward(dc.to(dev), None if dd is None else dd.to(dev), None if da is None else da.to(dev), cotangent_dot=dot)
    out = dict(color=color.cpu().numpy(), radii=radii.cpu().numpy(), depth=depth.cpu().numpy(), alpha=alpha.cpu().numpy(),
               dot=dot.cpu().numpy())
    return out, {k: (v.cpu().numpy() if v is not None else None) for k, v in g.items()}, batch


@pytest.mark.parametrize("n_sets,n_cams,sh_degree", [(2, 3, None), (4, 2, None), (3, 1, None), (2, 8, 3), (3, 2, 3), (2, 2, 1)])
def test_parameter_sets_match_the_oracle_and_one_frame_launches(n_sets, n_cams, sh_degree, render_build, monkeypatch):
    from scaffold import scene
    H = W = 96
    V = n_sets * n_cams
    rvs = _frames(n_sets, 18, 28, sh_degree=sh_degree, seed=3)
    _, cams = util.make_scene(4, 4, H, W, n_cams, sh_degree=sh_degree)
    dc, dd, da = scene.output_cotangents(V, H, W, seed=4, depth_alpha=True)
    hip, hg, batch = _render_sets(cams, rvs, dc, dd, da)
    assert batch.prob.views_per_param_set == n_cams
    keys = ("means3D", "means2D", "opacities", "scales", "rotations", "shs" if sh_degree is not None else "colors_precomp")
    for s in range(n_sets):
        for c in range(n_cams):
            v = s * n_cams + c
            r, g = util.c_oracle_render(cams[c], rvs[s], dc[v], dd[v], da[v])
            np.testing.assert_array_equal(hip["radii"][v], r.radii)
            check_outputs(hip, r.color, r.depth, r.alpha, v)
            check_grads(hg, g, v, keys=keys)
    # bit for bit what the SAME view index of a one-frame launch of the same shape gives (the other views' parameters are
    # irrelevant to a view: here they are simply the same frame's)
    all_cams = [cams[c] for _ in range(n_sets) for c in range(n_cams)]
    if sh_degree == 3 and n_cams % 8 != 0:
        # the degree-3 SH backward that serves eight views from one fetch of the coefficient rows needs whole groups of eight views
        # per set; otherwise the general kernel runs - "the same launch shape" of a one-frame call is then the general kernel too
        monkeypatch.setenv("T4D_SH_BWD_PLAIN", "1")
    for s in range(n_sets):
        one, og, _ = util.hip_render(all_cams, rvs[s], dc, dd, da)
        for c in range(n_cams):
            v = s * n_cams + c
            for k in ("color", "depth", "alpha", "radii"):
                np.testing.assert_array_equal(hip[k][v], one[k][v])
            for k in keys:
                np.testing.assert_array_equal(hg[k][v], og[k][v])


def test_parameter_sets_with_raw_optimiser_parameters_and_cov3d():
    """T4D_FLAG_RAW_PARAMS (activations inside the rasterizer) and cov3D_precomp read their rows from the view's set too."""
    from oracle import torch_oracle as TO
    from scaffold import scene
    H = W = 80
    n_sets, n_cams = 2, 3
    V = n_sets * n_cams
    rvs = _frames(n_sets, 16, 24, seed=21)
    _, cams = util.make_scene(4, 4, H, W, n_cams)
    dc, _, _ = scene.output_cotangents(V, H, W, seed=22)
    from topo4d_amd.boundary import activate_backward, activate_forward
    raws, acts = [], []
    for rv in rvs:
        raw = dict(rv)
        raw["rotations"] = rv["rotations"] * torch.linspace(0.5, 2.0, rv["rotations"].shape[0])[:, None]   # un-normalised quaternions
        raw["opacities"] = torch.logit(rv["opacities"].clamp(1e-6, 1 - 1e-6))
        raw["scales"] = torch.log(rv["scales"])
        raws.append(raw)
        # what the library's own activation kernel makes of them (the arithmetic T4D_FLAG_RAW_PARAMS applies inside the rasterizer)
        rot, op, sc = activate_forward(raw["rotations"].cuda().contiguous(), raw["opacities"].cuda().contiguous(), raw["scales"].cuda().contiguous())
        acts.append(dict(rv, rotations=rot.cpu(), opacities=op.cpu(), scales=sc.cpu()))
    act, ag, _ = _render_sets(cams, acts, dc)
    hip, hg, _ = _render_sets(cams, raws, dc, raw=True)
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(hip[k], act[k])                 # bit for bit: same activated values, same kernels
    for k in ("means3D", "means2D", "colors_precomp"):
        np.testing.assert_array_equal(hg[k], ag[k])
    # the raw gradients are the activation's vector-Jacobian products of the activated ones, per view
    for s in range(n_sets):
        for c in range(n_cams):
            v = s * n_cams + c
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            d_rot, d_op, d_sc = activate_backward(raws[s]["rotations"].cuda().contiguous(), acts[s]["opacities"].cuda().contiguous(),
                                                  acts[s]["scales"].cuda().contiguous(), t(ag["rotations"][v]), t(ag["opacities"][v]), t(ag["scales"][v]))
            for name, want in (("rotations", d_rot), ("opacities", d_op), ("scales", d_sc)):
                w = want.cpu().numpy()
                assert np.abs(hg[name][v] - w).max() <= 2e-6 * max(np.abs(w).max(), 1e-30) + 1e-12, (name, v)
    covs = []
    for rv in rvs:
        R = TO.quat_to_rot(rv["rotations"].double())
        RS = R * rv["scales"].double()[:, None, :]
        Sg = RS @ RS.transpose(1, 2)
        c = dict(rv)
        c["cov3D_precomp"] = torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).float()
        del c["scales"], c["rotations"]
        covs.append(c)
    hip, hg, _ = _render_sets(cams, covs, dc)
    for s in range(n_sets):
        for c in range(n_cams):
            v = s * n_cams + c
            r, g = util.c_oracle_render(cams[c], covs[s], dc[v])
            check_outputs(hip, r.color, r.depth, r.alpha, v)
            check_grads(hg, g, v, keys=("means3D", "means2D", "opacities", "colors_precomp", "cov3D_precomp"))


def test_parameter_sets_argument_checks():
    from topo4d_amd import ViewBatch, pack_views
    dev = torch.device("cuda")
    _, cams = util.make_scene(4, 4, 64, 64, 3)
    views = pack_views(util.to_device(cams, dev), dev)
    with pytest.raises(ValueError):
        ViewBatch(views, 64, 64, param_sets=2)                      # 3 views are not a multiple of 2 sets
    rvs = _frames(3, 8, 8)
    b = ViewBatch(views, 64, 64, param_sets=3)
    d = lambda k: rvs[0][k].to(dev)
    with pytest.raises(ValueError):                                  # inputs without the set axis
        b.forward(d("means3D"), d("opacities"), d("scales"), d("rotations"), d("colors_precomp"))


def test_config3_rank_launch_equals_the_24_view_launches_of_its_frames():
    """BASELINE config 3, one rank of the 8-way view shard: cameras r::8 of EIGHT consecutive frames in one 24-view launch set.
    Per view - outputs, every gradient, the per-view loss scalar the backward emits - it equals, BIT FOR BIT, the 24-view launch
    of that frame on one GPU (same launch shape: 24 views x 1,024 tiles, whole-tile throughput kernels)."""
    from scaffold import reference_boundary as boundary, scene
    from topo4d_amd import ViewBatch, pack_views
    dev = torch.device("cuda")
    cfg = scene.CONFIGS["C2"]
    H, W, V = cfg["H"], cfg["W"], cfg["n_views"]
    params = scene.make_gaussians(cfg["n_lat"], cfg["n_lon"], opacity="A", seed=0)
    cams = scene.camera_rig(H, W, n_views=V, device=dev)
    dc_all, _, _ = scene.output_cotangents(V, H, W, seed=0)
    rank, world, n_frames = 5, 8, 8
    mine = list(range(rank, V, world))
    rvs = []
    for t in range(n_frames):
        p = dict(params)
        p["means3D"] = scene.frame_displacement(params["means3D"], t, 64)
        rvs.append({k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()})
    views = pack_views([cams[i] for i in mine], dev).repeat(n_frames, 1).contiguous()
    dc = dc_all[mine].to(dev).repeat(n_frames, 1, 1, 1).contiguous()
    b = ViewBatch(views, H, W, param_sets=n_frames)
    stack = lambda k: torch.stack([rv[k] for rv in rvs], 0).contiguous()
    color, radii, _, _ = b.forward(stack("means3D"), stack("opacities"), stack("scales"), stack("rotations"), stack("colors_precomp"))
    dot = torch.empty(views.shape[0], device=dev)
    g = b.backward(dc, cotangent_dot=dot)
    assert b.fetch_status().overflow == 0
    full = ViewBatch(pack_views(cams, dev), H, W)
    dcf = dc_all.to(dev).contiguous()
    for t in (0, 3, 7):
        rv = rvs[t]
        c1, r1, _, _ = full.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"])
        dot1 = torch.empty(V, device=dev)
        g1 = full.backward(dcf, cotangent_dot=dot1)
        for j, i in enumerate(mine):
            v = t * len(mine) + j
            assert torch.equal(color[v], c1[i]) and torch.equal(radii[v], r1[i])
            assert dot[v].item() == dot1[i].item()
            for k in ("means3D", "means2D", "opacities", "scales", "rotations", "colors_precomp"):
                assert torch.equal(g[k][v], g1[k][i]), (t, i, k)


@pytest.mark.parametrize("scale_modifier", [0.6, 1.7])
def test_scale_modifier_other_than_one(scale_modifier, render_build):
    """GaussianRasterizationSettings.scale_modifier (helpers.py:79 passes 1.0; the field scales the three axes before cov3D
    and therefore dL/dscales): forward and all gradients against the C oracle and the float64 autograd oracle."""
    from scaffold import scene
    H = W = 96
    V = 3
    rv, cams = util.make_scene(20, 32, H, W, V, opacity="B", seed=31)
    cams = [c._replace(scale_modifier=scale_modifier) for c in cams]
    dc, dd, da = scene.output_cotangents(V, H, W, seed=32, depth_alpha=True)
    hip, hg, batch = util.hip_render(cams, rv, dc, dd, da)
    assert abs(batch.prob.scale_modifier - scale_modifier) < 1e-6
    ref1, _, _ = util.hip_render([c._replace(scale_modifier=1.0) for c in cams], rv)
    assert (hip["radii"] != ref1["radii"]).any(), "the modifier must change the footprints"
    for v in range(V):
        r, g = util.c_oracle_render(cams[v], rv, dc[v], dd[v], da[v])
        np.testing.assert_array_equal(hip["radii"][v], r.radii)
        check_outputs(hip, r.color, r.depth, r.alpha, v)
        check_grads(hg, g, v)
        outs, grads = util.torch_oracle_render(cams[v], rv, dc[v], dd[v], da[v])
        check_outputs(hip, outs["color"].numpy(), outs["depth"].numpy(), outs["alpha"].numpy(), v)
        check_grads(hg, {k: t.numpy() for k, t in grads.items()}, v)


def test_scale_modifier_through_the_drop_in_with_sh_colours():
    """... and through `GaussianRasterizer` itself (the settings tuple carries it), SH colours, one view."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from scaffold import scene
    H = W = 80
    rv, cams = util.make_scene(16, 24, H, W, 2, opacity="B", sh_degree=2, seed=33)
    cam = cams[1]._replace(scale_modifier=1.7)
    dev = torch.device("cuda")
    leaf = {k: v.to(dev).requires_grad_(True) for k, v in rv.items()}
    dcam = util.to_device([cam], dev)[0]
    im, radii, depth, alpha = GaussianRasterizer(raster_settings=dcam)(
        means3D=leaf["means3D"], means2D=torch.zeros_like(leaf["means3D"], requires_grad=True), opacities=leaf["opacities"],
        shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])
    dc, _, _ = scene.output_cotangents(1, H, W, seed=34)
    (im * dc[0].to(dev)).sum().backward()
    r, g = util.c_oracle_render(cam, rv, dc[0])
    assert np.abs(im.detach().cpu().numpy() - r.color).max() <= 2e-5
    np.testing.assert_array_equal(radii.cpu().numpy(), r.radii)
    for k in ("means3D", "scales", "rotations", "opacities", "shs"):
        a = leaf[k].grad.cpu().numpy().astype(np.float64)
        b = np.asarray(g[k], np.float64).reshape(a.shape)
        assert np.abs(a - b).max() <= 2e-4 * max(np.abs(b).max(), 1e-30) + 1e-9, k
