"""CPU: host-side logic of the drop-in module — API surface, argument validation (mirrors upstream's messages),
view-record packing, capacity policy, and the "no CPU fallback" contract."""
import os

import numpy as np
import pytest
import torch

import topo4d_amd
from scaffold import reference_boundary as boundary, scene
from topo4d_amd import _lib, rasterizer


def _cam():
    return scene.camera_rig(64, 64, n_views=3)[1]


def test_drop_in_module_surface():
    import diff_gaussian_rasterization as dgr
    assert dgr.GaussianRasterizer is topo4d_amd.GaussianRasterizer
    assert dgr.GaussianRasterizationSettings is topo4d_amd.GaussianRasterizationSettings
    s = _cam()
    assert len(s) == 12
    r = dgr.GaussianRasterizer(raster_settings=s)            # keyword used at train.py:307
    assert isinstance(r, torch.nn.Module) and r.raster_settings is s
    assert hasattr(r, "markVisible")
    import inspect
    assert list(inspect.signature(r.forward).parameters) == ["means3D", "means2D", "opacities", "shs", "colors_precomp",
                                                              "scales", "rotations", "cov3D_precomp"]


def test_argument_validation_messages_match_upstream():
    r = topo4d_amd.GaussianRasterizer(_cam())
    P = 4
    m = torch.zeros(P, 3); o = torch.ones(P, 1); s = torch.ones(P, 3); q = torch.ones(P, 4); c = torch.ones(P, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, scales=s, rotations=q)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, shs=torch.ones(P, 1, 3), colors_precomp=c, scales=s, rotations=q)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, colors_precomp=c, scales=s)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, colors_precomp=c, scales=s, rotations=q, cov3D_precomp=torch.ones(P, 6))


def test_no_cpu_fallback():
    r = topo4d_amd.GaussianRasterizer(_cam())
    P = 4
    m = torch.zeros(P, 3); o = torch.ones(P, 1); s = torch.ones(P, 3); q = torch.ones(P, 4); c = torch.ones(P, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(m, m, o, colors_precomp=c, scales=s, rotations=q)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(m)
    with pytest.raises(RuntimeError, match="no CPU path"):
        topo4d_amd.ViewBatch(torch.zeros(1, 40), 64, 64)


def test_missing_extension_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtopo4d_raster.so")
    with pytest.raises(_lib.ExtensionMissing, match="not built"):
        _lib.load()


def test_pack_views_layout_and_cache():
    cams = scene.camera_rig(64, 48, n_views=3)
    rec = rasterizer.pack_views(cams, torch.device("cpu"))
    assert rec.shape == (3, _lib.T4D_VIEW_FLOATS) and rec.dtype == torch.float32
    for i, c in enumerate(cams):
        np.testing.assert_array_equal(rec[i, :16].numpy(), c.viewmatrix.reshape(-1).numpy())
        np.testing.assert_array_equal(rec[i, 16:32].numpy(), c.projmatrix.reshape(-1).numpy())
        np.testing.assert_array_equal(rec[i, 32:35].numpy(), c.campos.numpy())
        np.testing.assert_array_equal(rec[i, 35:38].numpy(), c.bg.numpy())
        assert rec[i, 38].item() == np.float32(c.tanfovx) and rec[i, 39].item() == np.float32(c.tanfovy)
    # element (row r, col c) of the mathematical world->view matrix sits at [c*4+r] (helpers.py:67 transposes)
    w2c = cams[0].viewmatrix[0].T
    assert rec[0, 1 * 4 + 2].item() == w2c[2, 1].item()
    rec2 = rasterizer.pack_views(cams, torch.device("cpu"))
    assert rec2[0].data_ptr() != 0 and torch.equal(rec, rec2)
    cams[0].viewmatrix.mul_(1.0)                       # in-place edit bumps the version -> cache entry is rebuilt
    assert torch.equal(rasterizer.pack_views(cams, torch.device("cpu")), rec)
    with pytest.raises(ValueError):
        rasterizer._check_common([cams[0], cams[1]._replace(image_height=32)])


def test_capacity_policy():
    assert rasterizer._initial_capacity(10) == 16384 and rasterizer._initial_capacity(30000) == 240000
    c = rasterizer._round_capacity(74000)
    assert c >= 1.5 * 74000 and c % 1024 == 0
    with pytest.raises(ValueError):
        topo4d_amd.set_sync_mode("sometimes")
    topo4d_amd.set_sync_mode("lazy"); assert topo4d_amd.get_sync_mode() == "lazy"
    topo4d_amd.set_sync_mode("checked")


def test_scene_generator_is_seeded_and_matches_reference_recipe():
    a = scene.make_gaussians(20, 30, opacity="A", seed=0)
    b = scene.make_gaussians(20, 30, opacity="A", seed=0)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert a["means3D"].shape == (600, 3)
    assert torch.all(torch.sigmoid(a["logit_opacities"]) == 1.0)           # train.py:142: logit 1000
    assert torch.allclose(a["log_scales"][:, 0], a["log_scales"][:, 1])    # isotropic (train.py:143)
    cams = scene.camera_rig(512, 512, 24)
    assert len(cams) == 24 and cams[0].viewmatrix.shape == (1, 4, 4)
    rv = boundary.params2rendervar(a)
    assert torch.allclose(rv["rotations"].norm(dim=1), torch.ones(600), atol=1e-6)
    assert scene.CONFIGS["C2"] == dict(n_lat=150, n_lon=200, H=512, W=512, n_views=24, sh_degree=None)


def test_graphed_views_argument_checks_need_no_gpu():
    from topo4d_amd import loop
    from topo4d_amd.optim import FusedAdamPins
    p = {"means3D": torch.nn.Parameter(torch.zeros(4, 3))}
    groups = [{"params": [p["means3D"]], "name": "means3D", "lr": 1e-3}]
    with pytest.raises(ValueError, match="capturable"):
        loop.GraphedViews(p, [], FusedAdamPins(groups))
    with pytest.raises(RuntimeError, match="GPU only"):
        loop.GraphedViews(p, [], FusedAdamPins(groups, capturable=True))
    opt = FusedAdamPins(groups, capturable=True)
    assert opt.steps() == [0]


def test_rasterizer_keeps_the_nn_module_contract_although_it_builds_its_state_lazily():
    """.eval() on a fresh instance used to be undone by the lazy nn.Module set-up, and assigning a submodule raised."""
    cam = scene.camera_rig(32, 32, n_views=1)[0]
    r = topo4d_amd.GaussianRasterizer(cam)
    assert r.eval().training is False and r.train().training is True
    r = topo4d_amd.GaussianRasterizer(cam)
    r.train(False)
    assert r.training is False
    r = topo4d_amd.GaussianRasterizer(raster_settings=cam)
    r.head = torch.nn.Linear(2, 2)
    assert "head" in dict(r.named_modules()) and len(list(r.parameters())) == 2
    assert r.raster_settings is cam
    seen = []
    h = torch.nn.modules.module.register_module_forward_pre_hook(lambda m, a: seen.append(type(m).__name__))
    try:
        with pytest.raises(Exception):                        # CPU tensors: the call itself fails, but the global hook ran first
            r(torch.zeros(1, 3), None, torch.zeros(1, 1), colors_precomp=torch.zeros(1, 3), scales=torch.zeros(1, 3), rotations=torch.zeros(1, 4))
    finally:
        h.remove()
    assert seen == ["GaussianRasterizer"]


def test_sync_mode_save_restore_leaves_no_trace():
    saved = rasterizer._save_sync_mode()
    try:
        rasterizer._restore_sync_mode(("checked", False))
        assert topo4d_amd.get_sync_mode() == "checked" and topo4d_amd.get_sync_mode(drop_in=True) == "auto"
        inner = rasterizer._save_sync_mode()
        topo4d_amd.set_sync_mode("lazy")
        assert topo4d_amd.get_sync_mode(drop_in=True) == "lazy"
        rasterizer._restore_sync_mode(inner)
        assert topo4d_amd.get_sync_mode(drop_in=True) == "auto"      # the drop-in's default is the default again
    finally:
        rasterizer._restore_sync_mode(saved)


def test_hand_chained_iteration_has_no_cpu_path_and_checks_its_arguments():
    """loop.explicit_iteration / explicit_frame_iteration / optimise_views(explicit=True) and the raw entry points they use fail
    loudly on CPU tensors (no fallback), and refuse what they cannot chain by hand."""
    from scaffold import scene
    from topo4d_amd import loop, loss, boundary
    from topo4d_amd.optim import FusedAdamPins
    p = {k: torch.nn.Parameter(v) for k, v in scene.make_gaussians(4, 6, opacity="A", seed=0).items()}
    cam = _cam()
    data = {"cam": cam, "im": torch.zeros(3, int(cam.image_height), int(cam.image_width)), "id": 0}
    with pytest.raises(RuntimeError, match="no CPU path|GPU only"):
        loop.explicit_iteration(p, data)
    with pytest.raises(RuntimeError, match="no CPU path|GPU only"):
        loop.explicit_frame_iteration(p, [data])
    with pytest.raises(RuntimeError, match="no CPU path"):
        loss.photometric_loss_raw(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="GPU only"):
        boundary.activate_forward(p["unnorm_rotations"].detach(), p["logit_opacities"].detach(), p["log_scales"].detach())
    groups = [{"params": [v], "name": k, "lr": 1e-3} for k, v in p.items()]
    with pytest.raises(ValueError, match="explicit=True needs"):           # CPU parameters cannot be chained by hand
        loop.optimise_views(p, [data], FusedAdamPins(groups), n_iters=1, explicit=True)
    with pytest.raises(ValueError, match="explicit=True needs"):           # nor can an extra loss term
        loop.optimise_views(p, [data], FusedAdamPins(groups), n_iters=1, extra_loss=lambda a, b: 0.0, explicit=True)
    assert _lib.T4D_FLAG_RAW_PARAMS == 128 and _lib.T4D_ADAM_CLEAR_GRAD == 1
    # the branches train.py really runs: masked targets, the texture iteration - same rules
    with pytest.raises(RuntimeError, match="no CPU path"):
        loss.label_mask_target(torch.zeros(3, 8, 8), [[0, 0, 64]], torch.zeros(3, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU path"):
        loss.soft_color_loss_raw(torch.zeros(4, 3), torch.zeros(4, 3), 0.02)
    with pytest.raises(ValueError, match="prepare_masked_targets|label_colors"):
        loop.target_image(data, use_mask=True, is_initial_timestep=False)
    assert loop.target_image(data, use_mask=True, is_initial_timestep=True) is data["im"]
    assert loop.target_image(data, use_mask=False, is_initial_timestep=False) is data["im"]
    with pytest.raises(ValueError, match="'mask'"):
        loop.prepare_masked_targets([dict(data)], [[0, 0, 64]])
    dense, init = scene.make_dense_params({k: v.detach() for k, v in p.items()}, per_vertex=2)
    dp = {k: torch.nn.Parameter(v) for k, v in dense.items()}
    dgroups = [{"params": [v], "name": k, "lr": 1e-3} for k, v in dp.items()]
    with pytest.raises(ValueError, match="explicit=True needs"):
        loop.optimise_dense_views(dp, {"dense_init_colors": init}, [data], FusedAdamPins(dgroups), n_iters=1, explicit=True)
    with pytest.raises(ValueError, match="dense_init_colors"):
        loop.GraphedViews(dp, [data], FusedAdamPins(dgroups, capturable=True), dense=True)
    opt = FusedAdamPins(dgroups)
    opt.apply_pins()                                                         # no pins: no launch, no library call
    assert not hasattr(loss, "photometric_loss_torch") and not hasattr(loss, "ssim_torch")    # the checkers live in oracle/


def test_parsing_colormap_is_the_reference_cmap():
    """scaffold.scene.parsing_colormap_bgr == helpers.py:806 `label_colormap(14)[:, [2, 1, 0]]` (golden G9 holds the real one)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_get_loss_masked.npz"))
    assert np.array_equal(scene.parsing_colormap_bgr(14), g["label_colors"])
    assert scene.PARSING_LABELS.index("inner_mouth") == int(g["inner_mouth_index"])
    m = scene.make_label_image(48, 40, seed=2)
    assert m.shape == (3, 48, 40) and m.dtype == torch.float32
    inner = (torch.abs(m * 255 - torch.tensor(g["label_colors"][8]).reshape(3, 1, 1)) < 1).all(0)
    assert 0 < int(inner.sum()) < 48 * 40 // 4


def test_drop_in_import_runs_the_backward_on_the_calling_thread():
    """Importing `diff_gaussian_rasterization` switches torch's per-device autograd worker thread off (the hand-over per
    `loss.backward()` made the one-view loop run at 4-5 k OR 8-9 k it/s: profiles/r06_dropin_regimes.txt);
    T4D_AUTOGRAD_ENGINE_THREAD=1 leaves torch's default alone.  Fresh interpreters: the setting is process-wide."""
    import subprocess
    import sys
    code = "import torch, diff_gaussian_rasterization; print(torch.autograd.is_multithreading_enabled())"
    env = {k: v for k, v in os.environ.items() if k != "T4D_AUTOGRAD_ENGINE_THREAD"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "False", out.stdout + out.stderr
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, T4D_AUTOGRAD_ENGINE_THREAD="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "True", out.stdout + out.stderr
