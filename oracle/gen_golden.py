"""
TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by IMPORTING THE REAL REFERENCE (/root/reference) in this
container.  The reference cannot travel to the GPU box, so its outputs are committed as small data fixtures
together with this script.  Nothing here is product code and no reference source text is stored in the fixtures —
only seeded inputs and the numbers the reference functions returned.

    python oracle/gen_golden.py            # rewrites tests/golden/g1..g4,g6 (g5 is written by oracle/build_ref.py)

G1  helpers.setup_camera          (helpers.py:63-88)   3 synthetic (K, w2c, w, h) -> the 12 Settings fields
G2  helpers.params2rendervar      (helpers.py:91-100)  seeded P=64 parameter dict -> rasterizer kwargs
G3  helpers.l1_loss_v1 + external.calc_ssim (helpers.py:115-116, external.py:73-116) on seeded 3x64x64 pairs,
    with the gradient of 0.8*L1 + 0.2*(1-SSIM) w.r.t. the rendered image (train.py:315)
G4  helpers.eval_sh               (helpers.py:865-922) degrees 0..3 on seeded [P=32,3,16] coefficients
G7  helpers.compute_vertex_attribute_by_weight_2 (helpers.py:237-253) on a seeded quad mesh
G8  train.get_loss_dense(..., use_mask=True) (train.py:380-417: helpers.get_mask + the masked L1 of :394-405), called for
    real with `Renderer` bound to a stub that returns a seeded render: loss['im'] and its gradient w.r.t. the render
G9  train.get_loss(..., use_mask=True) (train.py:300-377) for is_initial_timestep True and False: the branch train.py:631 makes the
    live one - helpers.get_mask(["inner_mouth"]) (helpers.py:811-823), masked_gt = 0.1 x gt there, camera affine (:310)
G10 train.get_loss_dense(..., use_mask=False) (train.py:380-417 as :632,:735 call it) incl. 0.02 x soft_color (helpers.py:119-120)
G6  SELF-GENERATED (not reference-derived): forward outputs + all gradients of oracle/torch_oracle.py (float64)
    on a 64x64 / P=200 scene; pins the oracle against accidental edits.
The rasterizer itself has no reference-derived golden vectors: its source is absent from /root/reference
 — parity unpinned.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def import_reference_helpers():
    from typing import NamedTuple

    class Camera(NamedTuple):
        image_height: int
        image_width: int
        tanfovx: float
        tanfovy: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        projmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool

    stubs = ["imageio", "open3d", "pywavefront", "nvdiffrast", "nvdiffrast.torch", "torchvision", "torchvision.utils",
             "skimage", "skimage.io", "trimesh", "face3d", "diff_gaussian_rasterization", "cv2", "pymesh"]
    for name in stubs:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    sys.modules["skimage"].io = sys.modules["skimage.io"]
    sys.modules["face3d"].mesh = types.ModuleType("face3d.mesh")
    sys.modules["nvdiffrast"].torch = sys.modules["nvdiffrast.torch"]
    sys.modules["diff_gaussian_rasterization"].GaussianRasterizer = object
    sys.modules["diff_gaussian_rasterization"].GaussianRasterizationSettings = Camera
    # no GPU in this container: .cuda() / device="cuda" become no-ops
    torch.Tensor.cuda = lambda self, *a, **k: self
    _tensor, _zeros_like = torch.tensor, torch.zeros_like

    def tensor(*a, **k):
        k.pop("device", None)
        return _tensor(*a, **k)

    def zeros_like(*a, **k):
        k.pop("device", None)
        return _zeros_like(*a, **k)
    torch.tensor, torch.zeros_like = tensor, zeros_like
    sys.path.insert(0, REF)
    import external  # noqa
    import helpers  # noqa
    return helpers, external


def import_reference_train():
    """train.py itself (after import_reference_helpers): modules it needs beyond the stubs above are stubbed on demand."""
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return lambda *a, **kw: None
    for _ in range(40):
        try:
            import train  # noqa
            return train
        except ModuleNotFoundError as e:
            sys.modules[e.name] = _Any(e.name)
    raise RuntimeError("could not import the reference train.py")


def gen_g8(helpers):
    train = import_reference_train()
    H, W, P = 48, 40, 8
    g = torch.Generator().manual_seed(8)
    cmap = np.asarray(helpers.cmap)                                        # helpers.py:806 (uint8 [14,3], BGR order)
    target_colors = [torch.tile(torch.tensor(cmap[i]).reshape(3, 1, 1), (1, H, W)) for i in range(14)]     # as helpers.py:807
    labels = torch.randint(0, 14, (H // 4, W // 4), generator=g).repeat_interleave(4, 0).repeat_interleave(4, 1)
    mask = torch.stack([torch.tensor(cmap[:, c].astype(np.float32))[labels] for c in range(3)]) / 255.0   # the [3,H,W] label image
    im = torch.rand(3, H, W, generator=g).requires_grad_(True)
    gt = torch.rand(3, H, W, generator=g)
    radius = torch.zeros(P, dtype=torch.int32)

    class StubRenderer:                                                    # stands in for the un-vendored rasterizer only
        def __init__(self, raster_settings=None):
            pass

        def __call__(self, **kw):
            return im, radius, None, None
    train.Renderer = StubRenderer
    params = {"dense_means3D": torch.rand(P, 3, generator=g), "dense_rgb_colors": torch.rand(P, 3, generator=g),
              "dense_unnorm_rotations": torch.rand(P, 4, generator=g), "dense_logit_opacities": torch.zeros(P, 1),
              "dense_log_scales": torch.zeros(P, 3)}
    variables = {"target_colors_dense": target_colors, "dense_init_colors": torch.rand(P, 3, generator=g),
                 "dense_max_2D_radius": torch.zeros(P)}
    curr = {"cam": None, "im": gt, "mask": mask, "id": 0}
    loss, variables, detail = train.get_loss_dense(params, curr, variables, 0, 0, None, use_mask=True, losses_list={},
                                                   losses_weights={"im": 1.0, "soft_color": 0.0})
    loss.backward()
    target_labels = ["skin", "l_eyebrow", "r_eyebrow", "nose", "upper_lip", "lower_lip", "l_ear", "r_ear", "hair"]   # train.py:396-398
    filtered = helpers.get_mask(target_labels, mask, train.cmap_index, target_colors)
    np.savez_compressed(os.path.join(OUT, "g8_masked_l1.npz"), im=im.detach().numpy(), gt=gt.numpy(), mask_image=mask.numpy(),
                        filtered_mask=filtered.numpy(), loss_im=np.float32(detail["im"].item()), grad_im=im.grad.numpy(),
                        masked_elements=np.int64((filtered == 1).sum().item()))


class _Recorder:
    """Wraps a reference function: calls the REAL one and keeps the arguments it was called with."""
    def __init__(self, fn):
        self.fn, self.calls = fn, []

    def __call__(self, *a, **k):
        self.calls.append(a)
        return self.fn(*a, **k)


def _label_image(cmap, labels, H, W, g, perturb=True):
    """A face-parsing label image as get_dataset loads it (train.py:84-92): uint8 colours / 255.0 in float64, then .float().
    With `perturb`, a band of pixels sits ONE or TWO grey levels off a label colour in one channel: the `< 1` test of
    helpers.get_mask (helpers.py:819) then runs within float32 round-off of its threshold."""
    img = np.stack([cmap[:, c][labels] for c in range(3)], axis=-1).astype(np.int64)          # [H,W,3]
    if perturb:
        off = torch.randint(-2, 3, (H, W), generator=g).numpy()
        ch = torch.randint(0, 3, (H, W), generator=g).numpy()
        band = np.zeros((H, W), bool)
        band[::3] = True
        for c in range(3):
            sel = band & (ch == c)
            img[..., c][sel] = np.clip(img[..., c][sel] + off[sel], 0, 255)
    mask64 = img.astype(np.uint8) / 255.0                                                  # train.py:90
    return torch.tensor(mask64).float().permute(2, 0, 1)                                   # train.py:92


def gen_g9(helpers):
    """G9: the REAL train.get_loss (train.py:300-377) with use_mask=True - what train.py:631,647 hard-code - for the first frame
    (is_initial_timestep: the plain photometric branch :318) and for every later frame (:320-327: helpers.get_mask on
    ["inner_mouth"], masked_gt = gt with the masked elements x 0.1).  `Renderer` is bound to a stub that returns a seeded
    render (the rasterizer is the un-vendored part); the topology regularisers (out of scope) are bound to
    stubs that return zero.  Captured: loss['im'], the gradient of the total loss w.r.t. the render, cam_m and cam_c,
    get_mask's filtered_mask, and the masked_gt the reference handed to l1_loss_v1."""
    train = import_reference_train()
    H, W, P, K, n_cams, cid = 48, 40, 12, 3, 4, 2
    g = torch.Generator().manual_seed(9)
    cmap = np.asarray(helpers.cmap)
    target_colors = [torch.tile(torch.tensor(cmap[i]).reshape(3, 1, 1), (1, H, W)) for i in range(14)]     # as train.py:635
    labels = torch.randint(0, 14, (H // 4, W // 4), generator=g)
    labels[2:5, 3:7] = 8                                                                   # a solid inner-mouth patch
    labels = labels.repeat_interleave(4, 0).repeat_interleave(4, 1).numpy()
    mask = _label_image(cmap, labels, H, W, g)
    gt = torch.rand(3, H, W, generator=g)
    radius = torch.randint(0, 5, (P,), generator=g, dtype=torch.int32)
    zero_pair = lambda *a, **k: (torch.tensor(0.0), None)
    zero_one = lambda *a, **k: torch.tensor(0.0)
    losses_list = {k: zero_one for k in ("flat", "flat_lip_bottom", "flat_lip_socket", "flat_eye", "flat_face_bottom")}
    losses_list.update({k: zero_pair for k in ("flat_lid_top", "flat_lid_bottom", "flat_lip", "flat_mouth")})
    weights = {'im': 1.0, 'rigid': 3.5, 'rot': 20.0, 'iso': 20.0, 'flat': 2e-4, 'flat_lip_bottom': 2e-4, 'flat_lid_top': 2e-4,
               'flat_lid_bottom': 1e-2, 'flat_lip': 1e-4, 'flat_mouth': 1e-3, 'flat_eye': 1e4, 'flat_face_bottom': 1e3,
               'flat_lip_socket': 1e3, 'scale': 10.0, 'scale_max': 10.0}                     # train.py:535-540
    _zeros = torch.zeros
    torch.zeros = lambda *a, **k: _zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"})      # external.py:29 says device='cuda'
    out = {"gt": gt.numpy(), "mask_image": mask.numpy(), "cam_id": np.int64(cid), "radius": radius.numpy()}
    try:
        for tag, initial in (("first", True), ("later", False)):
            im = torch.rand(3, H, W, generator=g).requires_grad_(True)

            class StubRenderer:
                def __init__(self, raster_settings=None):
                    pass

                def __call__(self, **kw):
                    return im, radius, None, None
            train.Renderer = StubRenderer
            params = {"means3D": torch.rand(P, 3, generator=g), "rgb_colors": torch.rand(P, 3, generator=g),
                      "unnorm_rotations": torch.rand(P, 4, generator=g) + 0.1, "logit_opacities": torch.zeros(P, 1),
                      "log_scales": torch.zeros(P, 3) - 4, "cam_m": (torch.rand(n_cams, 3, generator=g) - 0.5) * 0.2,
                      "cam_c": (torch.rand(n_cams, 3, generator=g) - 0.5) * 0.1}
            params = {k: v.requires_grad_(True) for k, v in params.items()}
            nb = torch.randint(0, P, (P, K), generator=g)
            variables = {"target_colors_low": target_colors, "max_2D_radius": torch.zeros(P), "init_scale": torch.ones(P),
                         "neighbor_indices": nb, "prev_inv_rot_fg": torch.rand(P, 4, generator=g), "prev_offset": torch.rand(P, K, 3, generator=g),
                         "rig_w": torch.rand(P, K, generator=g), "rot_w": torch.rand(P, K, generator=g),
                         "neighbor_dist": torch.rand(P, K, generator=g), "iso_w": torch.rand(P, K, generator=g),
                         "cos_init_lid_top": None, "cos_init_lid_bottom": None, "cos_init_lip": None, "cos_init_mouth": None}
            rec_l1, rec_ssim = _Recorder(helpers.l1_loss_v1), _Recorder(train.calc_ssim)
            train.l1_loss_v1, train.calc_ssim = rec_l1, rec_ssim
            curr = {"cam": None, "im": gt, "mask": mask, "id": cid}
            try:
                loss, variables, detail = train.get_loss(params, curr, variables, initial, use_mask=True, losses_list=losses_list,
                                                         losses_weights=weights)
            finally:
                train.l1_loss_v1, train.calc_ssim = rec_l1.fn, rec_ssim.fn
            loss.backward()
            target_seen = rec_l1.calls[0][1]                                # what the reference compared the render with
            out[f"{tag}_im"] = im.detach().numpy()
            out[f"{tag}_cam_m"] = params["cam_m"].detach().numpy()
            out[f"{tag}_cam_c"] = params["cam_c"].detach().numpy()
            out[f"{tag}_loss_im"] = np.float32(detail["im"].item())
            out[f"{tag}_l1"] = np.float32(helpers.l1_loss_v1(rec_l1.calls[0][0], target_seen).item())
            out[f"{tag}_ssim"] = np.float32(rec_ssim.fn(rec_ssim.calls[0][0], rec_ssim.calls[0][1]).item())
            out[f"{tag}_grad_im"] = im.grad.numpy()
            out[f"{tag}_grad_cam_m"] = params["cam_m"].grad.numpy()
            out[f"{tag}_grad_cam_c"] = params["cam_c"].grad.numpy()
            out[f"{tag}_target"] = target_seen.detach().numpy()
            out[f"{tag}_seen"] = variables["seen"].numpy()
            out[f"{tag}_max_2D_radius"] = variables["max_2D_radius"].numpy()
    finally:
        torch.zeros = _zeros
    filtered = helpers.get_mask(["inner_mouth"], mask, train.cmap_index, target_colors)        # train.py:322-323
    out["filtered_mask"] = filtered.numpy()
    # a label image that went through an interpolating rotation / resize (camera.rotate_image, train.py:91): values OFF the
    # 8-bit grid, many within float32 round-off of the `< 1` threshold - pins the arithmetic (mask * 255, then - colour, in float32)
    noise = (torch.rand(3, H, W, generator=g) - 0.5) * (2.4 / 255.0)
    edge = torch.where(torch.rand(3, H, W, generator=g) < 0.5, 1.0, -1.0) * (1.0 + (torch.randint(-3, 4, (3, H, W), generator=g).float() * 2e-7))
    soft = mask + torch.where(torch.rand(3, H, W, generator=g) < 0.3, edge / 255.0, noise)
    out["mask_image_soft"] = soft.numpy()
    out["filtered_mask_soft"] = helpers.get_mask(["inner_mouth"], soft, train.cmap_index, target_colors).numpy()
    two = helpers.get_mask(["upper_lip", "inner_mouth", "lower_lip"], soft, train.cmap_index, target_colors)
    out["filtered_mask_soft_3_labels"] = two.numpy()
    out["label_colors"] = cmap.astype(np.uint8)                                                 # helpers.py:806 (BGR order)
    out["inner_mouth_index"] = np.int64(train.cmap_index["inner_mouth"])
    assert np.array_equal(out["first_target"], gt.numpy())                                      # first frame: the plain target
    assert not np.array_equal(out["later_target"], gt.numpy())
    np.savez_compressed(os.path.join(OUT, "g9_get_loss_masked.npz"), **out)


def gen_g10(helpers):
    """G10: the REAL train.get_loss_dense (train.py:380-417) as train.py:632,735-737 call it - use_mask False (plain 0.8 L1 + 0.2
    (1-SSIM) on the render, NO camera affine) plus losses_weights_dense['soft_color'] = 0.02 times helpers.l1_loss_v2
    (helpers.py:119-120) between dense_rgb_colors and dense_init_colors.  Some colours equal their initial value exactly (the
    rows train.py:732-734 pin to zero start that way): torch's |x| has gradient 0 there."""
    train = import_reference_train()
    H, W, P = 40, 56, 96
    g = torch.Generator().manual_seed(10)
    im = torch.rand(3, H, W, generator=g).requires_grad_(True)
    gt = torch.rand(3, H, W, generator=g)
    radius = torch.randint(0, 4, (P,), generator=g, dtype=torch.int32)

    class StubRenderer:
        def __init__(self, raster_settings=None):
            pass

        def __call__(self, **kw):
            return im, radius, None, None
    train.Renderer = StubRenderer
    init = torch.rand(P, 3, generator=g)
    rgb = init + (torch.rand(P, 3, generator=g) - 0.5) * 0.2
    rgb[::5] = init[::5]                                                                     # exact ties: sign(0) = 0
    rgb[1::7] = 0.0
    params = {"dense_means3D": torch.rand(P, 3, generator=g), "dense_rgb_colors": rgb.clone().requires_grad_(True),
              "dense_unnorm_rotations": torch.rand(P, 4, generator=g), "dense_logit_opacities": torch.zeros(P, 1),
              "dense_log_scales": torch.zeros(P, 3)}
    variables = {"dense_init_colors": init, "dense_max_2D_radius": torch.zeros(P)}
    curr = {"cam": None, "im": gt, "mask": None, "id": 0}
    weights = {"im": 1.0, "soft_color": 0.02}                                                # train.py:541-543
    loss, variables, detail = train.get_loss_dense(params, curr, variables, 1, 0, None, use_mask=False, losses_list={},
                                                   losses_weights=weights)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "g10_get_loss_dense.npz"), im=im.detach().numpy(), gt=gt.numpy(), radius=radius.numpy(),
                        dense_rgb_colors=rgb.numpy(), dense_init_colors=init.numpy(), loss=np.float32(loss.item()),
                        loss_im=np.float32(detail["im"].item()), loss_soft_color=np.float32(detail["soft_color"].item()),
                        grad_im=im.grad.numpy(), grad_dense_rgb_colors=params["dense_rgb_colors"].grad.numpy(),
                        dense_seen=variables["dense_seen"].numpy(), dense_max_2D_radius=variables["dense_max_2D_radius"].numpy())


def main():
    os.makedirs(OUT, exist_ok=True)
    helpers, external = import_reference_helpers()
    rng = np.random.default_rng(0)

    # ---- G1 -------------------------------------------------------------------------------------------------
    g1 = {}
    cases = [(375, 512, 820.0, 815.0, 190.0, 250.0), (512, 512, 1344.0, 1344.0, 256.0, 256.0), (750, 1024, 1650.5, 1640.25, 370.0, 515.5)]
    for i, (w, h, fx, fy, cx, cy) in enumerate(cases):
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
        a = rng.normal(size=3)
        a /= np.linalg.norm(a)
        th = rng.uniform(0.2, 1.2)
        Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        w2c = np.eye(4)
        w2c[:3, :3] = R
        w2c[:3, 3] = rng.normal(size=3) * 0.3 + np.array([0, 0, 0.9])
        cam = helpers.setup_camera(None, w, h, K, w2c, near=0.01, far=100)
        g1[f"K{i}"] = K
        g1[f"w2c{i}"] = w2c
        g1[f"wh{i}"] = np.array([w, h])
        g1[f"scalars{i}"] = np.array([cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, cam.scale_modifier,
                                      cam.sh_degree, int(cam.prefiltered), int(cam.debug)], dtype=np.float64)
        g1[f"bg{i}"] = cam.bg.numpy()
        g1[f"viewmatrix{i}"] = cam.viewmatrix.numpy()
        g1[f"projmatrix{i}"] = cam.projmatrix.numpy()
        g1[f"campos{i}"] = cam.campos.numpy()
    np.savez(os.path.join(OUT, "g1_setup_camera.npz"), **g1)

    # ---- G2 -------------------------------------------------------------------------------------------------
    P = 64
    params = {"means3D": rng.normal(size=(P, 3)) * 0.1, "rgb_colors": rng.uniform(size=(P, 3)),
              "unnorm_rotations": rng.normal(size=(P, 4)), "logit_opacities": rng.normal(size=(P, 1)) * 3,
              "log_scales": rng.normal(size=(P, 3)) - 5}
    params_t = {k: torch.tensor(v).float() for k, v in params.items()}
    rv = helpers.params2rendervar(params_t)
    g2 = {f"in_{k}": v.numpy() for k, v in params_t.items()}
    g2.update({f"out_{k}": v.detach().numpy() for k, v in rv.items()})
    np.savez(os.path.join(OUT, "g2_params2rendervar.npz"), **g2)

    # ---- G3 -------------------------------------------------------------------------------------------------
    g3 = {}
    for i in range(3):
        im = torch.tensor(rng.uniform(size=(3, 64, 64))).float().requires_grad_(True)
        gt = torch.tensor(np.clip(im.detach().numpy() + rng.normal(size=(3, 64, 64)) * 0.1, 0, 1)).float()
        l1 = helpers.l1_loss_v1(im, gt)
        ssim = external.calc_ssim(im, gt)
        loss = 0.8 * l1 + 0.2 * (1.0 - ssim)
        (grad,) = torch.autograd.grad(loss, im)
        g3[f"im{i}"] = im.detach().numpy()
        g3[f"gt{i}"] = gt.numpy()
        g3[f"l1_{i}"] = np.float64(l1.item())
        g3[f"ssim_{i}"] = np.float64(ssim.item())
        g3[f"loss_{i}"] = np.float64(loss.item())
        g3[f"grad{i}"] = grad.numpy()
    np.savez(os.path.join(OUT, "g3_photometric.npz"), **g3)

    # ---- G4 -------------------------------------------------------------------------------------------------
    sh = torch.tensor(rng.normal(size=(32, 3, 16))).float()
    d = rng.normal(size=(32, 3))
    d = torch.tensor(d / np.linalg.norm(d, axis=1, keepdims=True)).float()
    g4 = {"sh": sh.numpy(), "dirs": d.numpy()}
    for deg in range(4):
        g4[f"eval_deg{deg}"] = helpers.eval_sh(deg, sh, d).numpy()
    g4["C0"] = np.float64(helpers.C0)
    g4["C1"] = np.float64(helpers.C1)
    g4["C2"] = np.array(helpers.C2)
    g4["C3"] = np.array(helpers.C3)
    g4["RGB2SH_half"] = helpers.RGB2SH(torch.tensor([0.0, 0.5, 1.0])).numpy()
    np.savez(os.path.join(OUT, "g4_eval_sh.npz"), **g4)

    # ---- G7: helpers.compute_vertex_attribute_by_weight_2 (helpers.py:237-253) ---------------------------------
    n_c, n_q, n_d = 50, 30, 400
    quads = rng.integers(0, n_c, size=(n_q, 4)).astype(np.int64)
    father = rng.integers(0, n_q, size=(n_d, 1)).astype(np.int64)
    wgt = rng.uniform(size=(n_d, 4)); wgt /= wgt.sum(1, keepdims=True)
    attr = rng.normal(size=(n_c, 3)).astype(np.float32)
    variables = {"dense_vertex_father": father, "dense_vertex_weight": wgt, "dense_quad_faces": quads,
                 "dense_vertex": np.zeros((n_c + n_d, 3))}
    dense = helpers.compute_vertex_attribute_by_weight_2(variables, attr)
    np.savez(os.path.join(OUT, "g7_dense_interp.npz"), quads=quads, father=father, weight=wgt, attr=attr, dense=dense,
             dense_f32=torch.tensor(dense).float().numpy())

    # ---- G8 -------------------------------------------------------------------------------------------------
    gen_g8(helpers)
    gen_g9(helpers)
    gen_g10(helpers)

    # ---- G6 (self-generated) --------------------------------------------------------------------------------
    from oracle import torch_oracle as TO
    from scaffold import reference_boundary as boundary, scene
    p = scene.make_gaussians(10, 20, opacity="B", seed=6)
    rvv = {k: v.detach() for k, v in boundary.params2rendervar(p).items()}
    cam = scene.camera_rig(64, 64, n_views=3)[1]
    dc, dd, da = scene.output_cotangents(1, 64, 64, seed=7, depth_alpha=True)
    outs, grads = TO.rasterize_with_grads(TO.View(*cam), rvv["means3D"], rvv["opacities"], rvv["scales"], rvv["rotations"],
                                          colors_precomp=rvv["colors_precomp"], dL_dcolor=dc[0], dL_ddepth=dd[0],
                                          dL_dalpha=da[0])
    g6 = {"color": outs["color"].numpy(), "depth": outs["depth"].numpy(), "alpha": outs["alpha"].numpy(),
          "radii": outs["radii"].numpy(), "n_contrib": outs["n_contrib"].numpy()}
    g6.update({f"grad_{k}": v.numpy() for k, v in grads.items()})
    np.savez_compressed(os.path.join(OUT, "g6_self_oracle_f64.npz"), **g6)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
