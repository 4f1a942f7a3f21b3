"""GPU: t4d_texture_bake through the C ABI == the reference's CPU rasterizer, bit for bit."""
import numpy as np
import pytest
import torch

from oracle import texture_oracle as TX
from scaffold.scene import uv_mesh
from tests.test_texture_oracle import G

pytestmark = pytest.mark.gpu


def test_golden_g5_bit_exact():
    from topo4d_amd import texture
    g = np.load(G)
    h, w = (int(x) for x in g["hw"])
    img = texture.render_colors(g["verts"], g["tris"], g["colors"], h, w).cpu().numpy()
    np.testing.assert_array_equal(img, g["image"])
    img = texture.render_colors(g["verts_z"], g["tris_z"], g["colors"], h, w).cpu().numpy()
    np.testing.assert_array_equal(img, g["image_z"])


@pytest.mark.parametrize("with_depth", [False, True])
def test_random_uv_mesh_bit_exact_vs_cpu_reference(with_depth):
    from topo4d_amd import texture
    for n, h, w, seed in ((60, 250, 333, 1), (257, 1024, 1024, 2)):     # 1024^2 / 131,072 triangles: BASELINE.md's case
        verts, tris, colors = uv_mesh(n, h, w, seed, with_depth)
        ref, dref = TX.render_colors_cpu(verts, tris, colors, h, w, return_depth=True)
        img, dep = texture.render_colors(verts, tris, colors, h, w, return_depth=True)
        np.testing.assert_array_equal(img.cpu().numpy(), ref)
        np.testing.assert_array_equal(dep.cpu().numpy(), dref)


def test_background_bands_and_overflow_retry():
    from topo4d_amd import texture
    h, w = 200, 160
    verts, tris, colors = uv_mesh(50, h, w, 4, True)
    bg = np.random.default_rng(5).uniform(size=(h, w, 3)).astype(np.float32)
    ref = TX.render_colors_cpu(verts, tris[:2000], colors, h, w, BG=bg)
    np.testing.assert_array_equal(texture.render_colors(verts, tris[:2000], colors, h, w, BG=bg).cpu().numpy(), ref)
    # row bands (the multi-GPU shard unit) assemble to the full image
    full = texture.render_colors(verts, tris, colors, h, w).cpu().numpy()
    parts = np.zeros_like(full)
    for r0, r1 in ((0, 37), (37, 128), (128, 200)):
        band = texture.render_colors(verts, tris, colors, h, w, rows=(r0, r1)).cpu().numpy()
        assert not band[:r0].any() and not band[r1:].any()
        parts[r0:r1] = band[r0:r1]
    np.testing.assert_array_equal(parts, full)
    np.testing.assert_array_equal(full, TX.render_colors_cpu(verts, tris, colors, h, w))
    # a giant triangle touches every tile: first attempt overflows the pair arena and is retried
    texture._CAP.clear()
    big = np.concatenate([tris, [[0, 49, 2499]]]).astype(np.int32)
    texture._CAP[(0, big.shape[0], h, w)] = 64
    a = texture.render_colors(verts, big, colors, h, w).cpu().numpy()
    np.testing.assert_array_equal(a, TX.render_colors_cpu(verts, big, colors, h, w))


def test_bake_texture_bytes_match_reference_pipeline():
    from topo4d_amd import texture
    rng = np.random.default_rng(7)
    n, res = 40, 256
    u, v = np.meshgrid(np.linspace(0, 1, n), np.linspace(0, 1, n), indexing="xy")
    uvs = np.stack([u.ravel(), v.ravel()], 1)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c_, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a, b, c_], 1), np.stack([b, d, c_], 1)]).astype(np.int32)
    colors = rng.uniform(0, 1, size=(n * n, 3))
    mine = texture.bake_texture(uvs, colors, faces, res)
    uvc = texture.process_uv(uvs, res, res)                       # helpers.py:955
    ref = TX.render_colors_cpu(uvc, faces, colors, res, res)      # helpers.py:956
    np.testing.assert_array_equal(mine, (ref * 255).astype(np.uint8))
    assert mine.dtype == np.uint8 and mine.shape == (res, res, 3)


def test_dense_attribute_interpolation_bit_exact_vs_reference_golden():
    """G7 = outputs of the real helpers.compute_vertex_attribute_by_weight_2 (float64) and their `.float()` cast."""
    import os
    from topo4d_amd import texture
    g = np.load(os.path.join(os.path.dirname(G), "g7_dense_interp.npz"))
    variables = {"dense_vertex_father": g["father"], "dense_vertex_weight": g["weight"], "dense_quad_faces": g["quads"],
                 "dense_vertex": np.zeros((g["dense"].shape[0], 3))}
    out = texture.compute_vertex_attribute_by_weight(variables, torch.tensor(g["attr"]).cuda())
    np.testing.assert_array_equal(out.cpu().numpy(), g["dense_f32"])
    # plain numpy restatement on fresh data, bigger
    rng = np.random.default_rng(1)
    n_c, n_q, n_d = 8280, 5000, 200000
    quads = rng.integers(0, n_c, size=(n_q, 4)); father = rng.integers(0, n_q, size=(n_d, 1))
    w = rng.uniform(size=(n_d, 4)); attr = rng.normal(size=(n_c, 4)).astype(np.float32)
    ref = np.zeros((n_c + n_d, 4)); ref[:n_c] = attr
    ref[n_c:] = np.sum(attr[quads[father].squeeze(1)] * w[..., None], axis=1)
    v2 = {"dense_vertex_father": father, "dense_vertex_weight": w, "dense_quad_faces": quads, "dense_vertex": ref}
    out = texture.compute_vertex_attribute_by_weight(v2, torch.tensor(attr).cuda())
    np.testing.assert_array_equal(out.cpu().numpy(), ref.astype(np.float32))


def _sharded_bake_worker(rank, world, port, out_path, backend="gloo"):
    import os
    import torch.distributed as dist
    from topo4d_amd import texture
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":                                                 # RCCL: one rank per GPU, so world == 1 on the test box
        os.environ["T4D_FORCE_COLLECTIVES"] = "1"                         # (a one-rank group skips the collective otherwise)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share cuda:0 in this test
    try:
        rng = np.random.default_rng(7)
        n, res = 40, 250
        uvs = rng.uniform(0.02, 0.98, size=(n * n, 2)).astype(np.float32)
        colors = rng.uniform(size=(n * n, 3)).astype(np.float32)
        idx = np.arange(n * n).reshape(n, n)
        faces = np.concatenate([np.stack([idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:]], -1).reshape(-1, 3),
                                np.stack([idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], -1).reshape(-1, 3)]).astype(np.int32)
        img = texture.bake_texture_sharded(uvs, colors, faces, res)
        if rank == 0:
            np.save(out_path, img)
            np.save(out_path + ".single.npy", texture.bake_texture(uvs, colors, faces, res))
    finally:
        dist.destroy_process_group()


def test_sharded_bake_two_ranks_equals_single_bake(tmp_path):
    """BASELINE config 5's shard: every rank bakes a row band, the bands are all-gathered (gloo here, RCCL on a real node)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "sharded.npy")
    mp.spawn(_sharded_bake_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = np.load(out), np.load(out + ".single.npy")
    assert a.shape == (250, 250, 3) and a.dtype == np.uint8 and a.any()
    np.testing.assert_array_equal(a, b)


def test_sharded_bake_eight_ranks_equals_single_bake(tmp_path):
    """BASELINE config 5's 1 -> 8 shard at its real width: eight ranks (sharing the test GPU, gloo for RCCL), 250 rows in bands of
    32 / 31, byte-identical to the single bake."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "sharded8.npy")
    mp.spawn(_sharded_bake_worker, args=(8, port, out), nprocs=8, join=True)
    a, b = np.load(out), np.load(out + ".single.npy")
    assert a.shape == (250, 250, 3) and a.any()
    np.testing.assert_array_equal(a, b)


def test_sharded_bake_through_rccl_with_one_rank(tmp_path):
    """The RCCL ("nccl") branch of the band gather executed for real: a process group of ONE rank on the test box's one GPU -
    init with device_id, all_gather_into_tensor of the band on the device, destroy - before an 8-GPU node ever sees it."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "sharded_rccl.npy")
    mp.spawn(_sharded_bake_worker, args=(1, port, out, "nccl"), nprocs=1, join=True)
    a, b = np.load(out), np.load(out + ".single.npy")
    assert a.shape == (250, 250, 3) and a.any()
    np.testing.assert_array_equal(a, b)


def test_in_out_entry_two_passes_equal_one_pass():
    """`t4d_texture_bake` keeps `_render_colors_core`'s in/out buffers (mesh_core.h:63-69): baking the first half of the triangles
    and then the second half onto the SAME image and depth buffer is the reference's serial walk cut in two - identical to one pass."""
    import ctypes as C
    from topo4d_amd import _lib, texture
    lib = _lib.load()
    h, w = 300, 260
    verts, tris, colors = uv_mesh(70, h, w, 9, True)
    ref, dref = TX.render_colors_cpu(verts, tris, colors, h, w, return_depth=True)
    v, t, c = (torch.as_tensor(np.ascontiguousarray(x)).cuda() for x in (verts.astype(np.float32), tris.astype(np.int32),
                                                                           colors.astype(np.float32)))
    image = torch.zeros(h, w, 3, device="cuda")
    depth = torch.full((h, w), -999999.0, device="cuda")
    half = tris.shape[0] // 2
    p = lambda x: C.c_void_p(x.data_ptr())
    for part in (t[:half].contiguous(), t[half:].contiguous()):
        cap = 8 * int(part.shape[0]) + 65536
        nb = lib.t4d_texture_bake_scratch_bytes(h, w, cap)
        sc = torch.empty(nb, dtype=torch.uint8, device="cuda")
        need = C.c_int64(0)
        rc = lib.t4d_texture_bake(p(v), p(part), p(c), int(v.shape[0]), int(part.shape[0]), h, w, 3, 0, h, p(image), p(depth), p(sc), nb,
                                  cap, C.byref(need), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, _lib.last_error()
    np.testing.assert_array_equal(image.cpu().numpy(), ref)
    np.testing.assert_array_equal(depth.cpu().numpy(), dref)
