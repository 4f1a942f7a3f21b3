"""
TEST INFRASTRUCTURE.  Builds the one piece of the REAL reference that sits next to the hot path and compiles from
its own single source file: the CPU UV-space triangle rasterizer `_render_colors_core`
(/root/reference/face3d/mesh/cython/mesh_core.cpp:169-234), used by Topo4D's texture bake
(helpers.py:953-960 -> face3d/mesh/render.py:52-86).

The source is compiled WHERE IT LIES (never copied into this repo), with the flags the reference's own
setup.py/distutils would use for the arithmetic (-O2, no fast-math), into oracle/_ref/libmesh_core_ref.so
(git-ignored; travels to the GPU box with the snapshot).  The C++ symbol is called through its mangled name, so no
shim source is needed.  Also (re)writes tests/golden/g5_render_colors.npz from that library.

Run:  python oracle/build_ref.py          (only works where /root/reference exists)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = "/root/reference/face3d/mesh/cython/mesh_core.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libmesh_core_ref.so")


def build(force=False):
    if not os.path.exists(SRC):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(SRC) > os.path.getmtime(OUT):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-fwrapv", "-fno-strict-aliasing", SRC, "-o", OUT])
    return OUT


def write_golden():
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import texture_oracle as TX
    rng = np.random.default_rng(5)
    # small UV mesh: jittered 9x9 grid over [0,1]^2 -> 128 triangles, rendered at 48x40 so that the 2-pixel border
    # ring (mesh_core.cpp:211) and interior texels are both populated; plus a few overlapping triangles with depth.
    n = 9
    u, v = np.meshgrid(np.linspace(0.02, 0.98, n), np.linspace(0.03, 0.97, n), indexing="xy")
    uv = np.stack([u.ravel(), v.ravel()], 1) + rng.normal(0, 0.01, size=(n * n, 2))
    tris = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i
            tris += [[a, a + 1, a + n], [a + 1, a + n + 1, a + n]]
    tris = np.asarray(tris, np.int32)
    colors = rng.uniform(0, 1, size=(n * n, 3)).astype(np.float32)
    h, w = 40, 48
    verts = np.stack([uv[:, 0] * (w - 1), h - uv[:, 1] * (h - 1) - 1, np.zeros(n * n)], 1).astype(np.float32)   # process_uv
    img = TX.render_colors_ref(verts, tris, colors, h, w, 3)
    verts_z = verts.copy()
    verts_z[:, 2] = rng.normal(0, 1, size=n * n).astype(np.float32)
    extra = np.asarray([[0, 40, 80], [8, 36, 72], [4, 44, 76]], np.int32)
    tris2 = np.concatenate([tris, extra])
    img_z = TX.render_colors_ref(verts_z, tris2, colors, h, w, 3)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g5_render_colors.npz"), verts=verts, tris=tris, colors=colors,
                        hw=np.array([h, w]), image=img, verts_z=verts_z, tris_z=tris2, image_z=img_z)


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print("reference mesh_core:", p)
    if p:
        write_golden()
        print("wrote tests/golden/g5_render_colors.npz")
