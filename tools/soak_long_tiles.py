"""Soak run of the long-tile path of big one-view launches (one view of more than 8,192 tiles: tiles of at least 2,048 pairs are cut
into depth segments, in a launch of their own behind the whole-tile backward - HISTORY.md section 5 item 1): random heads with
random clusters of thin splats, the hybrid backward against the whole-tile replay of everything (T4D_NO_SEGMENTS=1; the forward is
the same program either way and must be bit-equal), and the long tiles' depth-parallel forward against the one-pass forward
(T4D_NO_LONG_FWD=1).  GPU box.
    python tools/soak_long_tiles.py [first_seed] [n_seeds]      -> one line per failing seed + a summary line."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scaffold import scene
from tests import util
from topo4d_amd import rasterizer

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad, cut, longest, moved = [], 0, 0, 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    H = int(rng.choice([1472, 1475, 1536, 2044, 2048])); W = int(rng.choice([1472, 1480, 1600, 2041, 2048]))      # (ragged sizes too: partial tiles on the right and bottom edges)
    g = torch.Generator().manual_seed(seed)
    rv, cams = util.make_scene(int(rng.integers(40, 90)), int(rng.integers(60, 140)), H, W, 1, opacity="B", seed=seed)
    for _ in range(int(rng.integers(1, 4))):                                   # one to three clusters somewhere on the head
        n = int(rng.integers(1500, 7000))
        centre = rv["means3D"][int(rng.integers(0, rv["means3D"].shape[0]))]
        spread = float(rng.uniform(0.0005, 0.004))
        extra = {
            "means3D": centre[None] + torch.randn(n, 3, generator=g) * torch.tensor([spread, spread, 2 * spread]),
            "colors_precomp": torch.rand(n, 3, generator=g),
            "rotations": torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1),
            "opacities": torch.rand(n, 1, generator=g) * float(rng.uniform(0.01, 0.3)) + 0.004,
            "scales": torch.rand(n, 3, generator=g) * float(rng.uniform(0.0003, 0.002)) + 0.0003,
        }
        rv = {k: torch.cat([v, extra[k]]).contiguous() if k in extra else v for k, v in rv.items()}
    cams = [c._replace(bg=torch.tensor(rng.uniform(0, 1, 3), dtype=torch.float32)) for c in cams]
    use_da = bool(seed % 3)
    dc, dd, da = scene.output_cotangents(1, H, W, seed=seed + 7, depth_alpha=True)
    if not use_da:
        dd = da = None
    try:
        # (the host keeps speed hints per scene SIZE - longest tile list seen, learned arena capacity: two seeds of the same (P, H, W)
        # - 3130 and 3836 are - would hand the first render of the later one the earlier one's "no long lists" hint, i.e. another,
        # equally correct, forward program than its second render runs: round 6 spent an hour on that "race")
        rasterizer._forget_scenes()
        os.environ["T4D_NO_SEGMENTS"] = "1"
        o_w, g_w, b_w = util.hip_render(cams, rv, dc, dd, da)
        os.environ.pop("T4D_NO_SEGMENTS")
        o_h, g_h, b_h = util.hip_render(cams, rv, dc, dd, da)
        st = util.decode_state(b_h)
        cut += int((st["tile_count"] >= 2048).sum()); longest = max(longest, int(st["tile_count"].max()))
        for k in o_w:
            if not np.array_equal(o_w[k], o_h[k]):
                # the same forward program twice: anything that differs is a race or a read of unwritten state - say where
                sw = util.decode_state(b_w)
                dpx = (o_w[k] != o_h[k]).reshape(-1, H, W).any(0) if o_w[k].ndim >= 3 else None
                where = ""
                if dpx is not None:
                    ys, xs = np.nonzero(dpx)
                    gx_ = (W + 15) // 16
                    tl = sorted(set(((ys // 16) * gx_ + xs // 16).tolist()))
                    where = f": {len(ys)} pixels in tiles {[(t, int(st['tile_count'][0][t])) for t in tl[:6]]}, max {np.abs(o_w[k].astype(np.float64) - o_h[k]).max():.3e}"
                tcw = sw["tile_count"][0]
                kd = [(int(t), int(tcw[t])) for t in np.nonzero(tcw > 0)[0]
                      if not np.array_equal(sw["keys"][0, sw["tile_off"][0, t]: sw["tile_off"][0, t] + tcw[t]], st["keys"][0, st["tile_off"][0, t]: st["tile_off"][0, t] + tcw[t]])]
                unsorted = [(int(t), int(tcw[t])) for t in np.nonzero(tcw > 1)[0]
                            if not np.all(np.diff(st["keys"][0, st["tile_off"][0, t]: st["tile_off"][0, t] + tcw[t]].astype(np.uint64)) > 0)]
                unsorted_w = [(int(t), int(tcw[t])) for t in np.nonzero(tcw > 1)[0]
                              if not np.all(np.diff(sw["keys"][0, sw["tile_off"][0, t]: sw["tile_off"][0, t] + tcw[t]].astype(np.uint64)) > 0)]
                raise AssertionError(f"forward output {k} differs{where}; keys differ in tiles {kd[:6]}; unsorted bins (default run) {unsorted[:6]}, (NO_SEGMENTS run) {unsorted_w[:6]}; "
                                     f"n_contrib differs on {int((sw['n_contrib'] != st['n_contrib']).sum())} px")
        for k in g_w:
            if g_w[k] is None:
                continue
            scale = np.abs(g_w[k]).max()
            err = np.abs(g_w[k].astype(np.float64) - g_h[k]).max()
            assert err <= 2e-5 * scale + 1e-12, f"grad {k}: {err:.3e} vs scale {scale:.3e}"
        # the long tiles' depth-parallel FORWARD against the one-pass forward (T4D_NO_LONG_FWD): the same decisions except for pixels
        # within rounding of a threshold (counted), sums associated differently
        os.environ["T4D_NO_LONG_FWD"] = "1"
        o_1, g_1, b_1 = util.hip_render(cams, rv, dc, dd, da)
        os.environ.pop("T4D_NO_LONG_FWD")
        s_1 = util.decode_state(b_1)
        mv = s_1["n_contrib"] != st["n_contrib"]
        moved += int(mv.sum())
        assert mv.sum() <= 6, f"{int(mv.sum())} pixels with another last contributor"
        for k in ("color", "depth", "alpha"):
            err = np.abs(o_1[k].astype(np.float64) - o_h[k])
            perpix = err.reshape(err.shape[0], -1, err.shape[-2], err.shape[-1]).max(axis=(0, 1)) if err.ndim == 4 else err.reshape(-1, err.shape[-2], err.shape[-1]).max(axis=0)
            assert perpix[~mv[0]].max() <= 5e-6, f"forward {k}: {perpix[~mv[0]].max():.3e}"
            assert err.max() <= 2e-3, f"forward {k} at a moved pixel: {err.max():.3e}"
        assert np.abs(s_1["final_T"] - st["final_T"])[~mv].max() <= 1e-6, "final_T"
        for k in g_1:
            if g_1[k] is None:
                continue
            scale = np.abs(g_1[k]).max()
            err = np.abs(g_1[k].astype(np.float64) - g_h[k]).max()
            assert err <= (2e-5 if mv.sum() == 0 else 2e-2) * scale + 1e-12, f"grad {k} (one-pass forward): {err:.3e} vs scale {scale:.3e}"
    except Exception as e:
        os.environ.pop("T4D_NO_SEGMENTS", None); os.environ.pop("T4D_NO_LONG_FWD", None)
        bad.append(seed)
        print(f"seed {seed} FAILED ({H}x{W}): {str(e).splitlines()[0][:900]}", flush=True)
print(f"soak long tiles: {n_seeds} scenes (seeds {first}..{first + n_seeds - 1}), {cut} tiles cut into segments, longest list {longest}, "
      f"{moved} pixels whose last contributor differs between the depth-parallel and the one-pass forward, "
      f"{len(bad)} failures {bad[:20]}")
