#!/usr/bin/env python
"""CPU model (no GPU needed): wave-instruction counts of k_render_fwd AS BUILT and of candidate decompositions of its visit lists,
counted on the C oracle's data for BASELINE config 2 and config 4 (count the FORWARD's decompositions the way the
backward's were counted - tools/experiments/model_bwd_decompositions.py - and build only what the calibrated model predicts at
<= 0.8 x today's wave-instructions at C2 AND C4).

    python tools/experiments/model_fwd_decompositions.py [--config C2|C4] [--views 0,9,18] [--tile-stride 1]  > profiles/r06_fwd_models_<cfg>.json

The forward differs from the backward in two ways that matter here: it has no cross-lane reduction (finer lists do not pay the
22-DPP tax that closed the backward), and it does not know where a pixel stops until it gets there - a wave walks until ALL of its
64 pixels have finished (T < 1e-4) or the staged batch ends, a tile stages batches until all four waves have finished.

For a few views the oracle supplies every tile's depth-ordered list and the splats' centres / conics / opacities; numpy re-evaluates
alpha for every (pixel, listed splat), replays the transmittance, and from that the script counts what each decomposition would
issue.  "as built" is the calibration: its wave-steps, row visits, blending lanes and live wave-batches must reproduce the counting
build's (profiles/lanes.json, `fwd`), and its price - vector instructions per wave-step read off the ISA of the built kernel
(.LBB*_125 of k_render_fwd<false,192,0,false>: 89 per group of four steps), the rest per live wave-batch - must reproduce
SQ_INSTS_VALU of the rocprofv3 pass (profiles/valu.json).  Candidates are priced OPTIMISTICALLY (no spills, free addresses).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from model_bwd_decompositions import block_touch, cutoff_r2          # noqa: E402  (the kernels' conservative circle test)

BATCH = 192                     # kFwdBatch: staged splats per batch
GROUP = 4                       # kU: steps per group of the throughput build
STEP_BUILT = 89 / 4.0           # vector instructions per wave-step (ISA)
T_STOP = 1e-4


def tile_eval(st, t, gx, H, W):
    lo, hi = int(st["ranges"][t, 0]), int(st["ranges"][t, 1])
    n = hi - lo
    if n <= 0:
        return None
    ids = st["point_list"][lo:hi].astype(np.int64)
    ty, tx = divmod(t, gx)
    tid = np.arange(256)
    w, r, i = tid >> 6, (tid >> 4) & 3, tid & 15
    px = tx * 16 + ((w & 1) << 3) + ((r & 1) << 2) + (i & 3)
    py = ty * 16 + ((w >> 1) << 3) + ((r >> 1) << 2) + (i >> 2)
    inside = (px < W) & (py < H)
    xy, co = st["xy"][ids], st["conic_opacity"][ids]
    dx = xy[None, :, 0] - px[:, None].astype(np.float32)
    dy = xy[None, :, 1] - py[:, None].astype(np.float32)
    power = -0.5 * (co[None, :, 0] * dx * dx + co[None, :, 2] * dy * dy) - co[None, :, 1] * dx * dy
    alpha = np.minimum(0.99, co[None, :, 3] * np.exp(power))
    valid = (power <= 0) & (alpha >= 1.0 / 255.0)
    # transmittance replay: the stopping splat is the first valid one that would take T below T_STOP (it is not blended)
    Tafter = np.cumprod(np.where(valid, 1.0 - alpha.astype(np.float64), 1.0), axis=1)
    stop = valid & (Tafter < T_STOP)
    done_pos = np.where(stop.any(1), stop.argmax(1), n)            # list position at which the pixel finishes (n: never)
    done_pos = np.where(inside, done_pos, -1)                      # pixels outside the image are finished from the start
    blends = valid & (np.arange(n)[None, :] < done_pos[:, None])
    return xy, co, valid, blends, done_pos, (tx, ty), n


def walked(list_idx_done, lens, group=GROUP):
    """Steps a wave walks in one batch: its lists' longest length, in groups - or, if all its pixels finish inside the batch, up to the
    group in which the last one does.  list_idx_done: per pixel, index IN ITS LIST of the entry that finishes it (None if it does
    not finish in this batch)."""
    nsteps = int(max(lens)) if len(lens) else 0
    full = -(-nsteps // group) * group
    if list_idx_done is None:
        return full
    return min(full, (int(list_idx_done) // group + 1) * group)


def count_view(st, H, W, stride):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    keys = ("pairs", "tiles", "blend_lanes", "batches", "steps", "visits",
            "prune_steps", "exact_steps", "exact_visits", "exact_prune_steps",
            "quad_steps", "quad_visits", "quad_exact_steps", "quad_exact_visits",
            "h42_steps", "h42_visits", "h42_exact_steps", "lane_steps", "ideal_steps")
    c = {k: 0 for k in keys}
    for t in range(0, gx * gy, stride):
        te = tile_eval(st, t, gx, H, W)
        if te is None:
            continue
        xy, co, valid, blends, done_pos, (tx, ty), n = te
        r2 = cutoff_r2(co)
        c["pairs"] += n
        c["tiles"] += 1
        c["blend_lanes"] += int(blends.sum())
        # circle touch per 4x4 sub-block (wave, row), per 2x2 quad and per 4x2 half-row block
        def touches(bw, bh):
            out = []
            for w in range(4):
                for by in range(8 // bh):
                    for bx in range(8 // bw):
                        x0 = tx * 16 + ((w & 1) << 3) + bw * bx
                        y0 = ty * 16 + ((w >> 1) << 3) + bh * by
                        out.append(block_touch(xy, r2, x0, y0, bw, bh))
            return np.stack(out)                                   # [4 waves * blocks per wave, n]
        # pixel index (tile_pixel order) -> block index for a (bw, bh) partition of the wave's 8x8
        tid = np.arange(256)
        w_, r_, i_ = tid >> 6, (tid >> 4) & 3, tid & 15
        lx = ((r_ & 1) << 2) + (i_ & 3)
        ly = ((r_ >> 1) << 2) + (i_ >> 2)
        def block_of(bw, bh):
            return w_ * ((8 // bw) * (8 // bh)) + (ly // bh) * (8 // bw) + (lx // bw)
        parts = {"44": (4, 4), "22": (2, 2), "42": (4, 2)}
        touch = {k: touches(*p) for k, p in parts.items()}
        blk = {k: block_of(*p) for k, p in parts.items()}
        exact = {}
        for k in parts:
            nb = touch[k].shape[0]
            ex = np.zeros((nb, n), bool)
            for b_ in range(nb):
                ex[b_] = valid[blk[k] == b_].any(0)
            exact[k] = ex
        wave_done = np.array([done_pos[64 * w:64 * w + 64].max() for w in range(4)])       # position after which the wave is finished
        nbatch = (n + BATCH - 1) // BATCH
        for b in range(nbatch):
            lo = b * BATCH
            hi = min(n, lo + BATCH)
            if b and (wave_done < lo).all():                       # every wave finished in an earlier batch: the tile stops staging
                break
            for w in range(4):
                if wave_done[w] < lo:                              # this wave is finished: it only takes part in the barriers
                    continue
                c["batches"] += 1
                pix = np.arange(64 * w, 64 * w + 64)
                dpos = done_pos[pix]
                fin = bool((dpos < hi).all())                      # the wave finishes inside this batch
                live_px = dpos >= lo                               # pixels still blending at the start of the batch

                def steps_for(member, blocks_of_px, prune):
                    """member [blocks of this wave, hi - lo] bool: list membership.  Returns (walked steps, visits)."""
                    m = member.copy()
                    if prune:                                      # a block all of whose pixels have finished takes no more splats
                        for bi in range(m.shape[0]):
                            if not live_px[blocks_of_px == bi].any():
                                m[bi] = False
                    lens = m.sum(1)
                    idx_done = None
                    if fin:
                        # the entry that finishes pixel p sits in ITS block's list at index (#members before its position)
                        worst = -1
                        for p in range(64):
                            if dpos[p] < lo:
                                continue
                            bi = blocks_of_px[p]
                            worst = max(worst, int(m[bi, :dpos[p] - lo].sum()))
                        idx_done = max(worst, 0)
                    return walked(idx_done, lens), int(lens.sum())

                for name, part, ex_, prune in (("steps", "44", False, False), ("prune_steps", "44", False, True),
                                               ("exact_steps", "44", True, False), ("exact_prune_steps", "44", True, True),
                                               ("quad_steps", "22", False, True), ("quad_exact_steps", "22", True, True),
                                               ("h42_steps", "42", False, True), ("h42_exact_steps", "42", True, True)):
                    nblk = touch[part].shape[0] // 4
                    src = exact[part] if ex_ else touch[part]
                    member = src[w * nblk:(w + 1) * nblk, lo:hi]
                    s_, v_ = steps_for(member, blk[part][pix] - w * nblk, prune)
                    c[name] += s_
                    vis = {"steps": "visits", "exact_steps": "exact_visits", "quad_steps": "quad_visits", "quad_exact_steps": "quad_exact_visits",
                           "h42_steps": "h42_visits"}.get(name)
                    if vis:
                        c[vis] += v_
                # per-lane exact lists: the busiest pixel's blends in this batch (the floor of any list scheme for this wave)
                c["lane_steps"] += -(-int(blends[pix, lo:hi].sum(1).max()) // GROUP) * GROUP
                # perfect packing: blending lanes / 64
                c["ideal_steps"] += blends[pix, lo:hi].sum() / 64.0
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--views", default=None)
    ap.add_argument("--opacity", default="A")
    ap.add_argument("--tile-stride", type=int, default=None)
    a = ap.parse_args()
    from oracle import c_oracle
    from scaffold import reference_boundary as boundary, scene
    cfg = scene.CONFIGS[a.config]
    H, W = cfg["H"], cfg["W"]
    views = [int(v) for v in (a.views or ("0,9,18" if a.config == "C2" else "9")).split(",")]
    stride = a.tile_stride or (1 if a.config == "C2" else 5)
    params = scene.make_gaussians(cfg["n_lat"], cfg["n_lon"], opacity=a.opacity, sh_degree=cfg["sh_degree"], seed=0)
    rv = {k: v.detach() for k, v in boundary.params2rendervar(params).items()}
    shs = None
    if cfg["sh_degree"] is not None:
        shs = params["shs"]
        rv.pop("colors_precomp")
    cams = scene.camera_rig(H, W, n_views=24, true_campos=cfg["sh_degree"] is not None)
    if cfg["sh_degree"] is not None:
        cams = [c._replace(sh_degree=cfg["sh_degree"]) for c in cams]
    tot = None
    for v in views:
        r = c_oracle.OracleRender(cams[v], rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv.get("colors_precomp"), shs)
        c = count_view(r.state(), H, W, stride)
        tot = c if tot is None else {k: tot[k] + c[k] for k in c}
    s = 24.0 / len(views) * stride                          # scale to a 24-view launch
    T = {k: v * s for k, v in tot.items()}
    key = a.config + ("" if a.opacity == "A" else "_" + a.opacity)
    lanes = json.load(open(os.path.join(ROOT, "profiles", "lanes.json"))).get(key, {}).get("fwd")
    valu = json.load(open(os.path.join(ROOT, "profiles", "valu.json"))).get(key, {}).get("k_render_fwd")
    out = {"scene": f"{a.config} scenario {a.opacity}, views {views}, every {stride}-th tile, scaled to 24 views",
           "counts_per_24_views": {k: round(v) for k, v in T.items()}}
    measured = valu["SQ_INSTS_VALU"] if valu else None
    # price of everything outside the walk, per live wave-batch: what is left of the MEASURED instruction count (staging of 192
    # records, three chunks of touch masks, four lists, padding, the barriers' bookkeeping, prologue and write-out of the tile)
    batch_built = (measured - lanes["wave_steps"] * STEP_BUILT) / lanes["wave_batches"] if (lanes and measured) else 284.0
    if lanes:
        out["calibration"] = {"wave_steps model / counting build": round(T["steps"] / lanes["wave_steps"], 3),
                              "row_visits": round(T["visits"] / lanes["row_visits"], 3),
                              "blending_lanes": round(T["blend_lanes"] / lanes["contributing_lane_steps"], 3),
                              "wave_batches": round(T["batches"] / lanes["wave_batches"], 3),
                              "SQ_INSTS_VALU measured": measured, "per_wave_batch_price_from_it": round(batch_built, 1)}
    built = T["steps"] * STEP_BUILT + T["batches"] * batch_built
    M = {}

    def model(name, steps, price_step, price_batch, note):
        wi = steps * price_step + T["batches"] * price_batch
        M[name] = {"wave_instructions": round(wi), "steps": round(steps), "vs_as_built": round(wi / built, 3),
                   "price": f"{price_step:.2f} per step + {price_batch:.0f} per live wave-batch", "note": note}

    model("as_built", T["steps"], STEP_BUILT, batch_built, "4x4 sub-block lists, conservative circle test, waves stop when their 64 pixels have finished")
    model("as_built_pruning_finished_rows", T["prune_steps"], STEP_BUILT, batch_built + 12,
          "the PRUNE instantiation (dense passes run it): a sub-block whose 16 pixels have finished walks an empty list; +12 per batch for the test")
    # exact elliptical test of a 4x4 block: minimum of the quadratic form over the block instead of the circle - the closest point of
    # the box in the conic's metric: ~ +16 instructions per sub-block and chunk of 64 staged splats (4 rows x 3 chunks per batch) -
    # priced at +190 per batch; the COUNT is the ideal (a visit only if some pixel of the block takes the splat)
    model("exact_culling_4x4", T["exact_steps"], STEP_BUILT, batch_built + 190, "4x4 lists, a visit only if one of the 16 pixels takes the splat (ideal exact test)")
    model("exact_culling_4x4_pruning", T["exact_prune_steps"], STEP_BUILT, batch_built + 202, "... and finished sub-blocks walk empty lists")
    # 2x2 quads: sixteen lists per wave.  Step price unchanged (no reduction in the forward; LDS reads are per lane anyway).  Per batch:
    # sixteen ballots per chunk instead of four (+12 x 3 x ~6), sixteen lists built instead of four (+12 x 3 x 8 = 288), sixteen padded (+12 x 2)
    model("quads_2x2_circle", T["quad_steps"], STEP_BUILT, batch_built + 530, "sixteen 2x2 lists per wave, circle test, finished quads pruned")
    model("quads_2x2_exact", T["quad_exact_steps"], STEP_BUILT, batch_built + 530 + 760, "... with the ideal exact test (+16 per quad and chunk)")
    model("blocks_4x2_circle", T["h42_steps"], STEP_BUILT, batch_built + 180, "eight 4x2 lists per wave (8 lanes each), circle test, finished blocks pruned")
    model("blocks_4x2_exact", T["h42_exact_steps"], STEP_BUILT, batch_built + 180 + 380, "... with the ideal exact test")
    model("floor_per_lane_lists", T["lane_steps"], STEP_BUILT, batch_built, "FLOOR of any list scheme: every wave steps as often as its busiest pixel blends (lists priced at zero)")
    model("floor_perfect_packing", T["ideal_steps"], STEP_BUILT, batch_built, "blending lanes / 64: nothing can issue fewer steps")
    out["models"] = M
    out["gate"] = "VERDICT r05 item 4: build only what is predicted <= 0.8 x as built at C2 AND C4"
    out["verdict"] = {k: ("candidate" if m["vs_as_built"] <= 0.8 and not k.startswith("floor") else "no") for k, m in M.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
