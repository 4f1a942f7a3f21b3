#!/usr/bin/env python
"""CPU model (no GPU needed): wave-instruction counts of k_render_bwd AS BUILT and of candidate decompositions, counted on the
C oracle's data for BASELINE config 2 ("model first, then build, else stop for good").

    python tools/experiments/model_bwd_decompositions.py [--views 0,9,18] [--opacity A]  > profiles/r05_bwd_models.json

For a few views of the config-2 scene the oracle supplies every tile's depth-ordered list, the splats' centres / conics / opacities
and every pixel's last contributor; numpy re-evaluates alpha for every (pixel, listed splat) and from that the script counts, per
decomposition, the wave-steps it would issue, and prices a step with the instruction count given next to each model (read off the
ISA of the built kernel for "as built", written out instruction by instruction in the docstrings for the candidates).  The
"as built" row is the calibration: its counts must reproduce the counting build's (profiles/lanes.json: wave-steps, row-visits,
contributing lanes) - printed as `calibration` - otherwise no other row means anything.

Instruction prices are deliberately OPTIMISTIC for the candidates (no spills, perfect packing, every address free): a candidate
that does not beat the built kernel here cannot beat it on the chip.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

BATCH = 128                     # kBwdBatch: staged splats per batch
STEP_BUILT = 55                 # vector instructions per wave-step of the built kernel (HISTORY.md section 6.3)
BATCH_BUILT = 340               # per wave-batch outside the walk (staging, masks, lists, conflict ballots, write-out)


def cutoff_r2(co):
    """cutoff_radius2 of t4d_raster_render_fwd.h (the conservative circle of the alpha >= 1/255 ellipse)."""
    A, B, Cc, op = co[:, 0], co[:, 1], co[:, 2], co[:, 3]
    lnarg = np.log(np.maximum(255.0 * op, 1e-30))
    mid = 0.5 * (A + Cc)
    det = A * Cc - B * B
    lmin = det / (mid + np.sqrt(np.maximum(mid * mid - det, 0.0)))
    r2 = 2.0 * (lnarg + 2e-3) / lmin * 1.001
    return np.where(lnarg > -1e-3, r2, -1.0)


def block_touch(xy, r2, x0, y0, w, h):
    """Can the circle (xy, r2) touch the pixel block [x0, x0+w-1] x [y0, y0+h-1]?  (wave_touch_masks)"""
    ddx = np.maximum(np.maximum(x0 - xy[:, 0], xy[:, 0] - (x0 + w - 1)), 0.0)
    ddy = np.maximum(np.maximum(y0 - xy[:, 1], xy[:, 1] - (y0 + h - 1)), 0.0)
    return ~(ddx * ddx + ddy * ddy > r2)


def tile_tables(st, t, gx, H, W):
    """For tile t: (ids [n], contrib [256, n] bool in tile_pixel order (wave, row, lane), last [256])."""
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
    last = np.where(inside, st["n_contrib"][np.minimum(py, H - 1), np.minimum(px, W - 1)], 0).astype(np.int64)
    pos = np.arange(n)[None, :]
    contrib = (pos < last[:, None]) & (power <= 0) & (alpha >= 1.0 / 255.0)
    return ids, xy, co, contrib, last, (tx, ty)


def count_view(st, H, W):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    c = {k: 0 for k in ("pairs", "tiles", "contrib", "built_steps", "built_visits", "built_batches",
                        "sorted_rows_steps", "quad_steps", "quad_visits", "b84_steps", "b84_visits",
                        "transposed_iters", "transposed_chunks", "transposed_iters_pooled", "lane_list_steps", "pair_box_px",
                        "exact44_steps", "exact44_visits", "transposed_alive_iters")}
    for t in range(gx * gy):
        tt = tile_tables(st, t, gx, H, W)
        if tt is None:
            continue
        ids, xy, co, contrib, last, (tx, ty) = tt
        n = len(ids)
        r2 = cutoff_r2(co)
        c["pairs"] += n
        c["tiles"] += 1
        c["contrib"] += int(contrib.sum())
        tile_max = int(last.max())
        nb = (n + BATCH - 1) // BATCH
        # ---- sub-block visit masks: [16 sub-blocks (wave, row), n] ----
        sb_last = last.reshape(16, 16).max(1)                                    # row_max
        touch = np.zeros((16, n), bool)
        for sb in range(16):
            w, r = sb >> 2, sb & 3
            x0 = tx * 16 + ((w & 1) << 3) + ((r & 1) << 2)
            y0 = ty * 16 + ((w >> 1) << 3) + ((r >> 1) << 2)
            touch[sb] = block_touch(xy, r2, x0, y0, 4, 4) & (np.arange(n) < sb_last[sb])
        pooled_rows = []
        for b in range(nb):
            lo = b * BATCH
            if lo >= tile_max:
                continue
            hi = min(n, lo + BATCH)
            per_row = touch[:, lo:hi].sum(1)                                     # list length of each (wave, row) in this batch
            # (A) as built: a wave steps max over its four rows, in groups of four
            for w in range(4):
                m = int(per_row[4 * w:4 * w + 4].max())
                c["built_steps"] += (m + 3) & ~3
                c["built_batches"] += 1
            c["built_visits"] += int(per_row.sum())
            # (E) rows dealt to waves by list length across the WORKGROUP (sorted: each wave gets four similar lengths)
            srt = np.sort(per_row)[::-1]
            for w in range(4):
                c["sorted_rows_steps"] += (int(srt[4 * w]) + 3) & ~3
            # (B) transposed walk: a row = 16 consecutive entries of its sub-block's list, 16 pixel iterations per chunk;
            #     chunks per batch and row = ceil(len / 16); the four rows of a wave in lock-step
            chunks = (per_row + 15) // 16
            for w in range(4):
                c["transposed_iters"] += int(chunks[4 * w:4 * w + 4].max()) * 16
            c["transposed_chunks"] += int(chunks.sum())
            pooled_rows.append(per_row)
            # (B'') the transposed walk skipping the pixels that finished in front of the chunk: iterations of a chunk = the largest
            #      number of live pixels among the wave's four sub-blocks (live: last contributor beyond the chunk's first entry)
            for w in range(4):
                nch = int(chunks[4 * w:4 * w + 4].max())
                for k in range(nch):
                    live = 0
                    for r in range(4):
                        sb = 4 * w + r
                        idx = np.flatnonzero(touch[sb, lo:hi])
                        if 16 * k < len(idx):
                            first = lo + int(idx[16 * k])
                            live = max(live, int((last[16 * sb:16 * sb + 16] > first).sum()))
                    c["transposed_alive_iters"] += live
            # (A') as built with EXACT culling: a row visits a splat only if one of its sixteen pixels contributes
            ex = np.stack([contrib[16 * sb:16 * sb + 16, lo:hi].any(0).sum() for sb in range(16)])
            for w in range(4):
                c["exact44_steps"] += (int(ex[4 * w:4 * w + 4].max()) + 3) & ~3
            c["exact44_visits"] += int(ex.sum())
            # (C) 2x2 quads, sixteen splats per wave-step: quad lists by the same circle test; a wave = 16 quads ... 64 px
            for w in range(4):
                qlen = []
                for q in range(16):
                    x0 = tx * 16 + ((w & 1) << 3) + 2 * (q & 3)
                    y0 = ty * 16 + ((w >> 1) << 3) + 2 * (q >> 2)
                    qt = block_touch(xy[lo:hi], r2[lo:hi], x0, y0, 2, 2)
                    # positions at or beyond the last contributor of the quad's four pixels cannot matter
                    qt &= (np.arange(lo, hi) < max(int(last_px(last, w, x0 - tx * 16, y0 - ty * 16, k)) for k in range(4)))
                    qlen.append(int(qt.sum()))
                c["quad_visits"] += sum(qlen)
                # 16 quads of a wave = 16 DPP quads; step = one entry of each quad's list
                c["quad_steps"] += (max(qlen) + 3) & ~3
            # (D) 8x4 half-blocks with EXACT elliptical culling (a visit only if some pixel of the block contributes)
            for w in range(4):
                hl = []
                for hb in range(2):
                    rows = slice(64 * w + 32 * hb, 64 * w + 32 * hb + 32)
                    hl.append(int(contrib[rows, lo:hi].any(0).sum()))
                c["b84_visits"] += sum(hl)
                c["b84_steps"] += (max(hl) + 3) & ~3
            # (F) per-lane exact lists: a wave steps as often as its busiest PIXEL contributes in this batch
            per_px = contrib[:, lo:hi].sum(1).reshape(4, 64)
            c["lane_list_steps"] += int(per_px.max(1).sum())
        # (B') transposed walk with lists per TILE instead of per staged batch (padding of 16 paid once per sub-block)
        if pooled_rows:
            tot = np.sum(pooled_rows, axis=0)
            ch = (tot + 15) // 16
            for w in range(4):
                c["transposed_iters_pooled"] += int(ch[4 * w:4 * w + 4].max()) * 16
        # (G) two-pass: bounding box of the alpha >= 1/255 circle inside the tile, per pair
        rad = np.sqrt(np.maximum(r2, 0))
        bx = np.minimum(tx * 16 + 15, np.floor(xy[:, 0] + rad)) - np.maximum(tx * 16, np.ceil(xy[:, 0] - rad)) + 1
        by = np.minimum(ty * 16 + 15, np.floor(xy[:, 1] + rad)) - np.maximum(ty * 16, np.ceil(xy[:, 1] - rad)) + 1
        alive = np.arange(n) < tile_max
        c["pair_box_px"] += int((np.maximum(bx, 0) * np.maximum(by, 0) * alive).sum())
    return c


def last_px(last, w, lx, ly, k):
    """last contributor of pixel k (0..3) of the 2x2 quad at local (lx, ly) of wave w (tile_pixel order)."""
    x, y = lx + (k & 1), ly + (k >> 1)
    r = ((y >> 2) & 1) * 2 + ((x >> 2) & 1)
    i = (y & 3) * 4 + (x & 3)
    return last[64 * w + 16 * r + i]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", default="0,9,18")
    ap.add_argument("--opacity", default="A")
    a = ap.parse_args()
    from oracle import c_oracle
    from scaffold import reference_boundary as boundary, scene
    cfg = scene.CONFIGS["C2"]
    H, W = cfg["H"], cfg["W"]
    params = scene.make_gaussians(cfg["n_lat"], cfg["n_lon"], opacity=a.opacity, seed=0)
    rv = {k: v.detach() for k, v in boundary.params2rendervar(params).items()}
    cams = scene.camera_rig(H, W, n_views=24)
    views = [int(v) for v in a.views.split(",")]
    tot = None
    for v in views:
        r = c_oracle.OracleRender(cams[v], rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"])
        c = count_view(r.state(), H, W)
        tot = c if tot is None else {k: tot[k] + c[k] for k in c}
    s = 24.0 / len(views)                                   # scale to a 24-view launch
    T = {k: v * s for k, v in tot.items()}
    built = T["built_steps"] * STEP_BUILT + T["built_batches"] * BATCH_BUILT
    out = {"scene": f"C2 scenario {a.opacity}, views {views} scaled to 24", "counts_per_24_views": {k: round(v) for k, v in T.items()}}
    lanes = json.load(open(os.path.join(ROOT, "profiles", "lanes.json"))).get("C2" if a.opacity == "A" else "C2_B", {}).get("bwd")
    if lanes:
        out["calibration"] = {"wave_steps model / counting build": round(T["built_steps"] / lanes["wave_steps"], 3),
                              "row_visits": round(T["built_visits"] / lanes["row_visits"], 3),
                              "contributing_lanes": round(T["contrib"] / lanes["contributing_lane_steps"], 3),
                              "wave_batches": round(T["built_batches"] / lanes["wave_batches"], 3)}
    M = {}
    M["as_built"] = {"wave_instructions": built, "steps": T["built_steps"], "price": f"{STEP_BUILT} per step + {BATCH_BUILT} per wave-batch"}
    # (E) same kernel, rows dealt by length: the step and batch prices stay, +40 per batch for a 16-element sort and the indirection
    M["rows_sorted_by_list_length"] = {"wave_instructions": T["sorted_rows_steps"] * STEP_BUILT + T["built_batches"] * (BATCH_BUILT + 40),
                                       "steps": T["sorted_rows_steps"], "price": "55 per step + 380 per wave-batch; costs the wave its 8x8 pixel block (a wave's rows no longer tile one block: per-row pixel state, 4 tile_dot partials)"}
    # (C) 2x2 quads: alpha 14 + chain 20 + products 5 + reduce over 4 lanes (no bank masks at quad granularity: 5 pairs x (2 selects + 1 add)
    #     then 2 pairs x 3 + 1 = 22) + 3 slab read-add-writes per lane instead of 1 (+6) + 8 list/branch = 61 (alpha and chain overlap as today: 55 + 6)
    M["quads_2x2_sixteen_splats_per_step"] = {"wave_instructions": T["quad_steps"] * 61 + T["built_batches"] * (BATCH_BUILT + 250),
                                              "steps": T["quad_steps"], "price": "61 per step + 590 per wave-batch (sixteen lists per wave instead of four)"}
    # (D) 8x4 half-blocks, exact culling: two splats per step; reduce of ten sums over 32 lanes for two splats at once:
    #     xor16 level 10 plain DPP adds (row_bcast/permlane: no banked form) + the 16-lane transpose-reduce 22 -> 32; chain 20 + products 5 + 8 = 65
    M["half_blocks_8x4_exact_culling"] = {"wave_instructions": T["b84_steps"] * 65 + T["built_batches"] * BATCH_BUILT,
                                          "steps": T["b84_steps"], "price": "65 per step + 340 per wave-batch (exact culling priced at ZERO)"}
    # (B) transposed walk (lanes = sixteen splats of a sub-block's list, loop over its sixteen pixels, sums in registers, no cross-lane reduce):
    #     per iteration 2 broadcast reads + alpha 14 + T product scan 6 + prefix-sum scan 6 + q 3 + dL/dalpha, e, w 6 + nine sums 7 + loop 2 = 46;
    #     per chunk: splat record into registers 10 + slab add of nine sums 27 = 37
    per_chunk = 37 * (T["transposed_chunks"] / 4 / 0.75)          # wave-instructions: chunks are per row; four rows share a wave at the built balance
    M["transposed_walk_lists_per_batch"] = {"wave_instructions": T["transposed_iters"] * 46 + per_chunk + T["built_batches"] * BATCH_BUILT,
                                            "iterations": T["transposed_iters"], "price": "46 per pixel iteration + 37 per chunk of 16 splats + 340 per wave-batch"}
    M["transposed_walk_lists_per_tile"] = {"wave_instructions": T["transposed_iters_pooled"] * 46 + per_chunk + T["built_batches"] * BATCH_BUILT,
                                           "iterations": T["transposed_iters_pooled"], "price": "as above, sub-block lists padded to 16 once per tile (needs per-tile list storage)"}
    M["transposed_walk_skipping_finished_pixels"] = {"wave_instructions": T["transposed_alive_iters"] * 48 + per_chunk + T["built_batches"] * BATCH_BUILT,
                                                     "iterations": T["transposed_alive_iters"], "price": "48 per pixel iteration (+2: the live-pixel index) + 37 per chunk + 340 per wave-batch"}
    M["as_built_with_exact_culling"] = {"wave_instructions": T["exact44_steps"] * STEP_BUILT + T["built_batches"] * BATCH_BUILT,
                                        "steps": T["exact44_steps"], "price": "55 per step + 340 per wave-batch: the exact test priced at ZERO (it needs alpha of 16 pixels per candidate: the walk itself)"}
    M["transposed_walk_skipping_finished_pixels"] = {"wave_instructions": T["transposed_alive_iters"] * 48 + per_chunk + T["built_batches"] * BATCH_BUILT,
                                                     "iterations": T["transposed_alive_iters"], "price": "48 per pixel iteration (+2: the live-pixel index) + 37 per chunk + 340 per wave-batch"}
    M["as_built_with_exact_culling"] = {"wave_instructions": T["exact44_steps"] * STEP_BUILT + T["built_batches"] * BATCH_BUILT,
                                        "steps": T["exact44_steps"], "price": "55 per step + 340 per wave-batch: the exact test priced at ZERO (it needs alpha of 16 pixels per candidate: the walk itself)"}
    # (F) per-lane exact lists: every lane walks only the splats IT contributes to (chain 20 + alpha 14 + list 4 = 38 per step), but the ten
    #     sums of a step belong to 64 different splats: LDS float atomics retire ~3 cycles per lane (tools/micro/lds_atomic.hip: 121 cycles
    #     per 40 lanes) = 10 x 64 x 3 cycles = 1,920 cycles per step ~ 470 instruction slots at 4.1 cycles; neighbouring pixels hit the same splat
    M["per_lane_lists_lds_atomics"] = {"wave_instructions": T["lane_list_steps"] * (38 + 470) + T["built_batches"] * (BATCH_BUILT + 400),
                                       "steps": T["lane_list_steps"], "price": "38 + 470 (ten LDS float atomics per lane) per step; exact per-pixel lists priced at 400 per wave-batch"}
    # (G) two-pass: pass 1 = (F)'s walk storing (e, w) = 8 bytes per contributing (pixel, splat) [scattered 8-byte stores], pass 2 = one LANE per
    #     (Gaussian, tile) pair looping over the pixels of its cut-off box inside the tile: 1 load + 2 offsets + 9 fused multiply-adds = 12 per pixel
    #     and 64 pairs; the records must be dense over the box (zero-filled) or carry an index: bytes = box pixels x 8 written + read
    box_bytes = T["pair_box_px"] * 8
    M["two_pass_spill_e_w_then_per_pair_gather"] = {
        "wave_instructions": T["lane_list_steps"] * 42 + T["built_batches"] * (BATCH_BUILT + 400) + T["pair_box_px"] / 64 * 12,
        "extra_HBM_bytes": round(2 * box_bytes + box_bytes), "extra_HBM_us_at_6.3TBs": round(3 * box_bytes / 6.3e12 * 1e6, 1),
        "price": "pass 1: 42 per step; pass 2: 12 per box pixel and 64 pairs; + a zero-fill, a scattered write and a read of 8 B per box pixel"}
    for k, m in M.items():
        m["wave_instructions"] = round(m["wave_instructions"])
        m["vs_as_built"] = round(m["wave_instructions"] / built, 3)
        m["predicted_us_at_4.1_cycles_1024_SIMDs_2.1GHz"] = round(m["wave_instructions"] * 4.1 / 1024 / 2.1e3, 1)
    out["models"] = M
    out["gate"] = "VERDICT r04 item 3: build only a candidate predicted <= 95 M wave-instructions (as built: 128 M measured by SQ_INSTS_VALU)"
    out["verdict"] = {k: ("BUILD" if m["wave_instructions"] <= 95e6 and "extra_HBM_bytes" not in m else "no") for k, m in M.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
