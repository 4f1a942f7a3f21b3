#!/usr/bin/env python
"""
bench.py — rasterizer forward+backward views/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C4] [--opacity A|B] [--scaling weak|strong]
                    [--shard frames|views] [--allreduce-grads] [--frames-in-flight F] [--no-cpu-baseline] [--no-extras]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: forward + backward of the 24 views of one synthetic frame
(BASELINE config 2: 24 x 512x512, P = 30,000 vertex-bound Gaussians, precomputed RGB) through the C ABI, with
all inputs already resident in HBM and seeded dL/dcolor supplied.  One "view" = one
(Settings, params) -> colour/depth/alpha -> per-view dL/dparams round trip.
Frames are independent units, so F = 3 of them (config 4: 2) are in flight on as many HIP streams (own state buffers each): the latency-bound
binning kernels of one frame run beside the issue-bound render kernels of the other.  Every step is still executed in full, K
steps are timed; `sequential` in the JSON line is the same run with F = 1, and the per-kernel figures of `roofline` are taken
with F = 1 (a kernel alone on the chip).

Multi-GPU (BASELINE config 3): the 64-frame sequence is sharded by frame — rank r renders frames r, r+N, ...
  --scaling weak   (default; what the driver runs): every rank times K steps, per-GPU work is fixed (24 views per step);
  --scaling strong : the job is fixed — K frame-steps in total (default 64 = config 3's sequence), K/N per rank.
  --shard views    : the split Topo4D's REAL loop could use (frames are sequential there, train.py:646: frame t starts from
                     frame t-1): every rank steps through ALL frames and renders views r::N of each (24/8 = 3 views per rank);
                     --allreduce-grads adds the data-parallel gradient all_reduce.  The default command carries this figure too
                     (`view_sharded`), next to the frame-sharded `value` / `strong`.
The only collective is an RCCL all_gather of the per-view scalar losses (24 floats per rank per step), overlapped with the
next step.

The JSON line carries `roofline` (dominant kernel by HIP-event time, algorithmic bytes per launch, HBM traffic and the
vector-ALU counters of the committed rocprofv3 passes), `cpu_baseline` (oracle/raster_oracle.c, OpenMP, bounded sample,
min of 5), `repeats` (every figure is the median of >= 5 timed regions of >= 100 ms each: a region is the --steps block
repeated as often as that takes), `scenario_b` (unsaturated opacities), `single_view` (the reference's own call shape: one
camera per call, P = 8,280, 512x375), `small_v` (1 and 3 views of the config-2 scene per call: a view-sharded rank's launch),
`forecast` (step time at 24 / 12 / 6 / 3 views => view-sharded strong scaling at 2 / 4 / 8 GPUs), `view_sharded` (the same
split executed: rank r renders views r::N of every frame), `c4` (BASELINE config 4 with its own roofline), `dense_1m` (one view,
P = 10^6, 4096x3008; `camera_4`: the same from the camera that sees the scene's longest tile lists; `texture_iteration`: one full
iteration of the texture loop, train.py:729-741, at that size), `loss` (the
fused photometric loss at three shapes with its own roofline), `bake_8192` (BASELINE config 5 with its own roofline and the
reference's own code timed beside it), `full_iteration` (render + fused loss + Adam/pins), `drop_in` and `sequential` — see
DESIGN.md §6.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STRONG_JOB_FRAMES = 64          # BASELINE config 3: the 64-frame synthetic sequence
PEAK_HBM_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable
ACHIEVABLE_HBM_GBS = 6290.0


def algorithmic_bytes(P, R, HW, S=0):
    """SURVEY.md §8d, per view.  Returns (per-kernel dict, whole-pipeline bytes)."""
    per_kernel = {
        "k_preprocess": P * (56 + S) + P * 60,            # param read + geometry-state write
        "k_scan_tiles": 0,
        "k_scatter": R * 12,                              # key/value write
        "k_sort_tiles": R * 24,                           # one sort pass, read + write (lower bound)
        "k_render_fwd": R * 40 + HW * 28,                 # per-tile gather + image/state outputs
        "k_render_bwd": HW * 32 + R * 44 + P * 44,        # pixel grads/state read + gather + intermediate grads write
        "k_preprocess_bwd": P * 44 + P * (80 + S) + P * (68 + S),
    }
    # kernels that only run where tile lists are long (dense passes) work on a share of the bytes counted above for the kernel
    # they relieve: no algorithmic bytes of their own
    per_kernel.update({"k_sort_long": 0, "k_fwd_long": 0, "k_render_bwd_long": 0})
    total = P * (352 + 3 * S) + R * 120 + HW * 60
    return per_kernel, total


def cpu_baseline(cfg, n_sample_views, reps):
    """C oracle (oracle/raster_oracle.c), OpenMP over all host cores, fwd+bwd on a bounded sample of the workload; min of
    `reps` repetitions (SURVEY.md §8d asks for min-of-5)."""
    from oracle import c_oracle as CO
    from scaffold import reference_boundary as boundary, scene
    CO.build()
    params = scene.make_gaussians(cfg["n_lat"], cfg["n_lon"], opacity="A", sh_degree=cfg["sh_degree"], seed=0)
    rv = {k: v.detach() for k, v in boundary.params2rendervar(params).items()}
    if cfg["sh_degree"] is not None:
        rv["shs"] = params["shs"]
        rv.pop("colors_precomp")
    cams = scene.camera_rig(cfg["H"], cfg["W"], n_views=cfg["n_views"], true_campos=cfg["sh_degree"] is not None)
    if cfg["sh_degree"] is not None:
        cams = [c._replace(sh_degree=cfg["sh_degree"]) for c in cams]
    dc, _, _ = scene.output_cotangents(n_sample_views, cfg["H"], cfg["W"], seed=0)
    # warm-up (page-in, OpenMP pool)
    r = CO.OracleRender(cams[0], rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"],
                        rv.get("colors_precomp"), rv.get("shs"))
    r.backward(dc[0])
    # views are independent: run them concurrently, each on its share of the host threads (one view alone cannot feed 256
    # threads: 1,024 tiles of which a third are non-empty).  ctypes releases the GIL during the C calls.
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    workers = max(1, min(n_sample_views, cores // 8))
    per = max(1, cores // workers)
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp, workers, per = None, 1, cores

    def one(v):
        if gomp is not None:
            gomp.omp_set_num_threads(per)                      # per calling thread (OpenMP ICV)
        rr = CO.OracleRender(cams[v % len(cams)], rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"],
                             rv.get("colors_precomp"), rv.get("shs"))
        rr.backward(dc[v])

    best, spent = None, 0.0
    done_reps = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(one, range(n_sample_views)))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        spent += dt
        done_reps += 1
        if spent > 25.0:                                       # bounded: the default run must finish within minutes
            break
    out = {"value": round(n_sample_views / best, 3), "unit": "views/s", "cores": cores, "kind": "port",
           "sample": f"{n_sample_views} of the {cfg['n_views']} views of the same scene, fwd+bwd, oracle/raster_oracle.c -O3 "
                     f"-fopenmp, {workers} views at a time x {per} OpenMP threads, min of {done_reps} runs ({best:.2f}s wall each)"}
    # the same code on ONE host thread, one view (SURVEY.md 8d asks for both figures)
    try:
        gomp.omp_set_num_threads(1)
        d1 = None
        for _ in range(3):
            t1 = time.perf_counter()
            r = CO.OracleRender(cams[0], rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"],
                                rv.get("colors_precomp"), rv.get("shs"))
            r.backward(dc[0])
            dd = time.perf_counter() - t1
            d1 = dd if d1 is None else min(d1, dd)
            if dd > 5.0:
                break
        gomp.omp_set_num_threads(os.cpu_count())
        out["one_thread"] = {"value": round(1.0 / d1, 4), "unit": "views/s", "cores": 1, "sample": f"1 view, min of up to 3 ({d1:.2f}s wall)"}
    except Exception:                            # libgomp not loadable: report the multi-threaded figure only
        out["one_thread"] = None
    return out


class Workload:
    """The timed thing: `step(i)` = forward + backward of this rank's frame i (all V views in one launch set) + the per-view
    loss scalars + (N > 1) their asynchronous all_gather.  Frames are independent units (SURVEY 8e shards them over GPUs for the
    same reason): `in_flight` of them are in flight on as many HIP streams, each with its own state buffers, so that the
    latency-bound binning kernels of one frame run beside the issue-bound render kernels of the other."""

    def __init__(self, cfgname, opacity, dev, rank=0, world=1, in_flight=2, n_frames=64, resident=8, gather=None,
                 view_shard=None, allreduce_grads=False, n_views=None, frames_per_launch=1):
        from scaffold import reference_boundary as boundary, scene
        from topo4d_amd import ViewBatch, dist as t4d_dist, pack_views
        self.t4d_dist = t4d_dist
        self.cfg = cfg = dict(scene.CONFIGS[cfgname])
        self.dev, self.rank, self.world = dev, rank, world
        # the loss all_gather runs whenever a process group exists - also a world-size-1 group (torchrun --nproc-per-node 1),
        # so that the RCCL branch can be executed on a one-GPU box
        self.gather = (world > 1) if gather is None else bool(gather)
        self.H, self.W, self.V = H, W, V = cfg["H"], cfg["W"], cfg["n_views"]
        params = scene.make_gaussians(cfg["n_lat"], cfg["n_lon"], opacity=opacity, sh_degree=cfg["sh_degree"], seed=0)
        self.P = params["means3D"].shape[0]
        base_means = params["means3D"].clone()
        cams = scene.camera_rig(H, W, n_views=V, device=dev, true_campos=cfg["sh_degree"] is not None)
        if cfg["sh_degree"] is not None:
            cams = [c._replace(sh_degree=cfg["sh_degree"]) for c in cams]
        dc, _, _ = scene.output_cotangents(V, H, W, seed=0)
        # view_shard = (r, N): this workload renders views r::N of EVERY frame (the split of SURVEY 8e / BASELINE config 3);
        # n_views = k: the first k::(24/k)-strided views (the launch of one rank of a 24/k-way view shard, for `forecast`)
        self.allreduce_grads = allreduce_grads
        self.full_views = V
        if view_shard is not None:
            mine = t4d_dist.shard_units(V, view_shard[0], view_shard[1])
        elif n_views is not None and n_views < V:
            mine = list(range(0, V, V // n_views))[:n_views]
        else:
            mine = list(range(V))
        self.my_views = mine
        cams = [cams[i] for i in mine]
        dc = dc[mine]
        self.V = V = len(mine)
        # frames_per_launch = G > 1: ONE launch set carries this workload's views of G consecutive frames, each frame with its own
        # Gaussians (T4DProblem.views_per_param_set).  The frames of config 3's synthetic sequence are independent units,
        # so a view-sharded rank - 3 cameras per frame - launches 3 x 8 views at the efficiency of a 24-view launch.  A step is then G frames.
        self.G = G = max(1, int(frames_per_launch))
        views = pack_views(cams, dev)
        if G > 1:
            views = views.repeat(G, 1)
            dc = dc.repeat(G, 1, 1, 1)
        self.dc = dc.to(dev).contiguous()
        # T4D_BENCH_DA=1 (experiments; never the default): depth and alpha cotangents too - the backward's DA = true instantiation
        # (Topo4D discards depth and alpha, train.py:307: its backward never runs it)
        self.dd = self.da = None
        if os.environ.get("T4D_BENCH_DA") == "1":
            _, dd, da = scene.output_cotangents(self.full_views, H, W, seed=0, depth_alpha=True)
            self.dd, self.da = dd[mine].repeat(G, 1, 1, 1).to(dev).contiguous(), da[mine].repeat(G, 1, 1, 1).to(dev).contiguous()
        # per-frame Gaussians of the synthetic 64-frame sequence (config 3); all resident in HBM before timing
        self.n_frames = n_frames
        my_frames = t4d_dist.shard_units(n_frames, rank, world) if view_shard is None else list(range(n_frames))
        self.rv_frames = []
        for t in my_frames[: max(1, min(len(my_frames), resident * G))]:
            p = dict(params)
            p["means3D"] = scene.frame_displacement(base_means, t, n_frames)
            rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
            if cfg["sh_degree"] is not None:
                rv["shs"] = params["shs"].to(dev)
                rv.pop("colors_precomp")
            self.rv_frames.append(rv)
        if G > 1:
            # groups of G consecutive frames, every per-Gaussian input stacked [G, P, .] (the last group wraps around)
            fr = self.rv_frames
            n_groups = max(1, len(fr) // G)
            self.rv_frames = [{k: torch.stack([fr[(j * G + i) % len(fr)][k] for i in range(G)], 0).contiguous() for k in fr[0]}
                              for j in range(n_groups)]
        self.F = F = max(1, in_flight)
        V = self.V * G                    # views per launch set
        # one slot per launch set in flight: its own state buffers (ViewBatch), stream, loss and gather buffers
        self.batches = [ViewBatch(views.contiguous(), H, W, 1.0, cfg["sh_degree"] or 0, param_sets=G) for _ in range(F)]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(F)] if F > 1 else [None]
        # multi-GPU: the loss all_gather of a step overlaps with the following steps; a slot waits for its previous gather
        # before it rewrites its buffers (F = 1: two buffer sets, the gather of step i is waited on at step i + 2)
        self.sequential = False          # True: one frame after the other on the current stream (per-kernel event pass, comparison run)
        self.n_bufs = nb = max(2, F)
        self.loss_bufs = [torch.zeros(V, device=dev) for _ in range(nb)]
        self.gath_bufs = [torch.zeros(V * world, device=dev) for _ in range(nb)]
        self.pending = [None] * nb
        # everything above was uploaded on the current stream: the side streams' first steps must not start before it
        for st in self.streams:
            if st is not None:
                st.wait_stream(torch.cuda.current_stream(dev))

    def step(self, i):
        rv = self.rv_frames[i % len(self.rv_frames)]
        slot, k = (0 if self.sequential else i % self.F), i % self.n_bufs
        with torch.cuda.stream(self.streams[slot]) if self.F > 1 and not self.sequential else contextlib.nullcontext():
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None
            losses = self.loss_bufs[k]
            b = self.batches[slot]
            b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv.get("colors_precomp"), rv.get("shs"))
            # per-view scalar loss term <colour, dL/dcolour>: the backward's replay ends holding exactly this inner product per
            # pixel, so it comes out of t4d_rasterize_backward (cotangent_dot), not out of a second pass over both images
            g = b.backward(self.dc, self.dd, self.da, cotangent_dot=losses)
            if self.allreduce_grads:
                # data-parallel training step over a view shard: the view-summed parameter gradients are summed over the ranks
                # (~14 floats x P; changes train.py:661-673's one-Adam-step-per-view schedule, so it is optional)
                # (several frames per launch set: views are frame-major - the sum runs over a frame's views, one gradient per frame)
                summed = [t.reshape(self.G, self.V, *t.shape[1:]).sum(1) for t in g.values() if t is not None]
                self.t4d_dist.all_reduce_grads(summed)
            if self.gather:
                out, work = self.t4d_dist.gather_losses_async(losses, self.gath_bufs[k])
                self.pending[k] = work
                return out, [g]
        return losses, [g]

    def drain(self):
        for k in range(self.n_bufs):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

    def learn_capacity(self):
        """first call "checked" (learns the pair-arena capacity), the rest of the run is lazy (no host sync)"""
        import topo4d_amd
        topo4d_amd.set_sync_mode("checked")
        self.step(0)
        for i in range(len(self.rv_frames)):
            self.step(i)
        topo4d_amd.set_sync_mode("lazy")

    def statuses(self):
        """binning status of the last frame each slot rendered (every slot renders all V views of a frame)"""
        return [b.fetch_status() for b in self.batches]


MIN_REGION_S = 0.1      # a timed region lasts at least this long: the --steps block is repeated as often as that takes
N_REGIONS = 5           # ... and there are this many regions; figures are the median, min and max ride along


def timed_run(wl, steps, warmup, prewarm_s, barrier, all_reduce_max, regions=N_REGIONS, min_region_s=MIN_REGION_S):
    """W untimed steps (after a clock-settling pre-warm), then `regions` timed regions, each EXACTLY `blocks * steps` steps
    between two barriers (barrier + torch.cuda.synchronize on both sides, MAX over ranks).  Returns (median seconds per
    block of `steps` steps, host enqueue seconds per block, stats dict)."""
    dev = wl.dev
    # A GPU that has just been idle (fresh box, or a profiler run before this one) needs tens of milliseconds of work
    # before its clocks settle: a cold 50-step run measured 1.23 ms/step against 0.62 warm.  Untimed, like the W steps.
    torch.cuda.synchronize(dev)
    t_pre = time.perf_counter()
    for i in range(8):
        wl.step(i)
    torch.cuda.synchronize(dev)
    est = all_reduce_max((time.perf_counter() - t_pre) / 8)          # every rank must run the same number of steps
    n_pre = int(min(4000, max(0.0, prewarm_s) / max(est, 1e-5)))
    for i in range(n_pre):
        wl.step(i)
    torch.cuda.synchronize(dev)
    blocks = max(1, int(-(-min_region_s // max(est * steps, 1e-6)))) if min_region_s > 0 else 1
    blocks = int(all_reduce_max(float(min(blocks, 100000))))
    for i in range(warmup):
        wl.step(i)
    times, enq = [], []
    for _ in range(max(1, regions)):
        barrier()
        t0 = time.perf_counter()
        for _b in range(blocks):
            for i in range(steps):
                wl.step(i)
        t_enqueue = time.perf_counter() - t0       # host time to enqueue the steps (GPU still running)
        barrier()
        dt = time.perf_counter() - t0
        times.append(all_reduce_max(dt) / blocks)
        enq.append(t_enqueue / blocks)
    srt = sorted(times)
    med = srt[len(srt) // 2]
    stats = {"regions": len(times), "blocks_per_region": blocks, "steps_per_region": blocks * steps,
             "region_ms": round(1e3 * med * blocks, 3),
             "ms_per_step": {"min": round(1e3 * srt[0] / steps, 4), "median": round(1e3 * med / steps, 4), "max": round(1e3 * srt[-1] / steps, 4)},
             "note": "a region = the --steps block repeated until it lasts >= 100 ms, bracketed by barrier + synchronize; value is the MEDIAN region"}
    return med, sorted(enq)[len(enq) // 2], stats


def kernel_profile(wl, steps):
    """Per-kernel durations with HIP events recorded on the launch stream inside the library (same steps again, so that the
    timed region carries no event overhead, and with ONE frame in flight, so that a kernel's duration is its own).  Returns
    ({kernel: (total_ms, launches)}, wall seconds)."""
    from topo4d_amd import _lib
    torch.cuda.synchronize(wl.dev)
    wl.sequential = True             # a kernel's own duration: one frame at a time, nothing else on the chip
    if wl.rank == 0:
        _lib.profile_begin()
    tp0 = time.perf_counter()
    for i in range(steps):
        wl.step(i)
    wl.drain()
    torch.cuda.synchronize(wl.dev)
    tp = time.perf_counter() - tp0
    prof = _lib.profile_end() if wl.rank == 0 else {}
    wl.sequential = False
    return prof, tp


def load_profile_json(name, config, kernel):
    f = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(f):
        return None
    try:
        return json.load(open(f)).get(config, {}).get(kernel)
    except Exception:
        return None


def counters_provenance(config, lanes_key=None):
    """The rocprofv3 counters in profiles/*.json were taken by separate runs (tools/prof.sh, tools/count_lanes.py): "current" when
    they were taken on the kernel source this run executes (sha256 over csrc/t4d_raster.hip and its t4d_raster_*.h parts:
    topo4d_amd.build.raster_source_sha256), otherwise "stale"."""
    try:
        from topo4d_amd.build import raster_source_sha256
        now = raster_source_sha256()
    except OSError:
        return {"kernel_source_sha256": None}
    out = {"kernel_source_sha256": now[:16]}
    for name in ("traffic.json", "valu.json", "lanes.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            then = (d.get(lanes_key or config) or {}).get("_kernel_source_sha256") if name == "lanes.json" else (d.get("_kernel_source_sha256") or {}).get(config)
            out[name] = "missing" if then is None else ("current" if then == now else "stale")
        except Exception:
            out[name] = "missing"
    return out


def _min_of(fn, reps=5):
    best = None
    for _ in range(reps):
        x = fn()
        best = x if best is None or x < best else best
    return best


def drop_in_probe(dev, reps=5):
    """The schedule Topo4D runs (train.py:661-673) through the UNMODIFIED drop-in: one camera per iteration, `params2rendervar`
    (the reference's five torch ops, helpers.py:91-100) -> GaussianRasterizer(raster_settings=cam)(**rendervar) -> backward with
    a supplied dL/dcolour.  P = 8,280, 512x375, the drop-in's default sync mode.  Host-bound: iterations/s (best of `reps` runs of
    400 iterations; the spread rides along), and the same loop without the torch ops of params2rendervar around it."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from scaffold import reference_boundary as boundary, scene
    from topo4d_amd import rasterizer
    H, W = 512, 375
    p = scene.make_gaussians(69, 120, opacity="A", seed=0)
    params = {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}
    cams = scene.camera_rig(H, W, n_views=24, device=dev)
    g = torch.Generator().manual_seed(0)
    dcs = [(torch.randn(3, H, W, generator=g) / (3 * H * W)).to(dev) for _ in range(24)]
    rv_fixed = {k: v.detach().clone().requires_grad_(True) for k, v in boundary.params2rendervar(params).items()}

    def it_full(i):
        rv = boundary.params2rendervar(params)
        im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv)
        im.backward(dcs[i % 24])

    def it_raster(i):
        im, radius, _, _ = Renderer(raster_settings=cams[i % 24])(**rv_fixed)
        im.backward(dcs[i % 24])

    saved = rasterizer._save_sync_mode()
    rasterizer._restore_sync_mode(("checked", False))      # what an unmodified train.py gets: the drop-in's default mode
    out = {"workload": "1 camera per iteration, P=8280, 512x375, params2rendervar -> GaussianRasterizer -> backward (train.py:661-673)",
           "sync_mode": rasterizer.get_sync_mode(drop_in=True) + " (drop-in default: un-synchronised forward; its own backward looks at the binning status without waiting and refuses the gradients of a truncated render)",
           "autograd_engine_thread": bool(torch.autograd.is_multithreading_enabled())}
    try:
        for name, fn in (("it_per_s", it_full), ("it_per_s_without_params2rendervar", it_raster)):
            for i in range(60):
                fn(i)
            torch.cuda.synchronize(dev)
            n = 400
            runs = []
            for _ in range(reps):
                t0 = time.perf_counter()
                for i in range(n):
                    fn(i)
                torch.cuda.synchronize(dev)
                runs.append(n / (time.perf_counter() - t0))
            out[name] = round(sorted(runs)[len(runs) // 2], 1)
            out[name + "_min_max"] = [round(min(runs), 1), round(max(runs), 1)]
    finally:
        rasterizer._restore_sync_mode(saved)
    out["note"] = ("host-bound: an iteration is the reference's own torch ops (params2rendervar forward + autograd backward: ~4 k it/s on "
                   "their own) plus the drop-in call (it_per_s_without_params2rendervar); median of the runs; GPU time per view is "
                   "single_view.gpu_us_per_view, the loop without autograd is full_iteration; importing the drop-in module runs the backward on "
                   "the calling thread (autograd_engine_thread false): with torch's worker thread the same loop ran at 4-5 k OR 8-9 k it/s "
                   "(profiles/r06_dropin_regimes.txt)")
    return out


SHAPES = {   # small-launch probes: (n_lat, n_lon, H, W)
    "topo4d": (69, 120, 512, 375),            # P = 8,280, helpers.py:807: (3, 512, 375) images
    "c2": (150, 200, 512, 512),               # the config-2 scene: P = 30,000, 512 x 512
}


def render_probe(dev, shape, n_views, reps=5, iters=200):
    """V views per call of a small scene through the C ABI, forward + backward: GPU time per call = sum of the HIP-event durations
    of the rasterizer's kernels, and wall time per un-synchronised call pair; both the MINIMUM over `reps` runs of `iters` calls
    (a single run of this probe once read 220 us where every other read 103-113: one-shot side probes are not measurements)."""
    import topo4d_amd
    from scaffold import reference_boundary as boundary, scene
    from topo4d_amd import ViewBatch, _lib, pack_views
    n_lat, n_lon, H, W = SHAPES[shape]
    p = scene.make_gaussians(n_lat, n_lon, opacity="A", seed=0)
    cams = scene.camera_rig(H, W, n_views=24, device=dev)
    sel = [12] if n_views == 1 else list(range(0, 24, 24 // n_views))[:n_views]
    rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
    g = torch.Generator().manual_seed(0)
    dc = (torch.randn(n_views, 3, H, W, generator=g) / (3 * H * W)).to(dev)
    b = ViewBatch(pack_views([cams[i] for i in sel], dev), H, W)
    f = lambda: (b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b.backward(dc))
    saved = topo4d_amd.rasterizer._save_sync_mode()
    try:
        topo4d_amd.set_sync_mode("checked")
        f()
        topo4d_amd.set_sync_mode("lazy")
        for _ in range(100):
            f()
        torch.cuda.synchronize(dev)
        walls, gpus, kern_best = [], [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(iters):
                f()
            torch.cuda.synchronize(dev)
            walls.append(1e6 * (time.perf_counter() - t0) / iters)
            _lib.profile_begin()
            for _ in range(iters):
                f()
            torch.cuda.synchronize(dev)
            prof = _lib.profile_end()
            kern = {name: round(1e3 * ms / max(cnt, 1), 2) for name, (ms, cnt) in prof.items() if cnt}
            gpus.append(sum(kern.values()))
            if kern_best is None or gpus[-1] <= min(gpus):
                kern_best = kern
        st = b.fetch_status()
    finally:
        topo4d_amd.rasterizer._restore_sync_mode(saved)
    return {"views_per_call": n_views, "gpu_us_per_call": round(min(gpus), 1), "gpu_us_runs": [round(x, 1) for x in gpus],
            "kernels_us": kern_best, "wall_us_per_call": round(min(walls), 1), "wall_us_runs": [round(x, 1) for x in walls],
            "pairs": int(st.total_pairs), "longest_tile_list": int(st.max_tile_pairs)}


def single_view_probe(dev):
    """The reference's own call shape (train.py:661-673): ONE camera per call, P = 8,280, 512x375."""
    r = render_probe(dev, "topo4d", 1)
    return {"workload": "1 view per call, P=8280, 512x375 (HxW), opacity scenario A, forward+backward through the C ABI",
            "gpu_us_per_view": r["gpu_us_per_call"], "gpu_us_runs": r["gpu_us_runs"], "kernels_us": r["kernels_us"],
            "wall_us_per_view_unsynchronised_calls": r["wall_us_per_call"], "wall_us_runs": r["wall_us_runs"],
            "views_per_s_wall": round(1e6 / r["wall_us_per_call"], 1), "pairs": r["pairs"], "longest_tile_list": r["longest_tile_list"],
            "note": "gpu_us = sum of HIP-event kernel durations; min over 5 runs of 200 calls"}


def small_v_probe(dev):
    """1 and 3 views of the config-2 scene per call: the launch of one rank of an 8-way view shard (24 / 8 = 3 views)."""
    out = {"workload": "V views per call of the config-2 scene (P=30000, 512x512), forward+backward through the C ABI, lazy sync"}
    for v in (1, 3):
        out[f"v{v}"] = render_probe(dev, "c2", v)
    return out


def forecast_probe(dev, in_flight):
    """What view sharding (BASELINE config 3: 'views sharded across 8 MI355X') gives, measured on ONE GPU: what one rank of a
    1 / 2 / 4 / 8-way view shard runs - its 24 / 12 / 6 / 3 cameras of N consecutive frames per launch set (the frames of the
    64-frame synthetic sequence are independent: each carries its own Gaussians, T4DProblem.views_per_param_set), `in_flight` sets
    in flight.  Predicted strong-scaling speed-up at N GPUs = N * t_set(1) / t_set(N): the job's 64 frames are 64 / N launch
    sets per rank (no collective cost included: the loss gather is 24 floats per frame).
    `sequential_frames`: the same split with ONE frame per launch on one stream - the schedule Topo4D's real loop allows, whose
    frames are sequential (train.py:646): speed-up = t(24 views) / t(24 / N views)."""
    sync = lambda: torch.cuda.synchronize(dev)
    out = {"workload": f"config-2 scene; rank 0 of an N-way view shard: 24 / N cameras x N frames per launch set, {in_flight} sets in flight", "steps": 24}
    t = {}
    for n in (1, 2, 4, 8):
        wl = Workload("C2", "A", dev, in_flight=in_flight, n_views=24 // n, resident=2, frames_per_launch=n)
        wl.learn_capacity()
        dt, _, st = timed_run(wl, 24, 3, 0.05, sync, lambda x: x, regions=5, min_region_s=0.05)
        t[n] = dt / 24
        out[f"ms_per_launch_set_n{n}"] = st["ms_per_step"]["median"]
        del wl
    out["predicted_view_sharded_speedup"] = {str(n): round(n * t[1] / t[n], 2) for n in (2, 4, 8)}
    out["predicted_views_per_s"] = {str(n): round(24 * n / t[n], 1) for n in (1, 2, 4, 8)}
    seq = {"workload": "V views of ONE frame per launch, one frame at a time on one stream", "steps": 50}
    ts = {}
    for v in (24, 12, 6, 3):
        wl = Workload("C2", "A", dev, in_flight=1, n_views=v, resident=2)
        wl.learn_capacity()
        dt, _, st = timed_run(wl, 50, 3, 0.05, sync, lambda x: x, regions=5, min_region_s=0.05)
        ts[v] = dt / 50
        seq[f"ms_per_step_v{v}"] = st["ms_per_step"]["median"]
        del wl
    seq["predicted_view_sharded_speedup"] = {str(n): round(ts[24] / ts[24 // n], 2) for n in (2, 4, 8)}
    seq["predicted_views_per_s"] = {str(n): round(24 / ts[24 // n], 1) for n in (1, 2, 4, 8)}
    out["sequential_frames"] = seq
    return out


def c4_probe(dev, steps=6):
    """BASELINE config 4 (24 x 2048^2, P = 120,000, SH degree 3: 'the 1-MI355X HBM-roofline run') in the default command."""
    wl = Workload("C4", "A", dev, in_flight=2, resident=2)
    wl.learn_capacity()
    dt, _, st = timed_run(wl, steps, 2, 0.1, lambda: torch.cuda.synchronize(dev), lambda x: x, regions=5, min_region_s=0.0)
    wl.sequential = True
    dts, _, sts = timed_run(wl, steps, 1, 0.0, lambda: torch.cuda.synchronize(dev), lambda x: x, regions=3, min_region_s=0.0)
    wl.sequential = False
    sts_ = wl.statuses()
    prof, _ = kernel_profile(wl, steps)
    cfg, V, P, H, W = wl.cfg, wl.V, wl.P, wl.H, wl.W
    S = 3 * (cfg["sh_degree"] + 1) ** 2 * 4
    R_view = sts_[0].total_pairs / V
    per_kernel, total_bytes = algorithmic_bytes(P, R_view, H * W, S)
    kernels = {n: {"avg_us": round(1e3 * ms / c, 1), "alg_GBs": round(per_kernel[n] * V / (1e-3 * ms / c) / 1e9, 1)} for n, (ms, c) in prof.items() if c}
    dom = max(kernels, key=lambda k: kernels[k]["avg_us"])
    ach = kernels[dom]["alg_GBs"]
    valu = load_profile_json("valu.json", "C4", dom)
    return {"workload": f"C4: {V} views x {H}x{W}, P={P}, SH degree 3, opacity scenario A, forward+backward, per-view gradients, 2 frames in flight",
            "steps": steps, "ms_per_step": st["ms_per_step"], "value": round(V * steps / dt, 1), "unit": "views/s",
            "sequential_ms_per_step": sts["ms_per_step"]["median"], "pairs_per_view": int(R_view), "overflow": bool(any(x.overflow for x in sts_)),
            "roofline": {"bound": ("valu" if valu and valu.get("valu_busy", 0) > 0.6 else "hbm"), "kernel": dom, "achieved": ach, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": load_profile_json("traffic.json", "C4", dom),
                         "alg_bytes_per_launch": int(per_kernel[dom] * V), "avg_us": kernels[dom]["avg_us"], "kernels": kernels,
                         "pipeline_alg_bytes_per_view": int(total_bytes),
                         "pipeline_frac_of_peak": round(V * steps / dt * total_bytes / 1e9 / PEAK_HBM_GBS, 4),
                         "counters": counters_provenance("C4")}}


def dense_1m_probe(dev, reps=5):
    """Topo4D's texture pass shape (train.py:729-741, params2rendervar_dense): ONE view per call, P = 10^6 at 4096 x 3008."""
    import topo4d_amd
    from scaffold import reference_boundary as boundary, scene
    from topo4d_amd import ViewBatch, _lib, pack_views
    H, W = 3008, 4096
    p = scene.make_gaussians(1000, 1000, opacity="A", seed=0)
    cams = scene.camera_rig(H, W, n_views=24, device=dev)[12:13]
    rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
    dc = torch.randn(1, 3, H, W, device=dev) / (3 * H * W)
    b = ViewBatch(pack_views(cams, dev), H, W)
    f = lambda: (b.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b.backward(dc))
    saved = topo4d_amd.rasterizer._save_sync_mode()
    try:
        topo4d_amd.set_sync_mode("checked")
        f()
        st = b.fetch_status()
        topo4d_amd.set_sync_mode("lazy")
        for _ in range(3):
            f()
        torch.cuda.synchronize(dev)
        walls = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(10):
                f()
            torch.cuda.synchronize(dev)
            walls.append(1e3 * (time.perf_counter() - t0) / 10)
        _lib.profile_begin()
        for _ in range(10):
            f()
        torch.cuda.synchronize(dev)
        kern = {n: round(1e3 * ms / c, 1) for n, (ms, c) in _lib.profile_end().items() if c}
    finally:
        topo4d_amd.rasterizer._restore_sync_mode(saved)
    out = {"workload": "1 view per call, P=1000000, 4096x3008 (WxH), opacity scenario A, forward+backward through the C ABI",
           "ms_per_view": round(min(walls), 3), "ms_runs": [round(x, 3) for x in walls], "views_per_s": round(1e3 / min(walls), 1),
           "pairs": int(st.total_pairs), "longest_tile_list": int(st.max_tile_pairs), "kernels_us": kern}
    # roofline of this one-view launch (SURVEY 8d's bytes with P = 10^6, the measured R, 12.3 M pixels): the pipeline as a whole and
    # the dominant kernel - the backward replay, whose long tiles run in a launch of their own (k_render_bwd_long: same bytes, both
    # durations).  The forward is k_render_fwd + the three depth-parallel launches of the long tiles (k_fwd_long).
    per_k, total_b = algorithmic_bytes(1000000, int(st.total_pairs), H * W, 0)
    t_bwd = kern.get("k_render_bwd", 0.0) + kern.get("k_render_bwd_long", 0.0)
    t_fwd = kern.get("k_render_fwd", 0.0) + kern.get("k_fwd_long", 0.0)
    t_sort = kern.get("k_sort_tiles", 0.0) + kern.get("k_sort_long", 0.0)
    gbs = lambda b, us: round(b / (us * 1e-6) / 1e9, 1) if us > 0 else None
    ach = gbs(per_k["k_render_bwd"], t_bwd)
    out["roofline"] = {"bound": "hbm", "kernel": "k_render_bwd + k_render_bwd_long (the long tiles' depth-segmented replay)", "achieved": ach,
                       "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4) if ach else None,
                       "traffic": load_profile_json("traffic.json", "DENSE_1M_1V", "k_render_bwd"),
                       "alg_bytes_per_launch": int(per_k["k_render_bwd"]), "avg_us": round(t_bwd, 1),
                       "stages": {"render_bwd": {"us": round(t_bwd, 1), "alg_GBs": ach},
                                  "render_fwd": {"us": round(t_fwd, 1), "alg_GBs": gbs(per_k["k_render_fwd"], t_fwd)},
                                  "sort": {"us": round(t_sort, 1), "alg_GBs": gbs(per_k["k_sort_tiles"], t_sort)},
                                  "preprocess": {"us": kern.get("k_preprocess"), "alg_GBs": gbs(per_k["k_preprocess"], kern.get("k_preprocess", 0.0))},
                                  "preprocess_bwd": {"us": kern.get("k_preprocess_bwd"), "alg_GBs": gbs(per_k["k_preprocess_bwd"], kern.get("k_preprocess_bwd", 0.0))}},
                       "pipeline_alg_bytes_per_view": int(total_b),
                       "pipeline_frac_of_peak": round(total_b / (min(walls) * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                       "counters": counters_provenance("DENSE_1M_1V")}
    # The same scene from the rig's camera 4, which looks at a polar cap of the lat-long head: hundreds of tiles with lists of
    # 2,000 - 9,000 pairs (camera 12 above: one of 12,614).  Same wave-steps; what differs is how long the longest tiles keep
    # their workgroups (HISTORY.md section 5 item 1: their backward is cut into depth segments).
    try:
        b4 = ViewBatch(pack_views(scene.camera_rig(H, W, n_views=24, device=dev)[4:5], dev), H, W)
        f4 = lambda: (b4.forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"], rv["colors_precomp"]), b4.backward(dc))
        saved = topo4d_amd.rasterizer._save_sync_mode()
        try:
            topo4d_amd.set_sync_mode("checked")
            f4()
            st4 = b4.fetch_status()
            topo4d_amd.set_sync_mode("lazy")
            for _ in range(3):
                f4()
            torch.cuda.synchronize(dev)
            _lib.profile_begin()
            for _ in range(10):
                f4()
            torch.cuda.synchronize(dev)
            k4 = {n: round(1e3 * ms / c, 1) for n, (ms, c) in _lib.profile_end().items() if c}
        finally:
            topo4d_amd.rasterizer._restore_sync_mode(saved)
        out["camera_4"] = {"longest_tile_list": int(st4.max_tile_pairs), "kernels_us": k4, "sum_us": round(sum(k4.values()), 1)}
        del b4
    except Exception as e:
        out["camera_4"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    del b, rv, dc
    torch.cuda.empty_cache()
    try:
        out["texture_iteration"] = texture_iteration_probe(dev, p, cams[0], H, W)
    except Exception as e:                                   # (never takes the render's numbers down with it)
        out["texture_iteration"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def texture_iteration_probe(dev, p, cam, H, W, n_iters=10, reps=3):
    """HOT LOOP 2 at its real size (train.py:729-741): pins on dense_rgb_colors -> render of the dense set -> 0.8 L1 + 0.2 (1-SSIM)
    without the camera affine + 0.02 soft colour -> backward -> Adam, one full-resolution view per iteration, P = 10^6 at 4096 x 3008:
    loop.optimise_dense_views chained by hand (no autograd).  ms per iteration, best of `reps` runs of `n_iters` iterations."""
    import topo4d_amd
    from topo4d_amd import loop as t4d_loop
    from topo4d_amd.optim import FusedAdamPins
    P = p["means3D"].shape[0]
    params = {"dense_" + k: torch.nn.Parameter(v.to(dev).contiguous()) for k, v in p.items()}
    params["dense_means3D"].requires_grad_(False)                                  # train.py:259-261
    lrs = {"dense_means3D": 0.0, "dense_unnorm_rotations": 0.001, "dense_logit_opacities": 0.0, "dense_log_scales": 0.0,
           "dense_rgb_colors": 0.0025}                                             # train.py:281-285
    opt = FusedAdamPins([{"params": [v], "name": k, "lr": lrs[k]} for k, v in params.items()], eps=1e-15)
    frozen = torch.zeros(P, dtype=torch.bool, device=dev)
    frozen[::5] = True
    opt.set_pin("dense_rgb_colors", frozen, 0.0)                                   # train.py:732-734
    variables = {"dense_init_colors": params["dense_rgb_colors"].detach().clone()}
    g = torch.Generator().manual_seed(1)
    dataset = [{"cam": cam, "im": torch.rand(3, H, W, generator=g).to(dev), "id": 0, "mask": None}]
    saved = topo4d_amd.rasterizer._save_sync_mode()
    try:
        topo4d_amd.set_sync_mode("auto")
        t4d_loop.optimise_dense_views(params, variables, dataset, opt, n_iters=3, explicit=True)      # first call checked: learns the arena
        torch.cuda.synchronize(dev)
        runs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            losses = t4d_loop.optimise_dense_views(params, variables, dataset, opt, n_iters=n_iters, explicit=True)
            torch.cuda.synchronize(dev)
            runs.append(1e3 * (time.perf_counter() - t0) / n_iters)
    finally:
        topo4d_amd.rasterizer._restore_sync_mode(saved)
    return {"workload": "texture loop (train.py:729-741), P=1000000, one 4096x3008 view per iteration: pins -> render -> photometric loss "
                        "(no affine) + 0.02 soft colour -> backward -> fused Adam; loop.optimise_dense_views, chained by hand",
            "ms_per_iteration": round(min(runs), 3), "ms_runs": [round(x, 3) for x in runs], "it_per_s": round(1e3 / min(runs), 1),
            "last_loss": round(float(losses[-1]), 6)}


def loss_probe(dev):
    """Row f1: the fused photometric loss (t4d_photometric_loss: per-camera affine + 0.8 L1 + 0.2 (1-SSIM), forward AND gradient,
    train.py:310,315) - kernel time by HIP events on the launch stream around back-to-back library calls, its own roofline.
    Algorithmic bytes per pixel and channel: read im + gt, write dL/dim = 12 B (36 B per pixel)."""
    import ctypes as C
    from topo4d_amd import _lib
    lib = _lib.load()

    def raw(V_, H_, W_, reps):
        g = torch.Generator().manual_seed(V_ + H_)
        a = torch.rand(V_, 3, H_, W_, generator=g).to(dev)
        b = torch.rand(V_, 3, H_, W_, generator=g).to(dev)
        cm, cc = (torch.randn(V_, 3, generator=g) * 0.1).to(dev), (torch.randn(V_, 3, generator=g) * 0.05).to(dev)
        l, d, dm, dcc = torch.empty(V_, device=dev), torch.empty_like(a), torch.empty_like(cm), torch.empty_like(cc)
        nb = lib.t4d_photometric_scratch_bytes(V_, H_, W_)
        sc = torch.empty(nb, dtype=torch.uint8, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        call = lambda: lib.t4d_photometric_loss(V_, H_, W_, p(a), p(b), p(cm), p(cc), None, p(l), p(d), p(dm), p(dcc), p(sc), nb, st)
        for _ in range(5):
            if call() != 0:
                raise RuntimeError(_lib.last_error())
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize(dev)
            t = 1e3 * e0.elapsed_time(e1) / reps
            best = t if best is None or t < best else best
        alg = V_ * 3 * H_ * W_ * 12
        return {"kernel_us": round(best, 1), "alg_bytes": alg, "achieved_GBs": round(alg / best / 1e3, 1),
                "frac": round(alg / best / 1e3 / PEAK_HBM_GBS, 4)}
    return {"workload": "t4d_photometric_loss: camera affine + 0.8 L1 + 0.2 (1 - SSIM 11x11), loss AND dL/dim, dL/dcam_m, dL/dcam_c in one call",
            "24x512x512": raw(24, 512, 512, 40), "1x512x375": raw(1, 512, 375, 100), "24x2048x2048": raw(24, 2048, 2048, 4),
            "roofline": {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "measured_limiter": "vector-ALU issue (154 multiply-adds per pixel and channel: two separable 11-tap passes over 4 + 3 maps)"},
            "timing": "HIP events on the launch stream around back-to-back calls, min of 5 runs (the kernels + the final-sum launch)"}


def bake_probe(dev, cpu=True, res=8192, n=1025):
    """BASELINE config 5: the 8192^2 texture bake (t4d_texture_bake behind topo4d_amd.texture.render_colors = the reference's
    face3d render_colors -> _render_colors_core, helpers.py:953-960, mesh_core.cpp:169-234) of a UV mesh of ~10^6 vertices /
    2.1 M triangles.  Algorithmic bytes: 16 B per texel (3 colour floats + the depth buffer) + the inputs read once."""
    from oracle import texture_oracle as TX
    from scaffold.scene import uv_mesh
    from topo4d_amd import texture
    verts, tris, colors = uv_mesh(n, res, res, seed=0)
    tris = np.sort(tris.view([("a", np.int32), ("b", np.int32), ("c", np.int32)]), order=["a"], axis=0).view(np.int32)   # mesh order
    v, t, c = (torch.as_tensor(x).to(dev) for x in (verts, tris, colors))
    img = texture.render_colors(v, t, c, res, res)          # warm-up + pair capacity
    torch.cuda.synchronize(dev)
    runs = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(5):
            img = texture.render_colors(v, t, c, res, res)
        torch.cuda.synchronize(dev)
        runs.append(1e3 * (time.perf_counter() - t0) / 5)
    ms = min(runs)
    alg = res * res * 16 + verts.nbytes + tris.nbytes + colors.nbytes
    out = {"workload": f"{res}x{res} texels, {int(tris.shape[0])} triangles, {int(verts.shape[0])} vertices, 3 channels; output + depth "
                       "buffer allocation and fill, binning and render included, inputs resident in HBM",
           "ms": round(ms, 3), "ms_runs": [round(x, 3) for x in runs], "texels_per_s": round(res * res / ms * 1e3, 1),
           "alg_bytes": int(alg),
           "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(alg / ms / 1e6 / PEAK_HBM_GBS, 4)}}
    if cpu:
        t0 = time.perf_counter()
        ref = TX.render_colors_cpu(verts, tris, colors, res, res)
        sec = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(res * res / sec, 1), "unit": "texels/s", "seconds": round(sec, 3), "cores": 1,
                               "kind": "reference" if TX.have_ref() else "port",
                               "sample": "the same full mesh, one run (the reference's code is single-threaded by construction)"}
        out["bit_identical_to_cpu"] = bool(np.array_equal(img.cpu().numpy(), ref))
        out["speedup"] = round(sec * 1e3 / ms, 1)
    return out


def full_iteration_probe(dev, reps=5):
    """The whole optimisation iteration around the rasterizer (train.py:661-700): activations -> render -> photometric loss ->
    backward -> Adam + region pins, on the fused pieces.  V = 1 (the reference's schedule: one camera per iteration, P = 8,280,
    512x375) eagerly and replayed from one HIP graph per camera; V = 24 (all cameras of a frame in one launch set, config-2
    scene, one Adam step per frame)."""
    from scaffold import scene
    from topo4d_amd import loop as t4d_loop, loss as t4d_loss, rasterize_views
    from topo4d_amd.boundary import params2rendervar_fused
    from topo4d_amd.optim import FusedAdamPins
    import topo4d_amd
    out = {}
    g = torch.Generator().manual_seed(0)

    def setup(n_lat, n_lon, H, W, capturable):
        p = scene.make_gaussians(n_lat, n_lon, opacity="A", seed=0)
        params = {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}
        P = params["means3D"].shape[0]
        opt = FusedAdamPins([{"params": [v], "name": k, "lr": 1e-5} for k, v in params.items()], eps=1e-15, capturable=capturable)
        opt.set_pin("means3D", torch.arange(0, P, 5), params["means3D"][::5].detach().clone())      # a "static region" (train.py:676-700)
        cams = scene.camera_rig(H, W, n_views=24, device=dev)
        gts = [torch.rand(3, H, W, generator=g).to(dev) for _ in range(24)]
        return params, opt, cams, gts

    saved = topo4d_amd.rasterizer._save_sync_mode()
    try:
        # ---- V = 1, eager (drop-in default sync mode) and graphed
        topo4d_amd.rasterizer._restore_sync_mode(("checked", False))
        params, opt, cams, gts = setup(69, 120, 512, 375, False)
        data = [{"cam": cams[i], "im": gts[i], "id": i} for i in range(24)]

        def it_eager(i):                                 # (what loop.optimise_views does per iteration: chained by hand)
            l, _, grads, _, _ = t4d_loop.explicit_iteration(params, data[i % 24])
            for k, gr in grads.items():
                params[k].grad = gr
            opt.step()
            opt.zero_grad(set_to_none=True)
        for i in range(48):
            it_eager(i)
        torch.cuda.synchronize(dev)
        runs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for i in range(300):
                it_eager(i)
            torch.cuda.synchronize(dev)
            runs.append(300 / (time.perf_counter() - t0))
        out["v1_eager_it_per_s"] = round(max(runs), 1)
        out["v1_eager_runs"] = [round(x, 1) for x in runs]
        gparams, gopt, cams, gts = setup(69, 120, 512, 375, True)
        gdata = [{"cam": cams[i], "im": gts[i], "id": i} for i in range(24)]
        gv = t4d_loop.GraphedViews(gparams, gdata, gopt)
        for i in range(48):
            gv.step(i % 24)
        torch.cuda.synchronize(dev)
        runs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for i in range(1000):
                gv.step(i % 24)
            torch.cuda.synchronize(dev)
            runs.append(1000 / (time.perf_counter() - t0))
        gv.check()
        out["v1_graphed_it_per_s"] = round(max(runs), 1)
        out["v1_graphed_runs"] = [round(x, 1) for x in runs]
        out["v1_graphed_us_per_iteration"] = round(1e6 / max(runs), 1)
        del gv
        # ---- V = 24: one launch set per frame
        topo4d_amd.set_sync_mode("checked")
        params, opt, cams, gts = setup(150, 200, 512, 512, False)
        gt = torch.stack(gts)

        frame = [{"cam": cams[i], "im": gts[i], "id": i} for i in range(24)]

        def it_frame():                                  # (loop.explicit_frame_iteration: one launch set per frame, chained by hand)
            _, _, grads, _ = t4d_loop.explicit_frame_iteration(params, frame, gt)
            for k, gr in grads.items():
                params[k].grad = gr
            opt.step()
            opt.zero_grad(set_to_none=True)
        it_frame()
        topo4d_amd.set_sync_mode("lazy")
        for _ in range(10):
            it_frame()
        torch.cuda.synchronize(dev)
        runs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(50):
                it_frame()
            torch.cuda.synchronize(dev)
            runs.append(1e3 * (time.perf_counter() - t0) / 50)
        out["v24_ms_per_frame_iteration"] = round(min(runs), 4)
        out["v24_runs_ms"] = [round(x, 4) for x in runs]
        out["v24_views_per_s"] = round(24e3 / min(runs), 1)
    finally:
        topo4d_amd.rasterizer._restore_sync_mode(saved)
    out["v1_path"] = "loop.explicit_iteration: the five library calls chained by hand, no autograd (eager: as loop.optimise_views; graphed: loop.GraphedViews)"
    out["workload"] = ("params2rendervar (fused activations) -> render -> fused photometric loss (L1 + SSIM) -> backward -> fused Adam + pins; "
                       "v1: P=8280, 512x375, one camera per iteration (train.py:661-700); v24: config-2 scene, 24 cameras per launch set, view-summed gradients, one Adam step per frame")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50; in --scaling strong: frame-steps of the whole job, default 64)")
    ap.add_argument("--prewarm-s", dest="prewarm_s", type=float, default=0.4,
                    help="seconds of untimed steps before the W warm-up steps (lets the GPU clocks ramp)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=["C2", "C4"])
    ap.add_argument("--opacity", default="A", choices=["A", "B"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--shard", default="frames", choices=["frames", "views"],
                    help="frames: rank r renders frames r, r+N, ... (24 views each); views: every rank steps through all frames and "
                         "renders views r::N of each (strong scaling of ONE frame: the split Topo4D's sequential frames allow)")
    ap.add_argument("--frames-per-launch", dest="frames_per_launch", type=int, default=0,
                    help="--shard views: independent frames whose views one launch set carries, each frame with its own Gaussians "
                         "(T4DProblem.views_per_param_set).  Default: N, the number of ranks - every launch set holds 24 views like the "
                         "one-GPU run's, --frames-in-flight of them in flight.  1: one frame at a time on one stream - the schedule of "
                         "Topo4D's real loop, whose frames are sequential (train.py:646)")
    ap.add_argument("--allreduce-grads", action="store_true", help="--shard views: sum the view-summed parameter gradients over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the scenario-B and single-view side measurements")
    ap.add_argument("--cpu-sample-views", type=int, default=0)
    ap.add_argument("--frames-in-flight", dest="in_flight", type=int, default=int(os.environ.get("T4D_BENCH_IN_FLIGHT", "0")),
                    help="independent frames in flight on as many HIP streams (1 = one frame after the other on one stream; "
                         "default: 3 for config 2, 2 for config 4)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 64 if args.scaling == "strong" else 50

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without a launcher: become `python -m torch.distributed.run ... bench.py <same flags>`
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 1000),
                                   os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists by design)")
    if os.environ.get("T4D_BENCH_SHARE_GPU") == "1":                   # dry run: several ranks on one GPU (never timed runs)
        local_rank = 0
    elif torch.cuda.device_count() < max(world, local_rank + 1):
        # every rank sees the same count and leaves before the rendezvous: no rank is left waiting in init_process_group
        raise SystemExit(f"bench.py --gpus {args.gpus}: {torch.cuda.device_count()} GPU(s) visible, one per rank needed "
                         f"(rank {rank}, local rank {local_rank}); nothing was run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # A process group exists whenever a launcher started us (torchrun sets RANK/WORLD_SIZE) - also for ONE rank, so that the
    # RCCL path (init, async all_gather on the step's stream, barrier with device_ids) can be exercised on a one-GPU box.
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("T4D_DIST_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm; gloo only for dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    # --scaling strong: the job is fixed; its frame-steps are rounded UP to a whole number per rank (64 for 1/2/4/8 ranks)
    strong_job = args.steps if args.scaling == "strong" else STRONG_JOB_FRAMES
    strong_steps_per_rank = -(-strong_job // world)

    import topo4d_amd

    if args.in_flight < 1:
        # config 2: 0.430 / 0.425 / 0.433 ms with 2 / 3 / 4 frames in flight; config 4: 4.06 / 4.10 ms with 2 / 3 (one: 4.20) - end of
        # round 3, same box; `sequential` in the JSON line is the same steps one frame at a time
        args.in_flight = 3 if args.config == "C2" else 2
    by_views = args.shard == "views"
    G = 1
    if by_views:
        args.scaling = "strong"
        shard = (rank, world)
        if world == 1 and os.environ.get("T4D_BENCH_VIEW_SHARD"):     # tests: ONE process renders rank r's shard of an N-way split ("r/N")
            shard = tuple(int(x) for x in os.environ["T4D_BENCH_VIEW_SHARD"].split("/"))
        # The 64 frames of config 3's synthetic sequence are independent units: a rank's launch set carries its 24 / N cameras of
        # N consecutive frames (24 views, like the one-GPU run's launch sets), several such sets in flight.  --frames-per-launch 1 is
        # the schedule Topo4D's real loop allows (its frames are sequential, train.py:646): one frame, one stream.
        G = args.frames_per_launch if args.frames_per_launch > 0 else shard[1]
        if G == 1:
            args.in_flight = 1
        wl = Workload(args.config, args.opacity, dev, rank, world, args.in_flight, gather=dist is not None, view_shard=shard,
                      allreduce_grads=args.allreduce_grads, frames_per_launch=G)
    else:
        wl = Workload(args.config, args.opacity, dev, rank, world, args.in_flight, gather=dist is not None)
    cfg, H, W, V, P = wl.cfg, wl.H, wl.W, wl.V, wl.P
    # a view-sharded step is one launch set = G whole frames (--steps counts FRAMES there, rounded up to whole launch sets)
    my_steps = -(-args.steps // G) if by_views else (strong_steps_per_rank if args.scaling == "strong" else args.steps)
    # views per step over ALL ranks: a view-sharded step is G whole frames; a frame-sharded step is one frame per rank
    views_per_step_job = wl.full_views * G if by_views else V * world

    def barrier():
        wl.drain()
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    def all_reduce_max(x):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    wl.learn_capacity()
    dt, t_enqueue, repeats = timed_run(wl, my_steps, args.warmup, args.prewarm_s, barrier, all_reduce_max)
    sequential = None
    if wl.F > 1:                     # the same steps, one frame after the other on one stream: what the overlap is worth
        wl.sequential = True
        dts, _, sts_seq = timed_run(wl, my_steps, args.warmup, 0.05, barrier, all_reduce_max)
        wl.sequential = False
        sequential = {"ms_per_step": round(1e3 * dts / my_steps, 4), "value": round(views_per_step_job * my_steps / dts, 2), "unit": "views/s",
                      "ms_per_step_min_median_max": sts_seq["ms_per_step"], "regions": sts_seq["regions"], "steps_per_region": sts_seq["steps_per_region"],
                      "note": "frames_in_flight = 1: the same steps one after the other on one stream"}
    # The OTHER scaling mode in the same line: the driver's one command per N must yield both the weak figure (`value`) and the
    # number north_star asks for - the fixed 64-frame job of BASELINE config 3 split over the ranks (strong scaling).
    other = None
    if args.config == "C2" and not by_views:
        o_steps = 20 if args.scaling == "strong" else max(1, min(strong_steps_per_rank, 512))
        dto, _, sto = timed_run(wl, o_steps, min(args.warmup, 2), 0.05, barrier, all_reduce_max, regions=3,
                                min_region_s=(0.0 if args.scaling == "weak" else MIN_REGION_S))
        if args.scaling == "weak":
            other = {"scaling": "strong", "job_frame_steps": o_steps * world, "steps_per_rank": o_steps,
                     "ms_job": round(1e3 * dto, 4), "value": round(V * o_steps * world / dto, 2), "unit": "views/s", "regions": sto["regions"],
                     "note": f"fixed job: the {STRONG_JOB_FRAMES}-frame sequence of config 3 (rounded up to a whole number of frames "
                             "per rank), sharded by frame; strong-scaling speed-up at N GPUs = this value at N / this value at 1"}
        else:
            other = {"scaling": "weak", "steps_per_rank": o_steps, "ms_per_step": round(1e3 * dto / o_steps, 4),
                     "value": round(V * o_steps * world / dto, 2), "unit": "views/s"}
    sts = wl.statuses()
    if any(x.overflow for x in sts):
        raise SystemExit("pair arena overflowed during the timed region: result invalid")
    total_pairs_all = sts[0].total_pairs
    # T4D_BENCH_DUMP_LOSSES=k (tests): the gathered per-view loss vectors of this rank's first k steps go into the JSON line
    last_losses = grad_sums = None
    if os.environ.get("T4D_BENCH_DUMP_LOSSES"):
        last_losses, grad_sums = [], []
        for i in range(int(os.environ["T4D_BENCH_DUMP_LOSSES"])):
            o, gs = wl.step(i)
            wl.drain()
            torch.cuda.synchronize(dev)
            last_losses.append(o.detach().float().cpu().tolist())
            # one float64 checksum per view of THIS rank over all its gradient tensors (tests: a view's gradients do not depend
            # on which other views share its launch)
            grad_sums.append(sum(t.double().reshape(t.shape[0], -1).sum(1) for t in gs[0].values() if t is not None).cpu().tolist())
    barrier()

    # ---- per-kernel durations with HIP events (same steps again; keeps `value` free of event overhead) ----
    # Every rank replays the steps (a step contains the loss all_gather when N > 1, so all ranks must take part);
    # only rank 0 records events.
    prof, tp = kernel_profile(wl, my_steps)
    roofline = None
    sh_bytes = 0 if cfg["sh_degree"] is None else 3 * (cfg["sh_degree"] + 1) ** 2 * 4
    VL = V * G                       # views per launch set
    R_view = total_pairs_all / VL
    per_kernel, total_bytes = algorithmic_bytes(P, R_view, H * W, sh_bytes)
    if rank == 0:
        kernels = {}
        for name, (ms, n) in prof.items():
            if n:
                kernels[name] = {"avg_us": round(1e3 * ms / n, 2), "launches": n,
                                 "alg_GBs": round(per_kernel[name] * VL / (1e-3 * ms / n) / 1e9, 1)}
        dom = max(kernels, key=lambda k: kernels[k]["avg_us"])
        ach = per_kernel[dom] * VL / (kernels[dom]["avg_us"] * 1e-6) / 1e9
        traffic = load_profile_json("traffic.json", args.config, dom)
        # vector-ALU counters of the committed rocprofv3 --pmc passes (tools/prof.sh -> profiles/valu.json): what actually limits
        # the render kernels (HISTORY.md section 5).  busy = SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (SIMDs * kernel cycles).
        valu = load_profile_json("valu.json", args.config, dom)
        # What bounds the dominant kernel, FROM THE COUNTERS: the vector ALUs are "busy" SQ_ACTIVE_INST_VALU x 4 / (SIMDs x kernel
        # cycles) of the time; above 0.6 the kernel is issue-bound and the HBM fraction is low by construction.  `achieved`/`peak`/
        # `frac` stay the HBM figures the contract asks for; `issue` and `useful_lane_fraction` sit next to them.
        busy = valu.get("valu_busy") if valu else None
        lanes_key = args.config + ("" if args.opacity == "A" else "_" + args.opacity)
        lanes = load_profile_json("lanes.json", lanes_key, "bwd" if dom == "k_render_bwd" else "fwd")
        issue = None
        if valu:
            issue = {"valu_busy": busy, "achieved_inst_per_cycle_per_simd": round(valu["SQ_INSTS_VALU"] / (1024.0 * valu["kernel_cycles"]), 4),
                     "peak_inst_per_cycle_per_simd": 0.45, "note": "peak = measured v_fma_f32 rate (tools/micro/valu_issue.hip); DPP adds run at 0.23"}
        roofline = {"bound": ("valu" if busy is not None and busy > 0.6 else "hbm"), "kernel": dom, "achieved": round(ach, 1),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": int(per_kernel[dom] * VL), "avg_us": kernels[dom]["avg_us"],
                    "pairs_per_view": int(R_view), "ms_per_step_profiled": round(1e3 * tp / my_steps, 4),
                    "pipeline_alg_bytes_per_view": int(total_bytes), "kernels": kernels,
                    "valu": valu, "issue": issue,
                    "useful_lane_fraction": (lanes or {}).get("useful_lane_fraction"), "row_balance": (lanes or {}).get("row_balance"),
                    "counters": counters_provenance(args.config, lanes_key),
                    "measured_limiter": ("vector-ALU issue" if busy is not None and busy > 0.6 else None)}
    if dist is not None:
        barrier()

    # ---- the view-sharded split, executed (all ranks): rank r renders views r::N of EVERY frame ----
    # `value`: the 64 independent frames of config 3, N of them per launch set (24 views per set at every N) and --frames-in-flight
    # sets in flight; `sequential_frames`: one frame at a time on one stream, the schedule Topo4D's real loop allows (train.py:646).
    view_sharded = None
    if not by_views and not args.no_extras and args.config == "C2":
        def run_view_sharded(frames_per_launch, in_flight):
            wv = Workload(args.config, args.opacity, dev, rank, world, in_flight, gather=dist is not None, view_shard=(rank, world),
                          resident=4, frames_per_launch=frames_per_launch)
            wv.learn_capacity()

            def barrier_v():
                wv.drain()
                if dist is not None:
                    dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()
                torch.cuda.synchronize(dev)
            n_sets = max(2, 48 // frames_per_launch)
            dtv, _, stv = timed_run(wv, n_sets, 3, 0.05, barrier_v, all_reduce_max, regions=5, min_region_s=0.05)
            per_frame = {k: round(x / frames_per_launch, 4) for k, x in stv["ms_per_step"].items()}
            r = {"views_per_rank_and_frame": wv.V, "frames_per_launch": frames_per_launch, "frames_in_flight_sets": wv.F, "ms_per_frame": per_frame,
                 "value": round(wv.full_views * frames_per_launch * n_sets / dtv, 2), "unit": "views/s"}
            del wv
            return r
        vs_batched = run_view_sharded(world, args.in_flight)
        vs_seq = run_view_sharded(1, 1)
        view_sharded = dict(vs_batched)
        view_sharded.update({
            "parallelism": f"view-sharded x{world}: rank r renders views r::{world} of every frame ({vs_batched['views_per_rank_and_frame']} views per rank and "
                           f"frame); one launch set carries {world} consecutive frames of the 64-frame sequence (each with its own Gaussians: "
                           "T4DProblem.views_per_param_set), loss all_gather per launch set",
            "scaling": "strong", "views_per_rank": vs_batched["views_per_rank_and_frame"], "sequential_frames": vs_seq,
            "note": "strong-scaling speed-up at N GPUs = this value at N / this value at 1 (the N = 1 line's `forecast` predicts it); "
                    "`sequential_frames` = one frame at a time on one stream, what Topo4D's real loop (sequential frames, train.py:646) allows"})

    # ---- side measurements on one GPU ----
    scenario_b = single_view = drop_in = small_v = forecast = c4 = dense_1m = full_iteration = loss_k = bake = None
    if world == 1 and not args.no_extras:
        def guarded(fn, *a):
            try:                        # side measurements must never take the headline number down with them
                return fn(*a)
            except Exception as e:
                return {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                topo4d_amd.set_sync_mode("lazy")
                torch.cuda.synchronize(dev)
        if args.opacity == "A":
            def scen_b():
                wb = Workload(args.config, "B", dev, in_flight=args.in_flight)
                wb.learn_capacity()
                dtb, _, stb_ = timed_run(wb, my_steps, args.warmup, 0.1, lambda: torch.cuda.synchronize(dev), lambda x: x)
                stb = wb.statuses()
                return {"value": round(V * my_steps / dtb, 2), "unit": "views/s", "ms_per_step": round(1e3 * dtb / my_steps, 4),
                        "ms_per_step_min_median_max": stb_["ms_per_step"], "steps": my_steps, "pairs_per_view": int(stb[0].total_pairs / V),
                        "overflow": bool(any(x.overflow for x in stb)),
                        "workload": "same as config.workload with opacity scenario B: uniform(0.05, 0.95)"}
            scenario_b = guarded(scen_b)
        single_view = guarded(single_view_probe, dev)
        drop_in = guarded(drop_in_probe, dev)
        if args.config == "C2":
            small_v = guarded(small_v_probe, dev)
            forecast = guarded(forecast_probe, dev, args.in_flight)
            full_iteration = guarded(full_iteration_probe, dev)
            del wl.batches, wl.rv_frames
            torch.cuda.empty_cache()
            c4 = guarded(c4_probe, dev)
            torch.cuda.empty_cache()
            dense_1m = guarded(dense_1m_probe, dev)
            torch.cuda.empty_cache()
            loss_k = guarded(loss_probe, dev)
            torch.cuda.empty_cache()
            bake = guarded(bake_probe, dev, not args.no_cpu_baseline)
            torch.cuda.empty_cache()

    if rank == 0:
        views_total = views_per_step_job * my_steps
        value = views_total / dt
        if roofline is not None:
            roofline["pipeline_frac_of_peak"] = round(value / world * total_bytes / 1e9 / PEAK_HBM_GBS, 4)
            roofline["pipeline_frac_of_achievable"] = round(value / world * total_bytes / 1e9 / ACHIEVABLE_HBM_GBS, 4)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            n_s = args.cpu_sample_views or (24 if args.config == "C2" else 4)
            try:
                cpu = cpu_baseline(cfg, n_s, reps=5)
            except Exception as e:  # the baseline must never take the GPU number down with it
                cpu = {"value": None, "unit": "views/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        out = {
            "metric": "rasterizer fwd+bwd views/sec", "value": round(value, 2), "unit": "views/s",
            "n_gpus": world, "steps": (my_steps * G if by_views else (my_steps * world if args.scaling == "strong" else args.steps)), "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / (my_steps * G), 4), "higher_is_better": True, "scaling": args.scaling, "repeats": repeats,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {V} views x {H}x{W}, P={P} vertex-bound Gaussians, "
                                   f"{'SH degree %d' % cfg['sh_degree'] if cfg['sh_degree'] is not None else 'precomputed RGB'}, "
                                   f"opacity scenario {args.opacity}, forward+backward, per-view gradients",
                       "views_per_step_per_gpu": V, "frames": wl.n_frames, "steps_per_rank": my_steps,
                       "parallelism": (f"view-sharded x{world}: rank r renders views r::{world} of every frame" +
                                       (f", {G} frames per launch set" if G > 1 else ", one frame at a time (the real loop's frames are sequential)") +
                                       (", parameter gradients all-reduced" if args.allreduce_grads else "")) if by_views else f"frame-sharded x{world}",
                       "frames_per_launch": G, "ms_per_launch_set": round(1e3 * dt / my_steps, 4),
                       "sync_mode": "lazy (capacity learned by checked warm-up)",
                       "host_enqueue_ms_per_step": round(1e3 * t_enqueue / my_steps, 4), "frames_in_flight": wl.F},
            "roofline": roofline, "cpu_baseline": cpu, "scenario_b": scenario_b, "single_view": single_view, "small_v": small_v,
            "forecast": forecast, "view_sharded": view_sharded, "drop_in": drop_in, "full_iteration": full_iteration, "c4": c4,
            "dense_1m": dense_1m, "loss": loss_k, "bake_8192": bake, "sequential": sequential,
            ("weak" if (args.scaling == "strong" and not by_views) else "strong"): other,
            "dist_backend": (dist.get_backend() if dist is not None else None),
        }
        if last_losses is not None:
            out["gathered_losses_first_steps"] = last_losses       # rank-major: out.view(world, -1).t() is view order in --shard views
            out["grad_checksums_first_steps_rank0"] = grad_sums
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
