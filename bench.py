#!/usr/bin/env python
"""
bench.py — rasterizer forward+backward views/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C4] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: forward + backward of the 24 views of one synthetic frame
(BASELINE config 2: 24 x 512x512, P = 30,000 vertex-bound Gaussians, precomputed RGB) through the C ABI, with
all inputs already resident in HBM and seeded dL/dcolor supplied.  One "view" = one
(Settings, params) -> colour/depth/alpha -> per-view dL/dparams round trip.

Multi-GPU (BASELINE config 3): the 64-frame sequence is sharded by frame — rank r renders frames r, r+N, ...;
per-GPU work per step is fixed (24 views) => "scaling": "weak".  The only collective is an RCCL all_gather of the
per-view scalar losses (24 floats per rank per step).  K*N = 64 steps is the strong-scaling run of config 3.

The JSON line carries `roofline` (dominant kernel, HIP-event duration, algorithmic bytes per launch) and
`cpu_baseline` (oracle/raster_oracle.c, OpenMP, bounded sample) — see DESIGN.md §Measurement.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    total = P * (352 + 3 * S) + R * 120 + HW * 60
    return per_kernel, total


def cpu_baseline(cfg, n_sample_views):
    """C oracle (oracle/raster_oracle.c), OpenMP over all host cores, fwd+bwd on a bounded sample of the workload."""
    from oracle import c_oracle as CO
    from topo4d_amd import boundary, scene
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

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(one, range(n_sample_views)))
    dt = time.perf_counter() - t0
    out = {"value": round(n_sample_views / dt, 3), "unit": "views/s", "cores": cores, "kind": "port",
           "sample": f"{n_sample_views} of the {cfg['n_views']} views of the same scene, fwd+bwd, oracle/raster_oracle.c -O3 "
                     f"-fopenmp, {workers} views at a time x {per} OpenMP threads, {dt:.1f}s wall"}
    # the same code on ONE host thread, one view (SURVEY.md 8d asks for both figures)
    try:
        gomp.omp_set_num_threads(1)
        t1 = time.perf_counter()
        r = CO.OracleRender(cams[0], rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"],
                            rv.get("colors_precomp"), rv.get("shs"))
        r.backward(dc[0])
        d1 = time.perf_counter() - t1
        gomp.omp_set_num_threads(os.cpu_count())
        out["one_thread"] = {"value": round(1.0 / d1, 4), "unit": "views/s", "cores": 1, "sample": f"1 view, {d1:.1f}s wall"}
    except Exception as e:                       # libgomp not loadable: report the multi-threaded figure only
        out["one_thread"] = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--prewarm-s", dest="prewarm_s", type=float, default=0.4,
                    help="seconds of untimed steps before the W warm-up steps (lets the GPU clocks ramp)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=["C2", "C4"])
    ap.add_argument("--opacity", default="A", choices=["A", "B"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-views", type=int, default=0)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("T4D_BENCH_STREAMS", "1")),
                    help="split the views of a step over this many HIP streams (independent views overlap their kernel tails)")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("T4D_DIST_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm; gloo only for dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import topo4d_amd
    from topo4d_amd import ViewBatch, _lib, boundary, dist as t4d_dist, pack_views, scene
    from topo4d_amd.rasterizer import view_dot

    cfg = dict(scene.CONFIGS[args.config])
    H, W, V = cfg["H"], cfg["W"], cfg["n_views"]
    params = scene.make_gaussians(cfg["n_lat"], cfg["n_lon"], opacity=args.opacity, sh_degree=cfg["sh_degree"], seed=0)
    P = params["means3D"].shape[0]
    base_means = params["means3D"].clone()
    cams = scene.camera_rig(H, W, n_views=V, device=dev, true_campos=cfg["sh_degree"] is not None)
    if cfg["sh_degree"] is not None:
        cams = [c._replace(sh_degree=cfg["sh_degree"]) for c in cams]
    views = pack_views(cams, dev)
    dc, _, _ = scene.output_cotangents(V, H, W, seed=0)
    dc = dc.to(dev)
    dc_flat = dc.flatten(1)

    # per-frame Gaussians of the synthetic 64-frame sequence (config 3); all resident in HBM before timing
    n_frames = 64
    my_frames = t4d_dist.shard_units(n_frames, rank, world)
    rv_frames = []
    for t in my_frames[: max(1, min(len(my_frames), 8))]:
        p = dict(params)
        p["means3D"] = scene.frame_displacement(base_means, t, n_frames)
        rv = {k: v.detach().to(dev) for k, v in boundary.params2rendervar(p).items()}
        if cfg["sh_degree"] is not None:
            rv["shs"] = params["shs"].to(dev)
            rv.pop("colors_precomp")
        rv_frames.append(rv)

    S = max(1, min(args.streams, V))
    bounds = [(V * k) // S for k in range(S + 1)]
    batches = [ViewBatch(views[bounds[k]:bounds[k + 1]].contiguous(), H, W, 1.0, cfg["sh_degree"] or 0) for k in range(S)]
    dcs = [dc[bounds[k]:bounds[k + 1]].contiguous() for k in range(S)]
    batch = batches[0]
    losses = torch.zeros(V, device=dev)
    # multi-GPU: the loss all_gather of step i overlaps with step i+1 (double-buffered, waited on two steps later)
    loss_bufs = [torch.zeros(V, device=dev), torch.zeros(V, device=dev)]
    gath_bufs = [torch.zeros(V * world, device=dev), torch.zeros(V * world, device=dev)]
    pending = [None, None]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [None]

    def step(i):
        nonlocal losses
        rv = rv_frames[i % len(rv_frames)]
        g = []
        if world > 1:
            if pending[i % 2] is not None:
                pending[i % 2].wait()
                pending[i % 2] = None
            losses = loss_bufs[i % 2]
        if S > 1:
            main = torch.cuda.current_stream(dev)
            for st in streams:
                st.wait_stream(main)
        for k in range(S):
            with torch.cuda.stream(streams[k]) if S > 1 else contextlib.nullcontext():
                color, radii, depth, alpha = batches[k].forward(rv["means3D"], rv["opacities"], rv["scales"], rv["rotations"],
                                                                rv.get("colors_precomp"), rv.get("shs"))
                g.append(batches[k].backward(dcs[k]))
                # per-view scalar loss term <colour, dL/dcolour>, one fused pass
                view_dot(color, dcs[k], out=losses[bounds[k]:bounds[k + 1]])
        if S > 1:
            for st in streams:
                main.wait_stream(st)
        if world > 1:
            out, work = t4d_dist.gather_losses_async(losses, gath_bufs[i % 2])
            pending[i % 2] = work
            return out, g
        return losses, g

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def barrier():
        drain()
        if world > 1:
            import torch.distributed as dist
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    # warm-up: first call is "checked" (learns the pair-arena capacity), the rest of the run is lazy (no host sync)
    topo4d_amd.set_sync_mode("checked")
    step(0)
    st0 = batch.fetch_status()
    for i in range(len(rv_frames)):
        step(i)
    topo4d_amd.set_sync_mode("lazy")
    # A GPU that has just been idle (fresh box, or a profiler run before this one) needs tens of milliseconds of work
    # before its clocks settle: a cold 50-step run measured 1.23 ms/step against 0.62 warm.  Untimed, like the W steps.
    torch.cuda.synchronize(dev)
    t_pre = time.perf_counter()
    for i in range(8):
        step(i)
    torch.cuda.synchronize(dev)
    est = torch.tensor([(time.perf_counter() - t_pre) / 8], device=dev, dtype=torch.float64)
    if world > 1:                                   # every rank must run the same number of steps (a step holds a collective)
        import torch.distributed as dist
        dist.all_reduce(est, op=dist.ReduceOp.MAX)
    n_pre = int(min(4000, max(0.0, args.prewarm_s) / max(float(est.item()), 1e-5)))
    for i in range(n_pre):
        step(i)
    torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i)

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_enqueue = time.perf_counter() - t0       # host time to enqueue K steps (GPU still running)
    barrier()
    dt = time.perf_counter() - t0
    sts = [b.fetch_status() for b in batches]
    if any(x.overflow for x in sts):
        raise SystemExit("pair arena overflowed during the timed region: result invalid")
    st = sts[0]
    total_pairs_all = sum(x.total_pairs for x in sts)

    t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())

    # ---- per-kernel durations with HIP events (same steps again; keeps `value` free of event overhead) ----
    # Every rank replays the steps (a step contains the loss all_gather when N > 1, so all ranks must take part);
    # only rank 0 records events.
    roofline = None
    kernels = {}
    torch.cuda.synchronize(dev)
    if rank == 0:
        _lib.profile_begin()
    tp0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize(dev)
    tp = time.perf_counter() - tp0
    if rank == 0:
        prof = _lib.profile_end()
        R_view = total_pairs_all / V
        sh_bytes = 0 if cfg["sh_degree"] is None else 3 * (cfg["sh_degree"] + 1) ** 2 * 4
        per_kernel, total_bytes = algorithmic_bytes(P, R_view, H * W, sh_bytes)
        for name, (ms, n) in prof.items():
            if n:
                kernels[name] = {"avg_us": round(1e3 * ms / n, 2), "launches": n,
                                 "alg_GBs": round(per_kernel[name] * V / (1e-3 * ms / n) / 1e9, 1)}
        dom = max(kernels, key=lambda k: kernels[k]["avg_us"])
        ach = per_kernel[dom] * V / (kernels[dom]["avg_us"] * 1e-6) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(args.config, {}).get(dom)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": int(per_kernel[dom] * V), "avg_us": kernels[dom]["avg_us"],
                    "pairs_per_view": int(R_view), "ms_per_step_profiled": round(1e3 * tp / args.steps, 4),
                    "pipeline_alg_bytes_per_view": int(total_bytes), "kernels": kernels}

    if world > 1:
        barrier()

    if rank == 0:
        views_total = V * args.steps * world
        value = views_total / dt
        _, total_bytes = algorithmic_bytes(P, total_pairs_all / V, H * W,
                                           0 if cfg["sh_degree"] is None else 3 * (cfg["sh_degree"] + 1) ** 2 * 4)
        if roofline is not None:
            roofline["pipeline_frac_of_peak"] = round(value / world * total_bytes / 1e9 / PEAK_HBM_GBS, 4)
            roofline["pipeline_frac_of_achievable"] = round(value / world * total_bytes / 1e9 / ACHIEVABLE_HBM_GBS, 4)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            n_s = args.cpu_sample_views or (24 if args.config == "C2" else 4)
            try:
                cpu = cpu_baseline(cfg, n_s)
            except Exception as e:  # the baseline must never take the GPU number down with it
                cpu = {"value": None, "unit": "views/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        out = {
            "metric": "rasterizer fwd+bwd views/sec", "value": round(value, 2), "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {V} views x {H}x{W}, P={P} vertex-bound Gaussians, "
                                   f"{'SH degree %d' % cfg['sh_degree'] if cfg['sh_degree'] is not None else 'precomputed RGB'}, "
                                   f"opacity scenario {args.opacity}, forward+backward, per-view gradients",
                       "views_per_step_per_gpu": V, "frames": n_frames, "parallelism": f"frame-sharded x{world}",
                       "sync_mode": "lazy (capacity learned by checked warm-up)",
                       "host_enqueue_ms_per_step": round(1e3 * t_enqueue / args.steps, 4), "hip_streams": S},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
