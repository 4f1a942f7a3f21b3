#!/usr/bin/env python
"""Small-launch regime (1-6 views per call) on the GPU box: kernel times (HIP events) and wall time per forward + backward call
of the Topo4D-sized scene and of the config-2 scene, under the switches that select the render builds:
    python tools/small_launch.py [topo4d|c2] [views ...]
Every line is bench.render_probe() (min over 5 runs of 200 calls); the environment variants are run in sub-processes."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [("default", {}),
            ("forward: throughput build", {"T4D_FWD_LATENCY_TILES": "0"}),
            ("forward: latency build without the fused sort", {"T4D_NO_FUSED_SORT": "1"}),
            ("backward: whole tiles", {"T4D_NO_SEGMENTS": "1"})]
if os.environ.get("T4D_SMALL_VARIANTS"):
    VARIANTS = [v for i, v in enumerate(VARIANTS) if str(i) in os.environ["T4D_SMALL_VARIANTS"].split(",")]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch, bench
    shape, views = sys.argv[2], [int(x) for x in sys.argv[3:]]
    for v in views:
        r = bench.render_probe(torch.device("cuda"), shape, v, reps=3, iters=150)
        k = r["kernels_us"]
        print(json.dumps({"shape": shape, "V": v, "gpu_us": r["gpu_us_per_call"], "wall_us": r["wall_us_per_call"],
                          "fwd": k.get("k_render_fwd"), "bwd": k.get("k_render_bwd"), "sort": k.get("k_sort_tiles"),
                          "pre": k.get("k_preprocess"), "scan": k.get("k_scan_tiles"), "scatter": k.get("k_scatter"),
                          "pre_bwd": k.get("k_preprocess_bwd"), "longest": r["longest_tile_list"]}), flush=True)
    sys.exit(0)

shapes = [sys.argv[1]] if len(sys.argv) > 1 else ["topo4d", "c2"]
views = sys.argv[2:] or ["1", "2", "3", "4", "6", "8"]
for shape in shapes:
    for name, env in VARIANTS:
        print(f"== {shape}: {name} {env}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", shape] + views, env=dict(os.environ, **env), cwd=ROOT)
