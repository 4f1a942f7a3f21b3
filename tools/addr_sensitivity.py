#!/usr/bin/env python
"""Does k_render_fwd's duration at config 4 depend on WHERE its buffers sit?  (The same build measures 1.11 or 1.22 ms from one
process to the next, stable within a process.)  Shifts every later allocation by holding a pad tensor of varying size, rebuilds
the workload, and prints the per-kernel HIP-event times.   usage: python tools/addr_sensitivity.py [C4|C2]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
dev = torch.device("cuda")
for pad in [int(a) for a in sys.argv[2:]] or (0, 4096, 64 << 10, (1 << 20) + 4096, (2 << 20) + (64 << 10), (16 << 20) + (1 << 20)):
    torch.cuda.empty_cache()
    hold = torch.empty(max(pad, 1), dtype=torch.uint8, device=dev)
    wl = bench.Workload(cfg, "A", dev, in_flight=1)
    wl.learn_capacity()
    for i in range(3): wl.step(i)
    wl.drain(); torch.cuda.synchronize()
    prof, _ = bench.kernel_profile(wl, 6)
    k = {n: 1e3 * ms / c for n, (ms, c) in prof.items() if c}
    print("pad %9d  fwd %7.1f  bwd %7.1f  preprocess %6.1f  pre_bwd %6.1f" % (pad, k["k_render_fwd"], k["k_render_bwd"], k["k_preprocess"], k["k_preprocess_bwd"]), flush=True)
    del wl, hold
