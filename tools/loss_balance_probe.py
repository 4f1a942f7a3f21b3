#!/usr/bin/env python
"""Runs ON THE GPU BOX: the strip kernel's time for V views of 512 x 512 (36 V workgroups of three waves on 256 CUs) - how much of
a 24-view launch (864 workgroups: four on 96 CUs, three on 160) is imbalance."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo4d_amd import _lib
lib = _lib.load()
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
for V in (14, 21, 24, 28, 32, 35, 42):
    H = W = 512
    a = torch.rand(V, 3, H, W, device="cuda"); b = torch.rand(V, 3, H, W, device="cuda")
    l = torch.empty(V, device="cuda"); d = torch.empty_like(a)
    nb = lib.t4d_photometric_scratch_bytes(V, H, W); sc = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.t4d_photometric_loss(V, H, W, p(a), p(b), None, None, None, p(l), p(d), None, None, p(sc), nb, st)
    for _ in range(5): call()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(40): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / 40)
    print("V = %2d: %4d workgroups = %.2f per CU   %.1f us   %.2f us per view" % (V, 36 * V, 36 * V / 256, best, best / V), flush=True)
