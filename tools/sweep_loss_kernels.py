#!/usr/bin/env python
"""Runs ON THE GPU BOX: times t4d_photometric_loss under the strip kernel's run-time switches (T4D_PH_THREADS = threads per
strip, T4D_PH_ROWS = rows per segment) and checks every variant's loss and gradient against the default's.  One line per
(variant, shape).  Arguments: variants as THREADS=192,ROWS=96 ...; none = the full sweep.  With T4D_LIB=<experiment build>
(tools/ab_build.sh) the same for that build."""
import ctypes as C, itertools, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo4d_amd import _lib
lib = _lib.load()
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def run(V, H, W, reps, env):
    for k in ("T4D_PH_THREADS", "T4D_PH_ROWS", "T4D_PH_TILE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    g = torch.Generator().manual_seed(V * H + W)
    a = torch.rand(V, 3, H, W, generator=g).cuda()
    b = (a.cpu() + torch.randn(V, 3, H, W, generator=g) * 0.1).clamp(0, 1).cuda()
    cm = (torch.randn(V, 3, generator=g) * 0.1).cuda(); cc = (torch.randn(V, 3, generator=g) * 0.05).cuda()
    l = torch.empty(V, device="cuda"); d = torch.zeros_like(a)
    dm = torch.empty(V, 3, device="cuda"); dc = torch.empty(V, 3, device="cuda")
    nb = lib.t4d_photometric_scratch_bytes(V, H, W); sc = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.t4d_photometric_loss(V, H, W, p(a), p(b), p(cm), p(cc), None, p(l), p(d), p(dm), p(dc), p(sc), nb, st)
    for _ in range(3): assert call() == 0
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best, l.clone(), d.clone(), dm.clone(), dc.clone()


shapes = [(24, 512, 512, 40), (1, 512, 375, 100), (3, 512, 375, 100), (24, 2048, 2048, 4)]
variants = [{}]
if len(sys.argv) > 1:
    variants += [{"T4D_PH_" + kv.split("=")[0]: kv.split("=")[1] for kv in a.split(",")} for a in sys.argv[1:]]
else:
    variants += [{"T4D_PH_THREADS": th} for th in ("64", "128", "192", "256")]
    variants += [{"T4D_PH_ROWS": r} for r in ("32", "64", "96", "128", "256")]
ref = {}
for env in variants:
    for (V, H, W, reps) in shapes:
        t, l, d, dm, dc = run(V, H, W, reps, env)
        key = (V, H, W)
        if not env:
            ref[key] = (l, d, dm, dc)
        rl, rd, rdm, rdc = ref[key]
        print("%-58s %2dx%4dx%4d  %8.1f us   dloss %.1e  dgrad %.1e (of max)  dcam %.1e" % (
            " ".join("%s=%s" % (k[7:], v) for k, v in env.items()) or "default", V, H, W, t, (l - rl).abs().max().item(),
            ((d - rd).abs().max() / rd.abs().max()).item(),
            max(((dm - rdm).abs().max() / rdm.abs().max()).item(), ((dc - rdc).abs().max() / rdc.abs().max()).item())), flush=True)
