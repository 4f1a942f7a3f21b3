#!/usr/bin/env python
"""Config 4's forward measured 1,062 us on one box / process and 1,206 us on another for identical work.
Which buffer's PLACEMENT does that?  Runs `bench.kernel_profile` of config 4 in fresh processes with a pad tensor in front of
everything (moves every later allocation).  Prints fwd / bwd kernel times and the state buffer's address mod 16 MiB.
Round 5 (profiles/r05_c4_placement_probe.txt; a build with two more knobs - the three output tensors carved from ONE allocation
256 B ... 1 MiB apart, the state's final_T / n_contrib planes shifted by as much): one box shows exactly two levels, 1,175 us
(three fresh default processes, pads of 4 KiB and 1 MiB) and 1,206-1,211 us (a 5 MiB pad and EVERY variant of the two knobs,
whatever the skew and whatever the state's address bits) - so it is not a stride between planes that a pad could break; it
follows which hipMalloc blocks back the buffers.  Boxes differ by more than that (the driver's r04 box: 1,062 us).
    python tools/placement_probe.py [C4]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
import bench
dev = torch.device("cuda")
pad = int(os.environ.get("PROBE_PAD", "0"))
hold = torch.empty(max(pad, 1), dtype=torch.uint8, device=dev)
wl = bench.Workload(sys.argv[1], "A", dev, in_flight=1)
wl.learn_capacity()
for i in range(3): wl.step(i)
wl.drain(); torch.cuda.synchronize()
prof, _ = bench.kernel_profile(wl, 6)
k = {n: round(1e3 * ms / c, 1) for n, (ms, c) in prof.items() if c}
b = wl.batches[0]
print(json.dumps({"fwd": k["k_render_fwd"], "bwd": k["k_render_bwd"], "state_mod_16M": b.state.data_ptr() %% (1 << 24)}))
''' % ROOT
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
cases = [{}] * 3 + [{"PROBE_PAD": str(p)} for p in (4096, (1 << 20) + 4096, (5 << 20) + (64 << 10), (37 << 20) + 4096)]
for env in cases:
    r = subprocess.run([sys.executable, "-c", CHILD, cfg], env=dict(os.environ, **env), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(json.dumps(env), line[-1] if line else ("FAILED " + r.stderr[-300:]), flush=True)
