#!/usr/bin/env python
"""Counting build of the render kernels (GPU box): what their visit loops do on BASELINE config 2 / 4, per 24-view launch.
    python tools/experiments/counting_build.py && T4D_LIB=topo4d_amd/csrc/variants/lib_count.so python tools/count_lanes.py [C2|C4] [A|B]
Prints one JSON object: wave-steps, row-visits, contributing lanes -> useful-lane fraction and row balance of both kernels."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from topo4d_amd import _lib

_args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = _args[0] if len(_args) > 0 else "C2"
opa = _args[1] if len(_args) > 1 else "A"
dev = torch.device("cuda")
wl = bench.Workload(cfg, opa, dev, in_flight=1)
wl.learn_capacity()
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
torch.cuda.synchronize()
lib.t4d_debug_read_counters(buf, 1)
wl.step(0)
torch.cuda.synchronize()
lib.t4d_debug_read_counters(buf, 1)
c = [int(x) for x in buf]
st = wl.statuses()[0]
out = {"config": cfg, "opacity": opa, "views": wl.V, "pairs": int(st.total_pairs)}
for name, o in (("bwd", 0), ("fwd", 8)):
    tiles, batches, steps, visits, lanes = c[o:o + 5]
    out[name] = {"nonempty_tiles": tiles, "wave_batches": batches, "wave_steps": steps, "row_visits": visits,
                 "contributing_lane_steps": lanes,
                 "row_balance": round(visits / max(1, 4 * steps), 4),            # filled rows per wave-step (1 = every row busy)
                 "useful_lane_fraction": round(lanes / max(1, 64 * steps), 4),   # lanes that blend / contribute, of all lane-steps
                 "lanes_per_row_visit": round(lanes / max(1, visits), 2),
                 "steps_per_wave_batch": round(steps / max(1, batches), 1),
                 "row_visits_per_pair": round(visits / max(1, int(st.total_pairs)), 2)}
from topo4d_amd.build import raster_source_sha256
out["_kernel_source_sha256"] = raster_source_sha256()
print(json.dumps(out))
if "--merge" in sys.argv:                      # profiles/lanes.json: what bench.py reads for roofline.useful_lane_fraction
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "lanes.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[f"{cfg}{'' if opa == 'A' else '_' + opa}"] = out
    json.dump(cur, open(path, "w"), indent=1)
