#!/bin/bash
# Runs ON THE GPU BOX: the default `bench.py` line under the loaded library (T4D_LIB=... for a variant of tools/ab_build.sh), reduced to the
# numbers an A/B needs: headline, per-kernel times of config 2, config 4 and the 1 M dense pass, the texture iteration.
#   usage: tools/full_ab.sh <label>
python - "$1" <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
k = lambda r: {n: v["avg_us"] for n, v in r["kernels"].items()}
print(sys.argv[1], "value", d["value"], "c2", k(d["roofline"]))
print(sys.argv[1], "c4 ms", d["c4"]["ms_per_step"]["median"], k(d["c4"]["roofline"]))
print(sys.argv[1], "dense", d["dense_1m"]["ms_per_view"], d["dense_1m"]["kernels_us"], "tex_it", d["dense_1m"]["texture_iteration"].get("ms_per_iteration"))
PY
