#!/bin/bash
# Runs ON THE GPU BOX: headline value / sequential step / the two render kernels of `bench.py` under the loaded library (T4D_LIB=... for a
# variant of tools/ab_build.sh).   usage: tools/value_ab.sh <label> [bench args]
L=$1; shift
python - "$L" "$@" <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extras"] + sys.argv[2:], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "sequential", d["sequential"]["ms_per_step"], "bwd", d["roofline"]["kernels"]["k_render_bwd"]["avg_us"], "fwd", d["roofline"]["kernels"]["k_render_fwd"]["avg_us"])
PY
