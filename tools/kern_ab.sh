#!/bin/bash
# Runs ON THE GPU BOX: per-kernel HIP-event times of one bench configuration under the loaded library (T4D_LIB=... for a variant of
# tools/ab_build.sh).   usage: tools/kern_ab.sh <label> [bench args, e.g. --config C4]
L=$1; shift
python - "$L" "$@" <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--no-cpu-baseline", "--no-extras"] + sys.argv[2:], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
print(sys.argv[1], "step", d["ms_per_step"], {n: v["avg_us"] for n, v in d["roofline"]["kernels"].items()})
PY
