"""Soak run of tests/test_gpu_configs.py::test_randomised_scenes_against_c_oracle over many more seeds than the suite holds
(GPU box):   python tools/soak_parity.py [first_trial] [n_trials]     -> one line per failing trial + a summary line."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_configs import test_randomised_scenes_against_c_oracle as one

first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = []
for builds in ("0", "1000000000"):                       # throughput build, latency build
    os.environ["T4D_LATENCY_TILES"] = builds
    for t in range(first, first + n):
        try:
            one(t)
        except Exception as e:                            # assertion text names the quantity and the worst element
            bad.append((builds, t))
            print(f"trial {t} (T4D_LATENCY_TILES={builds}) FAILED: {str(e).splitlines()[0][:200]}", flush=True)
print(f"soak: {2 * n} runs (trials {first}..{first + n - 1}, both builds), {len(bad)} failures {bad[:20]}")
