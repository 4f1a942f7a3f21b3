"""Soak run of the randomised parity scenes (tests/test_gpu_configs.py::_randomised_trial) over many more seeds than the suite
holds, every build of the render kernels (GPU box):
    python tools/soak_parity.py [first_trial] [n_trials] [--strict]     -> one line per failing trial + a summary line.
--strict: plain gradient tolerance, no allowance for Gaussians that hold a threshold pixel (how round 2 found trial seeds whose
ONE flipped pixel moves one Gaussian's gradient by 3e-4 of the scene's largest)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_configs import BUILDS, _randomised_trial as one

args = [a for a in sys.argv[1:] if not a.startswith("--")]
strict = "--strict" in sys.argv
first = int(args[0]) if len(args) > 0 else 12
n = int(args[1]) if len(args) > 1 else 200
bad, flips = [], 0
for name, env in BUILDS:                                  # throughput / latency + segments / throughput + segments
    os.environ.pop("T4D_NO_SEGMENTS", None)
    os.environ.update(env)
    for t in range(first, first + n):
        try:
            flips += one(t, strict=strict)
        except Exception as e:                            # assertion text names the quantity and the worst element
            bad.append((name, t))
            print(f"trial {t} ({name}) FAILED: {str(e).splitlines()[0][:200]}", flush=True)
print(f"soak{' (strict)' if strict else ''}: {len(BUILDS) * n} runs (trials {first}..{first + n - 1}, {len(BUILDS)} builds), {flips} threshold pixels, "
      f"{len(bad)} failures {bad[:20]}")
