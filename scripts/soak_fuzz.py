"""More seeds of the randomised sweep than the test-suite runs (GPU box):

    python scripts/soak_fuzz.py          # D = 1 ... 64, seeds 3 ... 42
    python scripts/soak_fuzz.py big      # the run-time-dimension unit: 21 dimensions 65 ... 300, seeds 10 ... 21
"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import fuzz_gpu
from pypmc_amd.backend import HipBackend
be = HipBackend()
worst = {}
big = len(sys.argv) > 1 and sys.argv[1] == "big"
BIG_DIMS = [65, 66, 70, 79, 80, 81, 95, 96, 97, 100, 112, 127, 128, 129, 144, 160, 161, 200, 256, 257, 300]
for seed in (range(10, 22) if big else range(3, 43)):
    if big:
        w = fuzz_gpu.sweep(seed=seed, rounds=1, be=be, verbose=False, dims=BIG_DIMS, kmax=12, nmax=900)
    else:
        w = fuzz_gpu.sweep(seed=seed, rounds=1, be=be, verbose=False)
    for k, v in w.items():
        worst[k] = max(worst.get(k, 0), v)
    print("seed", seed, "ok", flush=True)
for k, v in sorted(worst.items()):
    print("worst %-30s %.3g" % (k, v))
