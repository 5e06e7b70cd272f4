"""More seeds of the randomised sweep than the test-suite runs (GPU box):

    python scripts/soak_fuzz.py          # D = 1 ... 64, seeds 3 ... 42
    python scripts/soak_fuzz.py big      # the run-time-dimension unit: 21 dimensions 65 ... 300, seeds 10 ... 21
    python scripts/soak_fuzz.py fast     # the fast paths (k_mgemm from 256 samples, k_resp_groups, k_stats_gemm, the emitting
                                         # pass): the dimensions that have them, K up to 140, seeds 100 ... 139
"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import fuzz_gpu
from pypmc_amd.backend import HipBackend
be = HipBackend()
worst = {}
big = len(sys.argv) > 1 and sys.argv[1] == "big"
fast = len(sys.argv) > 1 and sys.argv[1] == "fast"
FAST_DIMS = [8, 12, 16, 17, 20, 21, 24, 30, 31, 32, 36, 40, 41, 48, 49, 56, 64]
BIG_DIMS = [65, 66, 70, 79, 80, 81, 95, 96, 97, 100, 112, 127, 128, 129, 144, 160, 161, 200, 256, 257, 300]
for seed in (range(10, 22) if big else (range(100, 100 + int(sys.argv[2]) if len(sys.argv) > 2 else 140) if fast else range(3, 43))):
    if fast:
        w = fuzz_gpu.sweep(seed=seed, rounds=1, be=be, verbose=False, dims=FAST_DIMS, kmax=140, fast_paths=True)
    elif big:
        w = fuzz_gpu.sweep(seed=seed, rounds=1, be=be, verbose=False, dims=BIG_DIMS, kmax=12, nmax=900)
    else:
        w = fuzz_gpu.sweep(seed=seed, rounds=1, be=be, verbose=False)
    for k, v in w.items():
        worst[k] = max(worst.get(k, 0), v)
    print("seed", seed, "ok", flush=True)
for k, v in sorted(worst.items()):
    print("worst %-30s %.3g" % (k, v))
