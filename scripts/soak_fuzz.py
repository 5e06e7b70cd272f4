import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import fuzz_gpu
from pypmc_amd.backend import HipBackend
be = HipBackend()
worst = {}
for seed in range(3, 43):
    w = fuzz_gpu.sweep(seed=seed, rounds=1, be=be, verbose=False)
    for k, v in w.items():
        worst[k] = max(worst.get(k, 0), v)
    print("seed", seed, "ok", flush=True)
for k, v in sorted(worst.items()):
    print("worst %-30s %.3g" % (k, v))
