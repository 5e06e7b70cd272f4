#!/usr/bin/env python3
"""Does the last round in pieces ever cost?  Every compiled dimension, launches of a few rounds: mixture log-pdf and importance weights
with split_components 0 / 1, kernel times (us).  A ratio above 1.01 would be a regression."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from pypmc_amd.backend import HipBackend
from test_gpu_kernels import mk, gauss_set, student_set
be = HipBackend()
be.configure("maha_gemm_min_n", 2 ** 40)
def kernel_us(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); be.kernel_timing(True); be.kernel_timings()
    for _ in range(reps): fn()
    t = be.kernel_timings(); be.kernel_timing(False)
    return sum(v["ms"] for k, v in t.items() if k.startswith("k_logpdf")) / reps * 1e3
worst = 0.0
for D in (1, 2, 4, 8, 12, 16, 20, 24, 30, 32, 40, 48, 64, 23, 36):
    for K, student in ((8, False), (32, False), (32, True)):
        N = 2_000_000 if D <= 32 else 600_000
        mu, cov, w = mk(K, D, 5)
        cs = student_set(mu, cov, w, np.full(K, 6.))[0] if student else gauss_set(mu, cov, w)[0]
        tg = gauss_set(*mk(4, D, 9))[0]
        x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 2.5)
        f = lambda: be.importance_weights(x, cs, tg)
        ts = []
        for sp in (0, 1, 0, 1):
            be.configure("split_components", sp)
            ts.append(kernel_us(f))
        off, on = min(ts[0], ts[2]), min(ts[1], ts[3])
        worst = max(worst, on / off)
        print("D=%2d K=%2d %s N=%d: one workgroup per block %8.1f   last round in pieces %8.1f   ratio %.3f" %
              (D, K, "t" if student else "g", N, off, on, on / off), flush=True)
        del x
be.reset_option("split_components")
print("worst ratio %.3f" % worst)
