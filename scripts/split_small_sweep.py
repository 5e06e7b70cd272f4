#!/usr/bin/env python3
"""Small launches (less than one round of the chip): how finely to cut the components of a block (pmc_api.hip::split_plan:
split_fill = workgroups per slot aimed at, split_min_components = smallest piece).  us per call, samples resident.

    python scripts/split_small_sweep.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from pypmc_amd.backend import HipBackend  # noqa: E402
from test_gpu_kernels import mk, gauss_set  # noqa: E402

be = HipBackend()
be.configure("maha_gemm_min_n", 2 ** 40)
be.configure("split_max_pieces", 1024)


def us(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for D, K, N in ((40, 128, 4096), (40, 128, 1024), (40, 128, 16384), (64, 64, 4096), (20, 32, 10000), (20, 32, 1000), (20, 128, 4096),
                (30, 32, 4096), (8, 32, 4096), (20, 16, 30000)):
    mu, cov, w = mk(K, D, 5)
    comps = gauss_set(mu, cov, w)[0]
    x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 3)
    f = lambda: be.logpdf(x, comps)
    be.configure("split_components", 0)
    row = ["D=%2d K=%3d N=%6d  unsplit %6.1f |" % (D, K, N, us(f))]
    be.configure("split_components", 1)
    for fill in (0.5, 1.0, 2.0, 4.0):
        be.configure("split_fill", fill)
        for minc in (1, 2, 4, 8):
            be.configure("split_min_components", minc)
            row.append("f%.1f m%d %5.1f" % (fill, minc, us(f)))
    be.reset_option("split_fill")
    be.reset_option("split_min_components")
    print("  ".join(row), flush=True)
