#!/usr/bin/env python3
"""From how many samples per call on does the matrix-product form of the Mahalanobis forms pay?  (Its coefficient image
is built per call, k_theta_build, whatever N is.)  Times mixture log-pdf calls of both forms at small N.

    python scripts/mgemm_crossover.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from pypmc_amd.backend import HipBackend  # noqa: E402
from test_gpu_kernels import mk, gauss_set  # noqa: E402

be = HipBackend()


def timeit(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


be.configure("maha_gemm_min_n", 0)
shapes = ((40, 128), (32, 32), (48, 64), (40, 32), (64, 64), (64, 128), (24, 64), (24, 128), (20, 128), (36, 64))
sizes = (256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144)
if len(sys.argv) > 1 and sys.argv[1] == "small":
    sizes = (256, 512, 1024, 2048, 4096, 16384, 65536)
for D, K in shapes:
    mu, cov, w = mk(K, D, 5)
    comps = gauss_set(mu, cov, w)[0]
    for N in sizes:
        x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 3)
        be.configure("maha_gemm_tolerance", 0.0)
        t_ex = timeit(lambda: be.logpdf(x, comps, want_scalars=True))
        be.configure("maha_gemm_tolerance", 5e-11)
        t_ge = timeit(lambda: be.logpdf(x, comps, want_scalars=True))
        print("D=%d K=%3d N=%7d   exact %8.1f us   matrix product %8.1f us   %s" % (D, K, N, t_ex, t_ge, "<-" if t_ge < t_ex else ""),
              flush=True)
