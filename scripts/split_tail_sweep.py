#!/usr/bin/env python3
"""Tail-mode parameters of the split kernels (pmc_api.hip::split_plan): how many rounds in front of the last one go in
pieces, how many pieces per block.  Kernel times (the library's events), us.

    python scripts/split_tail_sweep.py
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
from test_gpu_split import _vb_set  # noqa: E402

be = HipBackend()
be.configure("maha_gemm_min_n", 2 ** 40)


def kernel_us(fn, name, reps=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    be.kernel_timing(True)
    be.kernel_timings()
    for _ in range(reps):
        fn()
    t = be.kernel_timings()
    be.kernel_timing(False)
    return sum(v["ms"] for k, v in t.items() if k.startswith(name)) / reps * 1e3


def sweep(label, fn, name):
    be.configure("split_components", 0)
    base = kernel_us(fn, name)
    be.configure("split_components", 1)
    out = ["%-34s unsplit %7.1f |" % (label, base)]
    for rounds in (0.0, 0.125, 0.25, 0.5, 1.0):
        be.configure("split_tail_rounds", rounds)
        for pieces, minc in ((2, 1), (4, 8), (8, 4)):
            be.configure("split_tail_pieces", pieces)
            be.configure("split_tail_min_components", minc)
            out.append("r%.1f p%d m%d %7.1f" % (rounds, pieces, minc, kernel_us(fn, name)))
    for k in ("split_tail_rounds", "split_tail_pieces", "split_tail_min_components"):
        be.reset_option(k)
    print("  ".join(out), flush=True)


for D, K, N in ((20, 16, 1000000), (20, 36, 1250000), (30, 36, 1250000), (20, 64, 1250000), (20, 36, 2500000), (20, 36, 5000000),
                (40, 128, 400000)):
    mu, cov, w = mk(K, D, 5)
    comps = gauss_set(mu, cov, w)[0]
    x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 3)
    sweep("logpdf D=%d K=%d N=%d" % (D, K, N), lambda: be.logpdf(x, comps, want_scalars=True), "k_logpdf")
for D, K, N in ((20, 64, 1250000), (20, 32, 1250000), (20, 64, 625000)):
    cs = _vb_set(K, D, 600 + K)[0]
    x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 3)
    sweep("resp_groups D=%d K=%d N=%d" % (D, K, N), lambda: be.estep(x, cs, 0), "k_resp")
