#!/usr/bin/env python3
"""Latency of the per-sample kernels at small and medium batches, with and without the components of a sample block split
over workgroups (round 6, verdict r5 #1).  `split_components` 0 is round 5's behaviour: one workgroup walks all K
components of its 256 samples.

    python scripts/smalln_latency.py [logpdf|frontend|estep|tail|all]
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
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timeit(fn, reps=200, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def kernel_us(fn, name="k_logpdf", reps=50):
    """the library's own event timing of the hot kernel (no host share)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    be.kernel_timing(True)
    be.kernel_timings()
    for _ in range(reps):
        fn()
    t = be.kernel_timings()
    be.kernel_timing(False)
    return sum(v["ms"] for k, v in t.items() if k.startswith(name)) / reps * 1e3


if what in ("logpdf", "all"):
    print("# mixture log-pdf (kernel level, samples resident), us per call: round-5 forms -> components in pieces")
    print("# exact = the exact kernels (maha_gemm off), mgemm = the matrix-product form where the shape takes it")
    for D, K in ((40, 128), (20, 32), (20, 16), (20, 128), (30, 32), (32, 32), (64, 64), (8, 32), (2, 8)):
        mu, cov, w = mk(K, D, 5)
        comps = gauss_set(mu, cov, w)[0]
        for N in (256, 1024, 4096, 16384, 65536, 262144):
            x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 3)
            f = lambda: be.logpdf(x, comps, want_scalars=True)
            row = []
            for label, opts in (("exact unsplit", dict(split_components=0, maha_gemm_min_n=2 ** 40)),
                                ("mgemm unsplit", dict(split_components=0, maha_gemm_min_n=256)),
                                ("exact pieces", dict(split_components=1, maha_gemm_min_n=2 ** 40))):
                for k, v in opts.items():
                    be.configure(k, v)
                row.append("%s %7.1f (kernel %7.1f)" % (label, timeit(f), kernel_us(f)))
            print("D=%2d K=%3d N=%7d   %s" % (D, K, N, "   ".join(row)), flush=True)
    be.configure("split_components", 1)
    be.configure("maha_gemm_min_n", 256)

if what in ("frontend", "all"):
    print("# front-end MixtureDensity.multi_evaluate (host arrays in and out), us per call")
    from pypmc_amd.density.mixture import create_gaussian_mixture
    for K, D, N in ((2, 2, 1000), (8, 10, 10000), (32, 20, 1000), (32, 20, 10000), (32, 20, 100000), (64, 40, 10000), (128, 40, 4096)):
        mix = create_gaussian_mixture(*mk(K, D, 1))
        np.random.seed(1)
        x = mix.propose(N)
        row = []
        for sp in (0, 1):
            be.configure("split_components", sp)
            row.append(timeit(lambda: mix.multi_evaluate(x), reps=100))
        print("K=%3d D=%2d N=%6d: multi_evaluate %7.1f -> %7.1f us per call" % (K, D, N, row[0], row[1]), flush=True)
    be.configure("split_components", 1)

if what in ("tail", "all"):
    print("# launches of a few rounds: the last round in pieces (us per call; kernel alone)")
    for D, K, N in ((20, 16, 1000000), (20, 36, 1250000), (20, 32, 1250000), (30, 36, 1250000), (20, 64, 1250000), (20, 36, 10000000)):
        mu, cov, w = mk(K, D, 5)
        comps = gauss_set(mu, cov, w)[0]
        x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 3)
        f = lambda: be.logpdf(x, comps, want_scalars=True)
        be.configure("maha_gemm_min_n", 2 ** 40)
        row = []
        for label, opts in (("unsplit", dict(split_components=0)), ("tail 2", dict(split_components=1, split_tail_pieces=2)),
                            ("tail 4", dict(split_components=1, split_tail_pieces=4)),
                            ("tail 8", dict(split_components=1, split_tail_pieces=8))):
            for k, v in opts.items():
                be.configure(k, v)
            row.append("%s %7.1f (kernel %7.1f)" % (label, timeit(f, reps=50), kernel_us(f, reps=30)))
        print("D=%2d K=%3d N=%8d   %s" % (D, K, N, "   ".join(row)), flush=True)
    be.configure("split_components", 1)
    be.configure("split_tail_pieces", 4)
    be.configure("maha_gemm_min_n", 256)

if what in ("estep", "all"):
    print("# VB E-step at one GPU's share of eight (grouped responsibilities + common-shift statistics), us per call")
    from test_gpu_split import _vb_set
    for D, K, N in ((20, 32, 10000), (20, 64, 10000), (40, 128, 4096), (20, 32, 100000), (20, 128, 50000)):
        cs, (mu, cov, w), _ = _vb_set(K, D, 600 + K)
        x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 3)
        f = lambda: be.estep(x, cs, 0)
        row = []
        for label, on in (("one workgroup per block (k_resp)", 0), ("groups in pieces", 1)):
            be.configure("estep_small_batch_pieces", on)
            row.append("%s %7.1f (k_resp %7.1f, k_stats %7.1f)" % (label, timeit(f, reps=100), kernel_us(f, "k_resp", reps=50), kernel_us(f, "k_stats", reps=50)))
        be.reset_option("estep_small_batch_pieces")
        print("small batch D=%2d K=%3d N=%7d   %s" % (D, K, N, "   ".join(row)), flush=True)
    for D, K, N in ((20, 64, 1250000), (20, 32, 1250000), (20, 64, 312500), (20, 32, 100000)):
        cs, (mu, cov, w), _ = _vb_set(K, D, 600 + K)
        x = be.asdevice(np.random.RandomState(1).normal(size=(N, D)) * 3)
        f = lambda: be.estep(x, cs, 0)
        row = []
        for label, opts in (("unsplit", dict(split_components=0)), ("tail 2", dict(split_components=1, split_tail_pieces=2)),
                            ("tail 4", dict(split_components=1, split_tail_pieces=4))):
            for k, v in opts.items():
                be.configure(k, v)
            row.append("%s %7.1f (k_resp %7.1f)" % (label, timeit(f, reps=50), kernel_us(f, "k_resp", reps=30)))
        print("D=%2d K=%3d N=%8d   %s" % (D, K, N, "   ".join(row)), flush=True)
    be.configure("split_components", 1)
    be.configure("split_tail_pieces", 4)
