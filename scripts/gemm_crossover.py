#!/usr/bin/env python3
"""Where the common-shift statistics (k_stats_gemm + its finishing kernels + the two skipped launches) start to pay
against the per-component-shift kernel: pmc_estep's statistics half over N and K (D = 20 unless given), both forms
in one process (pmc_configure).  Run on the GPU box."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypmc_amd.backend import HipBackend, ComponentSet

be = HipBackend()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(1)


def run(N, K):
    mu = rs.normal(0, 3, (K, D))
    A = rs.normal(size=(K, D, D))
    cov = np.einsum('kij,klj->kil', A, A) / D + 0.5 * np.eye(D)
    inv = np.linalg.inv(cov)
    nu = D + 2. + np.arange(K) * 0.1
    W = inv / nu[:, None, None]
    W = 0.5 * (W + W.transpose(0, 2, 1))
    vb = ComponentSet(2, mu, W, c0=D / (1. + np.arange(K)), c1=nu, c2=np.log(np.full(K, 1. / K)), c3=np.linalg.slogdet(W)[1] + 3.)
    x = torch.tensor(mu, device="cuda")[torch.randint(K, (N,), device="cuda")] + torch.randn(N, D, dtype=torch.float64, device="cuda")
    pack = be.pack(vb)
    out = be.zeros(be.stats_len(K, D))
    res = {}
    for label, limit, mink in (("common", 1000.0, 1), ("per-component", 0.0, 1)):
        be.configure("stats_common_shift_limit", limit)
        be.configure("stats_common_shift_min_k", mink)
        for _ in range(3):
            be.estep(x, vb, 0, pack=pack, out=out)
        torch.cuda.synchronize()
        be.kernel_timings()
        be.kernel_timing(True)
        for _ in range(10):
            be.estep(x, vb, 0, pack=pack, out=out)
        torch.cuda.synchronize()
        be.kernel_timing(False)
        kt = be.kernel_timings()
        res[label] = (kt["k_stats"]["ms"] + kt["finishing reductions"]["ms"]) / 10.0
    be.configure("stats_common_shift_limit", 1000.0)
    be.configure("stats_common_shift_min_k", 17)
    return res                                           # (stats_common_shift_min_n stays 0 for the sweep)


if len(sys.argv) > 2 and sys.argv[2] == "ksweep":       # which component counts fill the 16-component row blocks well enough
    be.configure("stats_common_shift_min_n", 0)
    for K in list(range(17, 50)) + [56, 64, 65, 72, 80, 96, 100, 128]:
        r = run(2000000, K)
        print("D=%d K=%3d N=2000000  common-shift %.4f ms   per-component %.4f ms   ratio %.2f" % (D, K, r["common"], r["per-component"], r["per-component"] / r["common"]), flush=True)
    sys.exit(0)
be.configure("stats_common_shift_min_n", 0)
for K in (8, 12, 16, 17, 24, 32):
    for N in (16384, 32768, 65536, 262144, 4000000):
        if K < 17 and N not in (65536, 4000000):
            continue
        r = run(N, K)
        print("D=%d K=%3d N=%8d  common-shift %.4f ms   per-component %.4f ms   ratio %.2f" % (D, K, N, r["common"], r["per-component"], r["per-component"] / r["common"]), flush=True)
