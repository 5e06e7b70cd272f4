#!/usr/bin/env python3
"""pmc_estep with k_resp + k_stats_gemm against k_resp_groups + k_stats_gemm (factors applied by the statistics kernel):
library timing per kernel, same process (pmc_configure "estep_grouped_responsibilities").  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypmc_amd.backend import HipBackend, ComponentSet
be = HipBackend()
be.configure("stats_common_shift_min_fill", 0)
cases = ((20, 32, 10_000_000), (20, 64, 10_000_000), (20, 128, 4_000_000), (40, 128, 2_000_000), (12, 32, 10_000_000), (30, 32, 4_000_000))
if len(sys.argv) > 1 and sys.argv[1] == "matrix":
    cases = [(D, K, 2_000_000) for D in (8, 10, 12, 16, 20, 24, 30, 32, 40, 48, 64) for K in (32, 64, 128)]
for D, K, N in cases:
    rs = np.random.RandomState(1)
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
    line = "D=%d K=%3d N=%.0e " % (D, K, N)
    for grouped in (0, 2, 0, 2):                              # 2 = grouped wherever the kernels exist (no pays-rule)
        be.configure("estep_grouped_responsibilities", grouped)
        for _ in range(2):
            be.estep(x, vb, 0, pack=pack, out=out)
        torch.cuda.synchronize()
        be.kernel_timings(); be.kernel_timing(True)
        for _ in range(5):
            be.estep(x, vb, 0, pack=pack, out=out)
        torch.cuda.synchronize()
        be.kernel_timing(False)
        kt = be.kernel_timings()
        line += " | %s resp %.3f stats %.3f fin %.3f" % ("groups" if grouped else "k_resp", kt["k_resp"]["ms"] / 5, kt["k_stats"]["ms"] / 5, kt["finishing reductions"]["ms"] / 5)
    print(line, flush=True)
    del x
be.configure("estep_grouped_responsibilities", 1)
