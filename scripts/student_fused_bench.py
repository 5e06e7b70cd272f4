#!/usr/bin/env python3
"""Student-t PMC E-step at small sample dimensions: the one-kernel form (pmc_estep, compiled D = 3 ... 7) against the two
kernels it replaces (pmc_responsibilities + pmc_sufficient_stats).  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypmc_amd.backend import HipBackend, ComponentSet
from scipy.special import gammaln
be = HipBackend()
N = 4_000_000
for D, K in ((3, 8), (3, 32), (5, 32), (7, 32), (4, 16)):
    rs = np.random.RandomState(D)
    mu = rs.normal(0, 3, (K, D))
    A = rs.normal(size=(K, D, D))
    cov = np.einsum('kij,klj->kil', A, A) / D + 0.5 * np.eye(D)
    inv = np.linalg.inv(cov)
    dof = np.full(K, 6.)
    ln = gammaln(.5 * (dof + D)) - gammaln(.5 * dof) - 0.5 * D * np.log(dof * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
    cs = ComponentSet(1, mu, inv, c0=ln, c1=-.5 * (dof + D), c2=1. / dof, c3=dof, weight=np.full(K, 1. / K))
    x = torch.tensor(mu, device="cuda")[torch.randint(K, (N,), device="cuda")] + torch.randn(N, D, dtype=torch.float64, device="cuda")
    pack = be.pack(cs)
    out = be.zeros(be.stats_len(K, D))
    u, scratch, ws = be._tilebuf("u", N, K), be._tilebuf("scratch", N, K), be._workspace(N, K, D)
    P, lib = be._p, be.lib

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return min(ts)
    ps = int(lib.pmc_stats_stride(D))
    one = timeit(lambda: be.estep(x, cs, 1, pack=pack, out=out))

    def two():
        lib.pmc_responsibilities(P(x), N, D, P(pack), K, 1, 1, 0, P(None), P(None), P(u), P(scratch), P(out[8 + K * ps:]),
                                 P(None), P(None), P(None), K, P(out), P(ws), be._stream())
        lib.pmc_sufficient_stats(P(x), N, D, P(pack), K, P(u), P(out[8:]), P(ws), be._stream())
    t2 = timeit(two)
    print("D=%d K=%2d N=%d  Student-t PMC E-step: one kernel %.3f ms (fused=%d), two kernels %.3f ms, ratio %.2f"
          % (D, K, N, one, lib.pmc_estep_is_fused(K, D, 1, 1), t2, t2 / one), flush=True)
