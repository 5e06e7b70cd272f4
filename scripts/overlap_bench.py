#!/usr/bin/env python3
"""Experiment: the bench step's two independent halves -- importance weights (k_logpdf, vector pipe) and the
VB E-step (k_resp, then k_stats on the matrix pipe) -- back to back on one stream against side by side on two."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from pypmc_amd.backend import HipBackend, ComponentSet
    be = HipBackend()
    N, K, D, K_T = 10_000_000, bench.K, bench.D, bench.K_T
    mu, cov, w = bench.mk(K, D, 1)
    tmu, tcov, tw = bench.mk(K_T, D, 11)
    inv, ln = bench.gauss_params(mu, cov)
    tinv, tln = bench.gauss_params(tmu, tcov)
    W, beta, nu, ln_pi, ln_lambda = bench.vb_params(mu, cov, w, N)
    proposal = ComponentSet(0, mu, inv, c0=ln, weight=w)
    target = ComponentSet(0, tmu, tinv, c0=tln, weight=tw)
    posterior = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    g = torch.Generator(device="cuda").manual_seed(1)
    comp = torch.multinomial(torch.tensor(w, device="cuda"), N, replacement=True, generator=g)
    x = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g)
    L = torch.tensor(np.linalg.cholesky(cov), device="cuda")
    x = torch.einsum('nij,nj->ni', L[comp], x) + torch.tensor(mu, device="cuda")[comp]
    stats = be.zeros(be.stats_len(K, D))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def sequential():
        be.importance_weights(x, proposal, target)
        be.estep(x, posterior, 0, out=stats)

    def overlapped():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            be.importance_weights(x, proposal, target)
        with torch.cuda.stream(s2):
            be.estep(x, posterior, 0, out=stats)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    for name, fn in (("one stream", sequential), ("two streams", overlapped), ("one stream", sequential),
                     ("two streams", overlapped)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print("%-12s %.3f ms per step" % (name, (time.perf_counter() - t0) / 10 * 1e3), flush=True)


if __name__ == "__main__":
    main()
