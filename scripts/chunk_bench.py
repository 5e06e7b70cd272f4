#!/usr/bin/env python3
"""Experiment: the D=20 E-step (k_resp -> u -> k_stats) over the whole batch against the same in chunks small
enough for the tile-major u buffer to stay in the 256 MB Infinity Cache between the two kernels."""
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
    N, K, D = 10_000_000, bench.K, bench.D
    mu, cov, w = bench.mk(K, D, 1)
    W, beta, nu, ln_pi, ln_lambda = bench.vb_params(mu, cov, w, N)
    posterior = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    g = torch.Generator(device="cuda").manual_seed(1)
    comp = torch.multinomial(torch.tensor(w, device="cuda"), N, replacement=True, generator=g)
    x = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g)
    L = torch.tensor(np.linalg.cholesky(cov), device="cuda")
    x = torch.einsum('nij,nj->ni', L[comp], x) + torch.tensor(mu, device="cuda")[comp]
    stats = be.zeros(be.stats_len(K, D))
    part = be.zeros(be.stats_len(K, D))

    def run(chunk):
        stats.zero_()
        for i in range(0, N, chunk):
            be.estep(x[i:i + chunk], posterior, 0, out=part)
            stats.add_(part)

    ref = None
    for chunk in (N, 2_500_000, 1_000_000, 500_000, 250_000, N):
        for _ in range(2):
            run(chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run(chunk)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        if ref is None:
            ref = stats.clone()
        err = float(((stats - ref).abs() / (ref.abs() + 1e-300)).max())
        print("chunk %9d  (u = %6.0f MB): %.3f ms per E-step   max rel diff of the statistics %.1e"
              % (chunk, chunk * K * 8 / 1e6, dt, err), flush=True)


if __name__ == "__main__":
    main()
