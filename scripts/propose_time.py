import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bench import mk
from pypmc_amd.backend import HipBackend
be = HipBackend(0)
for D, K, N in ((40, 128, 12_500_000), (30, 32, 10_000_000), (20, 32, 10_000_000), (48, 64, 4_000_000), (64, 64, 4_000_000), (40, 4, 4_000_000)):
    mu, cov, w = mk(K, D, 5)
    chol = np.linalg.cholesky(cov)
    counts = np.random.RandomState(1).multinomial(N, w)
    for dof in (None, np.full(K, 6.0)):
        x, _ = be.propose(mu, chol, dof, counts, 7)
        torch.cuda.synchronize()
        be.kernel_timings(); be.kernel_timing(True)
        for _ in range(5):
            be.propose(mu, chol, dof, counts, 7, out=x)
        torch.cuda.synchronize()
        be.kernel_timing(False)
        t = be.kernel_timings()["k_propose"]
        print("D=%d K=%d N=%.3g %s: k_propose %.3f ms  (%.0f ps per sample)" % (D, K, N, "student" if dof is not None else "gauss  ", t["ms"] / t["calls"], t["ms"] / t["calls"] * 1e9 / N), flush=True)
    del x
