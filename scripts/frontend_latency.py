"""Per-call latency of the front-end on small problems (host arrays in and out, as a pypmc script calls it)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pypmc_amd.density.mixture import create_gaussian_mixture
def mk(K, D, seed):
    rs = np.random.RandomState(seed); mu = rs.normal(0, 3, size=(K, D)); cov = np.empty((K, D, D))
    for k in range(K):
        A = rs.normal(0, 1, size=(D, D)); cov[k] = A.dot(A.T) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K); return mu, cov, w / w.sum()
for K, D, N in ((2, 2, 1000), (2, 2, 10000), (4, 5, 1000), (8, 10, 10000), (32, 20, 10000), (32, 20, 100000), (64, 40, 10000)):
    mix = create_gaussian_mixture(*mk(K, D, 1))
    np.random.seed(1)
    x = mix.propose(N)
    for _ in range(5): mix.multi_evaluate(x)
    t0 = time.perf_counter(); reps = 100
    for _ in range(reps): out = mix.multi_evaluate(x)
    t = (time.perf_counter() - t0) / reps
    print("K=%2d D=%2d N=%6d: multi_evaluate %7.1f us per call (%.2e pairs/s)" % (K, D, N, t * 1e6, N * K / t), flush=True)
