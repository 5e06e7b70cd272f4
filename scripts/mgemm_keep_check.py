"""The weighting pass that KEEPS the Mahalanobis forms (pmc_importance_weights_keep: the pass of a PMC iteration that cannot emit,
e.g. with pruned components) through the matrix-product form against the exact engine, and the iteration's two calls together."""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from pypmc_amd.backend import HipBackend, ComponentSet
from test_gpu_kernels import mk, gauss_set
from pypmc_amd.density.mixture import create_gaussian_mixture
be = HipBackend()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for D, K, N in [(40, 128, 1000000), (64, 64, 1000000), (32, 64, 2000000)]:
    mu, cov, w = mk(K, D, 5)
    x = create_gaussian_mixture(mu, cov, w).propose(N, np.random.mtrand.RandomState(7), device=True)
    wd = np.where(np.arange(K) % 5 == 2, 0.0, w); wd /= wd.sum()
    cs, inv, ln = gauss_set(mu, cov, wd)
    live = np.flatnonzero(wd > 0)
    sub = ComponentSet(0, mu[live], inv[live], c0=ln[live], weight=wd[live], column=live, ld=K)
    target = gauss_set(*mk(4, D, 6))[0]
    def iteration():
        r = be.importance_weights(x, cs, target, keep=True)
        return be.estep_from_tiles(x, sub, r["tiles"], max_init_zero=True, sample_w=r["weights"])
    for tol in (0.0, 5e-11):
        be.configure("maha_gemm_tolerance", tol)
        t_pass = timeit(lambda: be.importance_weights(x, cs, target, keep=True))
        t_it = timeit(iteration)
        print("D=%d K=%d (a fifth pruned) %s: weighting pass with kept forms %.3f ms, pass + update from the forms %.3f ms"
              % (D, K, "exact engine  " if tol == 0 else "matrix product", t_pass, t_it), flush=True)
    be.configure("maha_gemm_tolerance", 5e-11)
    del x; be.release(); torch.cuda.empty_cache()
