"""k_mgemm with components without weight (pruned components of a PMC run): log-pdf pass, matrix-product form against the exact
engine, all components alive against a fifth of them dead."""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from pypmc_amd.backend import HipBackend
from test_gpu_kernels import mk, gauss_set
from pypmc_amd.density.mixture import create_gaussian_mixture
be = HipBackend()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def both(fn):
    be.configure("maha_gemm_tolerance", 0.0); a = fn(); be.configure("maha_gemm_tolerance", 5e-11); b = fn(); return a, b
for D, K, N in [(40, 128, 1000000), (32, 64, 2000000), (64, 64, 1000000), (24, 128, 1000000)]:
    mu, cov, w = mk(K, D, 5)
    x = create_gaussian_mixture(mu, cov, w).propose(N, np.random.mtrand.RandomState(7), device=True)
    for frac in (0.0, 0.2):
        wd = np.where(np.arange(K) % 5 == 2, 0.0, w) if frac else w.copy()
        wd /= wd.sum()
        comps = gauss_set(mu, cov, wd)[0]
        ex, ge = both(lambda: be.tohost(be.logpdf(x, comps, want_scalars=True)["out"]))
        rep = be.maha_gemm_report(N, K, D)
        t_ex, t_ge = both(lambda: timeit(lambda: be.logpdf(x, comps, want_scalars=True)))
        print("D=%d K=%d dead %2.0f %%: refused %d of %d, max|gemm-exact| %.2e; exact %.3f ms, gemm %.3f ms (%+.1f %%)"
              % (D, K, 100 * frac, rep["refused"], rep["workgroups"], np.abs(ex - ge).max(), t_ex, t_ge, 100 * (t_ge / t_ex - 1)), flush=True)
    del x; be.release(); torch.cuda.empty_cache()
