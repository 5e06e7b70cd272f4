"""Student-t mixtures through the matrix-product form: how many workgroups does the a-priori guard refuse at the default
tolerance (its price uses the slope |da / dmaha| at maha = 0, (nu + D) / 2 nu, for every pair), and what would the form give?"""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from pypmc_amd.backend import HipBackend
from test_gpu_kernels import mk, student_set
from pypmc_amd.density.mixture import create_t_mixture
be = HipBackend()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for D, K, N, dof in [(40, 128, 1000000, 8.0), (40, 128, 1000000, 3.0), (32, 64, 2000000, 8.0), (64, 64, 1000000, 8.0), (40, 128, 1000000, 50.0)]:
    mu, cov, w = mk(K, D, 5)
    dofs = np.full(K, dof)
    x = create_t_mixture(mu, cov, dofs, w).propose(N, np.random.mtrand.RandomState(7), device=True)
    comps = student_set(mu, cov, w, dofs)[0]
    be.configure("maha_gemm_tolerance", 0.0); ex = be.tohost(be.logpdf(x, comps)["out"]); t_ex = timeit(lambda: be.logpdf(x, comps, want_scalars=True))
    for tol in (5e-11, 1e-9):
        be.configure("maha_gemm_tolerance", tol)
        ge = be.tohost(be.logpdf(x, comps)["out"]); rep = be.maha_gemm_report(N, K, D)
        t_ge = timeit(lambda: be.logpdf(x, comps, want_scalars=True))
        print("D=%d K=%d nu=%g tol %g: refused %d of %d; max|gemm-exact| %.2e; exact %.3f ms, with the form %.3f ms (%+.1f %%)"
              % (D, K, dof, tol, rep["refused"], rep["workgroups"], np.abs(ex - ge).max(), t_ex, t_ge, 100 * (t_ge / t_ex - 1)), flush=True)
    be.configure("maha_gemm_tolerance", 5e-11)
    del x; be.release(); torch.cuda.empty_cache()
