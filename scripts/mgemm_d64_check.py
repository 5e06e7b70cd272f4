import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from pypmc_amd.backend import HipBackend
from test_gpu_kernels import mk, gauss_set, student_set
from oracle import oracle as orc
be = HipBackend()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def both(fn):
    be.configure("maha_gemm_tolerance", 0.0); a = fn(); be.configure("maha_gemm_tolerance", 5e-11); b = fn(); return a, b
for D, K, N in [(64, 64, 1000000), (64, 128, 500000), (56, 64, 1000000), (64, 32, 1000000), (48, 64, 1000000)]:
    mu, cov, w = mk(K, D, 5)
    from pypmc_amd.density.mixture import create_gaussian_mixture
    x = create_gaussian_mixture(mu, cov, w).propose(N, np.random.mtrand.RandomState(7), device=True)
    comps, inv, ln = gauss_set(mu, cov, w)
    ex, ge = both(lambda: be.tohost(be.logpdf(x, comps, want_scalars=True)["out"]))
    rep = be.maha_gemm_report(N, K, D)
    ref, _ = orc.mixture_multi_evaluate(0, be.tohost(x[:3000]), w, mu, inv, ln)
    print("D=%d K=%d: report %s  max|gemm-exact| %.3e  vs oracle (3000): gemm %.3e exact %.3e" % (D, K, None if rep is None else (rep["refused"], rep["workgroups"]),
          np.abs(ex - ge).max(), np.max(np.abs(ge[:3000] - ref) / np.abs(ref)), np.max(np.abs(ex[:3000] - ref) / np.abs(ref))), flush=True)
    t_ex, t_ge = both(lambda: timeit(lambda: be.logpdf(x, comps, want_scalars=True)))
    print("   exact %.3f ms  gemm %.3f ms  (%.2f -> %.2f ps/pair, %+.1f %%)" % (t_ex, t_ge, t_ex * 1e9 / (N * K), t_ge * 1e9 / (N * K), 100 * (t_ge / t_ex - 1)), flush=True)
