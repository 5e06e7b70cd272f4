"""k_mgemm tried at D = 20 / 24 (variant library built with -DPMC_MG_TRY_SMALL): the log-pdf pass and the Gaussian-PMC E-step,
matrix-product form against the exact kernels, on the same samples.

    PMC_VARIANT=mgsmall PMC_VARIANT_UNITS=pmc_mgemm_d20_p0,pmc_mgemm_d24_p0 PMC_EXTRA_FLAGS=-DPMC_MG_TRY_SMALL python -m pypmc_amd.build
    PMC_HIP_LIBRARY=pypmc_amd/lib/libpmc_hip_mgsmall.so python scripts/mgemm_small_check.py
"""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from pypmc_amd.backend import HipBackend
from test_gpu_kernels import mk, gauss_set
be = HipBackend()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def both(fn):
    be.configure("maha_gemm_tolerance", 0.0); a = fn(); be.configure("maha_gemm_tolerance", 5e-11); b = fn(); return a, b
for D, K, N in [(20, 32, 4000000), (20, 64, 2000000), (20, 128, 1000000), (24, 64, 2000000), (24, 128, 1000000)]:
    mu, cov, w = mk(K, D, 5)
    from pypmc_amd.density.mixture import create_gaussian_mixture
    x = create_gaussian_mixture(mu, cov, w).propose(N, np.random.mtrand.RandomState(7), device=True)
    comps, inv, ln = gauss_set(mu, cov, w)
    print("D=%d K=%d N=%d: tiles per pass %d" % (D, K, N, be.lib.pmc_maha_gemm_tiles(N, K, D)), flush=True)
    ex, ge = both(lambda: be.tohost(be.logpdf(x, comps, want_scalars=True)["out"]))
    rep = be.maha_gemm_report(N, K, D) if be.lib.pmc_maha_gemm_tiles(N, K, D) else None
    print("   logpdf: refused %s  max|gemm-exact| %.3e" % (None if rep is None else (rep["refused"], rep["workgroups"]), np.abs(ex - ge).max()), flush=True)
    t_ex, t_ge = both(lambda: timeit(lambda: be.logpdf(x, comps, want_scalars=True)))
    print("   logpdf  exact %.3f ms  gemm %.3f ms  (%.2f -> %.2f ps/pair, %+.1f %%)" % (t_ex, t_ge, t_ex * 1e9 / (N * K), t_ge * 1e9 / (N * K), 100 * (t_ge / t_ex - 1)), flush=True)
    be.kernel_timing(True)
    for tol in (0.0, 5e-11):
        be.configure("maha_gemm_tolerance", tol)
        be.estep(x, comps, 1); be.kernel_timings()
        for _ in range(10): be.estep(x, comps, 1)
        kt = be.kernel_timings()
        print("   estep tol %g: %s" % (tol, {k: round(v["ms"] / max(v["calls"], 1), 4) for k, v in kt.items() if v["calls"]}), flush=True)
    be.kernel_timing(False)
    s_ex, s_ge = both(lambda: be.estep(x, comps, 1)["stats"].cpu().numpy().copy())
    print("   estep stats max rel diff %.3e" % np.max(np.abs(s_ex - s_ge) / (np.abs(s_ex) + 1e-300 + 1e-12 * np.abs(s_ex).max())), flush=True)
    del x
    be.release(); torch.cuda.empty_cache()
