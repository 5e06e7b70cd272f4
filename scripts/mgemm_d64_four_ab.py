"""k_mgemm<64>: four component tiles per pass on two wavefronts per SIMD (theta in chunks of 4 steps: 154 KB of LDS) -- the
product -- against two tiles on one wavefront (chunks of 8), variant library built with -DPMC_MG_D64_TWO_TILES, same samples.

    PMC_VARIANT=mg64two PMC_VARIANT_UNITS=pmc_mgemm_d64_p0 PMC_EXTRA_FLAGS=-DPMC_MG_D64_TWO_TILES python -m pypmc_amd.build
    python scripts/mgemm_d64_four_ab.py
"""
import os, subprocess, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) == 1:
    for name, lib in (("four tiles, chunks of 4 steps (product)", ""), ("two tiles, chunks of 8 (until round 5)", os.path.join(root, "pypmc_amd/lib/libpmc_hip_mg64two.so"))):
        env = dict(os.environ)
        if lib: env["PMC_HIP_LIBRARY"] = lib
        print(name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=env)
    sys.exit(0)
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
for D, K, N in [(64, 128, 1000000), (64, 64, 1000000), (56, 64, 1000000)]:
    mu, cov, w = mk(K, D, 5)
    x = create_gaussian_mixture(mu, cov, w).propose(N, np.random.mtrand.RandomState(7), device=True)
    comps = gauss_set(mu, cov, w)[0]
    be.configure("maha_gemm_tolerance", 0.0); ex = be.tohost(be.logpdf(x, comps)["out"]); t_ex = timeit(lambda: be.logpdf(x, comps, want_scalars=True))
    be.configure("maha_gemm_tolerance", 5e-11); ge = be.tohost(be.logpdf(x, comps)["out"]); t_ge = timeit(lambda: be.logpdf(x, comps, want_scalars=True))
    print("  D=%d K=%d: tiles per pass %d; max|gemm-exact| %.2e; logpdf exact %.3f ms, gemm %.3f ms (%.2f ps/pair, %+.1f %%)"
          % (D, K, be.lib.pmc_maha_gemm_tiles(N, K, D), np.abs(ex - ge).max(), t_ex, t_ge, t_ge * 1e9 / (N * K), 100 * (t_ge / t_ex - 1)), flush=True)
    del x; be.release(); torch.cuda.empty_cache()
