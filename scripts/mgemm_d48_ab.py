"""k_mgemm<48>: four component tiles per pass on two wavefronts per SIMD (chunks of 8 steps) against round 4's two tiles
on one wavefront (variant library built with -DPMC_MG_D48_TWO_TILES), in ONE process on the same samples.

    PMC_VARIANT=mg48two PMC_VARIANT_UNITS=pmc_mgemm_d48_p0 PMC_EXTRA_FLAGS=-DPMC_MG_D48_TWO_TILES python -m pypmc_amd.build
    python scripts/mgemm_d48_ab.py
"""
import os, subprocess, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) == 1:
    for name, lib in (("four tiles (product)", ""), ("two tiles (round 4)", os.path.join(root, "pypmc_amd/lib/libpmc_hip_mg48two.so"))):
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
for D, K, N in [(48, 128, 1000000), (48, 64, 2000000), (44, 120, 1000000), (48, 32, 2000000)]:
    mu, cov, w = mk(K, D, 5)
    x = create_gaussian_mixture(mu, cov, w).propose(N, np.random.mtrand.RandomState(7), device=True)
    comps = gauss_set(mu, cov, w)[0]
    be.configure("maha_gemm_tolerance", 0.0); t_ex = timeit(lambda: be.logpdf(x, comps, want_scalars=True))
    be.configure("maha_gemm_tolerance", 5e-11); t_ge = timeit(lambda: be.logpdf(x, comps, want_scalars=True))
    t_es = timeit(lambda: be.estep(x, comps, 1))
    print("  D=%d K=%d: tiles per pass %d; logpdf exact %.3f ms, gemm %.3f ms (%.2f ps/pair, %+.1f %%); E-step %.3f ms"
          % (D, K, be.lib.pmc_maha_gemm_tiles(N, K, D), t_ex, t_ge, t_ge * 1e9 / (N * K), 100 * (t_ge / t_ex - 1), t_es), flush=True)
    del x; be.release(); torch.cuda.empty_cache()
