import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from pypmc_amd.backend import HipBackend
from test_gpu_kernels import mk, gauss_set
be = HipBackend()
for D, K in ((40, 128), (64, 128), (32, 32)):
    mu, cov, w = mk(K, D, 5)
    comps = gauss_set(mu, cov, w)[0]
    x = be.asdevice(np.random.RandomState(1).normal(size=(4096, D)) * 3)
    for tol in (0.0, 5e-11):
        be.configure("maha_gemm_tolerance", tol)
        for _ in range(20): be.logpdf(x, comps, want_scalars=True)
        torch.cuda.synchronize()
