#!/usr/bin/env python3
"""Kernel time of one mixture log-pdf call on the matrix-product path (D = 40, K = 128, N = 4e6 by default), from the
library's own records; used by scripts/mgemm_ab.sh with PMC_HIP_LIBRARY pointing at a variant of the library.

    python scripts/mgemm_time.py [D K N]
"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from pypmc_amd.backend import HipBackend  # noqa: E402
from pypmc_amd.density.mixture import create_gaussian_mixture  # noqa: E402
from test_gpu_kernels import mk, gauss_set  # noqa: E402

be = HipBackend()
D, K, N = (int(float(a)) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (40, 128, 4000000)
mu, cov, w = mk(K, D, 5)
x = create_gaussian_mixture(mu, cov, w).propose(N, np.random.mtrand.RandomState(7), device=True)
comps = gauss_set(mu, cov, w)[0]
for _ in range(3):
    be.logpdf(x, comps, want_scalars=True)
torch.cuda.synchronize()
be.kernel_timings()
be.kernel_timing(True)
for _ in range(10):
    be.logpdf(x, comps, want_scalars=True)
torch.cuda.synchronize()
be.kernel_timing(False)
t = be.kernel_timings()
print("%-24s D=%d K=%d N=%d " % (os.path.basename(os.environ.get("PMC_HIP_LIBRARY", "product")), D, K, N),
      {k: round(v["ms"] / v["calls"], 4) for k, v in t.items()},
      "ps/pair %.2f" % (sum(v["ms"] / v["calls"] for v in t.values()) * 1e9 / (N * K)))
