#!/usr/bin/env python3
"""The VB E-step of the headline (N = 1e7, D = 20, K = 32 and 64) in a loop: the library's own per-kernel times.
Used by scripts/estep_traffic_ab.sh with PMC_HIP_LIBRARY pointing at an A/B variant of the library.

    python scripts/estep_loop.py [--reps 20] [--K 32,64] [--N 10000000]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import mk, vb_params  # noqa: E402
from pypmc_amd.backend import HipBackend, ComponentSet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--K", default="32,64")
ap.add_argument("--N", type=int, default=10_000_000)
ap.add_argument("--D", type=int, default=20)
args = ap.parse_args()
be = HipBackend(0)
D, N = args.D, args.N
for K in [int(k) for k in args.K.split(",")]:
    mu, cov, w = mk(K, D, 3)
    W, beta, nu, ln_pi, ln_lambda = vb_params(mu, cov, w, N)
    post = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, D, dtype=torch.float64, device='cuda', generator=g) * 1.2
    x += torch.tensor(mu, device='cuda')[torch.randint(0, K, (N,), device='cuda', generator=g)]
    stats = be.zeros(be.stats_len(K, D))
    pack = be.pack(post)
    for _ in range(5):
        be.estep(x, post, 0, pack=pack, out=stats)
    torch.cuda.synchronize()
    be.kernel_timings()
    be.kernel_timing(True)
    for _ in range(args.reps):
        be.estep(x, post, 0, pack=pack, out=stats)
    torch.cuda.synchronize()
    be.kernel_timing(False)
    t = be.kernel_timings()
    print("lib %s  D=%d K=%d N=%d: " % (os.path.basename(os.environ.get("PMC_HIP_LIBRARY", "libpmc_hip.so")), D, K, N) +
          "  ".join("%s %.4f ms" % (k, v["ms"] / v["calls"]) for k, v in t.items()) +
          "   pair %.4f ms" % sum(v["ms"] / v["calls"] for k, v in t.items() if k in ("k_resp", "k_stats")), flush=True)
    del x
