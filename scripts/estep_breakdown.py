#!/usr/bin/env python3
"""Where the wall time of GaussianInference.E_step goes at one GPU's share of 8 (BASELINE config 4: D = 20, K = 64,
N = 1.25e6): host stages by perf_counter (no synchronisation added), kernels by the library's own records.

    python scripts/estep_breakdown.py [--N 1250000] [--K 64] [--reps 200]
"""
import argparse
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import mk  # noqa: E402
import pypmc_amd.mix_adapt.variational as V  # noqa: E402
from pypmc_amd.backend import get_backend, HipBackend  # noqa: E402
from pypmc_amd.density.mixture import create_gaussian_mixture  # noqa: E402
from pypmc_amd import parallel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=1_250_000)
ap.add_argument("--K", type=int, default=64)
ap.add_argument("--D", type=int, default=20)
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()

acc = defaultdict(float)


def wrap(obj, name, label=None):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t0
    setattr(obj, name, timed)


be = get_backend()
mix = create_gaussian_mixture(*mk(args.K, args.D, 3))
np.random.seed(9)
x = mix.propose(args.N, device=True)
vb = V.GaussianInference(x, initial_guess=mix)
for _ in range(5):
    vb.E_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    vb.E_step()
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / args.reps * 1e3

for name in ("pack", "_means_pack", "zeros", "tohost", "_workspace", "_tilebuf", "estep"):
    wrap(be, name)
wrap(V, "convert_stats")
wrap(parallel, "all_reduce_sum")
wrap(vb, "_update_expectation_det_ln_lambda")
wrap(vb, "_update_expectation_ln_pi")
wrap(V, "ComponentSet")
be.kernel_timings()
be.kernel_timing(True)
t0 = time.perf_counter()
for _ in range(args.reps):
    vb.E_step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.reps * 1e3
be.kernel_timing(False)
kt = be.kernel_timings()
print("E_step D=%d K=%d N=%d: %.4f ms per call (%.4f with the timers on)" % (args.D, args.K, args.N, plain, wall))
ksum = 0.0
for k, v in kt.items():
    print("   kernel %-24s %.4f ms" % (k, v["ms"] / args.reps))
    ksum += v["ms"] / args.reps
print("   kernels together %.4f ms; wall - kernels %.4f ms" % (ksum, plain - ksum))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   host %-36s %.4f ms" % (k, v / args.reps * 1e3))
print("   (estep includes pack, _means_pack, zeros, _workspace, _tilebuf; tohost is where the host waits for the kernels)")
