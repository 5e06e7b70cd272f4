#!/usr/bin/env python3
"""Where the host time of a PMC iteration goes (BASELINE config 5's shape, K = 128, D = 40) -- the iteration of bench.py's
cfg5 at a sample count small enough that the kernels vanish, under cProfile (GPU box)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import mk
from pypmc_amd.density.mixture import create_gaussian_mixture
from pypmc_amd.mix_adapt.pmc import gaussian_pmc
from pypmc_amd.sampler.importance_sampling import ImportanceSampler

D5, K5, KT5 = 40, 128, 4
N5 = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rs = np.random.RandomState(5)
tmu, tcov, tw = mk(KT5, D5, 11)
tmu /= 3.0
target = create_gaussian_mixture(tmu, tcov, tw)
which = np.arange(K5) % KT5
proposal = create_gaussian_mixture(tmu[which] + rs.normal(0, 0.15, (K5, D5)), 1.5 * tcov[which])
np.random.seed(100)
sampler = ImportanceSampler(target.evaluate, proposal)


def iteration():
    run = sampler.run_device(N5, trace_sort=True, prepare_update=True)
    gaussian_pmc(run["samples"], sampler.proposal, run["weights"], run["origin"], mincount=0, rb=True,
                 copy=False, mahalanobis=run["mahalanobis"], responsibilities=run["responsibilities"])


for _ in range(5):
    iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    iteration()
torch.cuda.synchronize()
print("iteration at N = %d: %.3f ms" % (N5, (time.perf_counter() - t0) / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    iteration()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
