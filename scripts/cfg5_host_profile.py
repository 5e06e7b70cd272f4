#!/usr/bin/env python3
"""Host side of one PMC iteration of BASELINE config 5 (D = 40, K = 128): cProfile at a small N, where the kernels
are short and what remains is the K-sized work between them.

    python scripts/cfg5_host_profile.py [N]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import mk  # noqa: E402
from pypmc_amd.density.mixture import create_gaussian_mixture  # noqa: E402
from pypmc_amd.sampler.importance_sampling import ImportanceSampler  # noqa: E402
from pypmc_amd.mix_adapt.pmc import gaussian_pmc  # noqa: E402

N5 = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000
D5, K5, KT5 = 40, 128, 4
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
for _ in range(20):
    iteration()
torch.cuda.synchronize()
print("N = %d: %.3f ms per iteration" % (N5, (time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    iteration()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
