#!/usr/bin/env python3
"""Where config 5's host time goes: cProfile of the PMC iteration (propose -> weights -> update) at one GPU's share of 8."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_kernels import mk
from pypmc_amd.density.mixture import create_gaussian_mixture
from pypmc_amd.sampler.importance_sampling import ImportanceSampler
from pypmc_amd.mix_adapt.pmc import gaussian_pmc
D5, K5, KT5, N5 = 40, 128, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
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


for _ in range(3):
    iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    iteration()
torch.cuda.synchronize()
print("ms per iteration", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    iteration()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("PROFILE_SORT", "tottime")).print_stats(int(os.environ.get("PROFILE_ROWS", "28")))
