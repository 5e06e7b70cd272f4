#!/usr/bin/env python3
"""The reference's working regime (examples/pmc.py: 1e3 samples per step): one PMC step = sampler.run(n) + gaussian_pmc through
the public API with host arrays -- microseconds per step and where they go (cProfile)."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pypmc_amd as pypmc
from test_gpu_kernels import mk
D, K, KT, n = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 3, 2, 1000))]
tmu, tcov, tw = mk(KT, D, 11)
target = pypmc.density.mixture.create_gaussian_mixture(tmu / 3.0, tcov, tw)
rs = np.random.RandomState(5)
which = np.arange(K) % KT
proposal = pypmc.density.mixture.create_gaussian_mixture(tmu[which] / 3.0 + rs.normal(0, 0.3, (K, D)), 1.5 * tcov[which])
sampler = pypmc.sampler.importance_sampling.ImportanceSampler(target.evaluate, proposal)
np.random.seed(42)


def step():
    origin = sampler.run(n, trace_sort=True)
    samples = sampler.samples[-1]
    weights = sampler.weights[-1][:, 0]
    pypmc.mix_adapt.pmc.gaussian_pmc(samples, sampler.proposal, weights, origin, mincount=0, rb=True, copy=False)


for _ in range(20):
    step()
sampler.clear()
t0 = time.perf_counter()
for _ in range(100):
    step()
print("D=%d K=%d n=%d: %.1f us per step (run + gaussian_pmc)" % (D, K, n, (time.perf_counter() - t0) / 100 * 1e6))
sampler.clear()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
