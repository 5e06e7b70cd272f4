#!/usr/bin/env python3
"""Config 5's PMC iteration (D = 40, K = 128, 1.25e7 samples) beyond its first iteration (verdict r5 #2):
all components alive | a fifth of them pruned (weight 0, left in the mixture: pmc.pyx:109-117) | Student-t proposal.
Wall ms per iteration and the library's kernel times.

    python scripts/cfg5_variants.py [N]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import mk  # noqa: E402
from pypmc_amd.backend import get_backend  # noqa: E402
from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture  # noqa: E402
from pypmc_amd.sampler.importance_sampling import ImportanceSampler  # noqa: E402
from pypmc_amd.mix_adapt.pmc import gaussian_pmc, student_t_pmc  # noqa: E402

be = get_backend()
D5, K5, KT5 = 40, 128, 4
N5 = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
rs = np.random.RandomState(5)
tmu, tcov, tw = mk(KT5, D5, 11)
tmu /= 3.0
target = create_gaussian_mixture(tmu, tcov, tw)
which = np.arange(K5) % KT5
means, covs = tmu[which] + rs.normal(0, 0.15, (K5, D5)), 1.5 * tcov[which]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.3:
        fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    be.kernel_timings()
    be.kernel_timing(True)
    fn()
    torch.cuda.synchronize()
    kern = be.kernel_timings()
    be.kernel_timing(False)
    return float(np.median(ts)) * 1e3, {k: round(v["ms"], 3) for k, v in kern.items()}


def run_case(label, proposal, update, **kw):
    np.random.seed(100)
    sampler = ImportanceSampler(target.evaluate, proposal)
    w0 = proposal.weights.copy()

    def iteration():
        sampler.proposal.weights[:] = w0               # (every timed iteration starts from the same mixture)
        run = sampler.run_device(N5, trace_sort=True, prepare_update=True)
        update(run["samples"], sampler.proposal, run["weights"], run["origin"], mincount=0, rb=True, copy=True,
               mahalanobis=run["mahalanobis"], responsibilities=run["responsibilities"], **kw)
    ms, kern = timed(iteration)
    print("%-44s %8.2f ms   %s" % (label, ms, kern), flush=True)
    return ms


base = run_case("Gauss, all alive", create_gaussian_mixture(means, covs), gaussian_pmc)
w = np.ones(K5)
w[rs.choice(K5, K5 // 5, replace=False)] = 0.
run_case("Gauss, a fifth pruned (26 of 128)", create_gaussian_mixture(means, covs, w / w.sum()), gaussian_pmc)
run_case("Gauss, the same 102 live components only", create_gaussian_mixture(means[w > 0], covs[w > 0]), gaussian_pmc)
for K in (100, 70, 40):
    run_case("Gauss, K = %d" % K, create_gaussian_mixture(means[:K], covs[:K]), gaussian_pmc)
run_case("Student-t nu = 8, all alive", create_t_mixture(means, covs, np.full(K5, 8.)), student_t_pmc)
