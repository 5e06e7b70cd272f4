#!/usr/bin/env python3
"""Call latencies of the front-end at the sizes pypmc's own examples use (host arrays in, host arrays out)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mk(K, D, seed):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, 3, size=(K, D))
    cov = np.array([np.eye(D) * (0.5 + k) for k in range(K)])
    w = rs.uniform(0.5, 1.5, size=K)
    return mu, cov, w / w.sum()


def timeit(fn, reps=30):
    fn()
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6


def main():
    import torch
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    from pypmc_amd.mix_adapt.variational import GaussianInference
    from pypmc_amd.sampler.importance_sampling import ImportanceSampler
    for N, D, K in ((1000, 2, 3), (10000, 10, 5), (100000, 10, 5), (100000, 20, 16)):
        mu, cov, w = mk(K, D, 1)
        mix = create_gaussian_mixture(mu, cov, w)
        x = mix.propose(N, np.random.RandomState(1))
        iw = np.random.RandomState(2).uniform(0.5, 1.5, N)
        t_eval = timeit(lambda: mix.multi_evaluate(x))
        t_pmc = timeit(lambda: gaussian_pmc(x, mix, iw), reps=10)
        vb = GaussianInference(x, initial_guess=mix)
        t_vb = timeit(vb.update, reps=10)
        sampler = ImportanceSampler(mix.evaluate, mix, rng=np.random.RandomState(3))
        t_is = timeit(lambda: (sampler.clear(), sampler.run(N)), reps=10)
        print("N=%-7d D=%-3d K=%-3d multi_evaluate %8.0f us   gaussian_pmc %8.0f us   VB update %8.0f us   "
              "ImportanceSampler.run %8.0f us" % (N, D, K, t_eval, t_pmc, t_vb, t_is), flush=True)


if __name__ == "__main__":
    main()
