"""The reference's PMC example (examples/pmc.py: a bimodal 2-D Gaussian target, a three-component Gaussian proposal adapted
by ten PMC updates) and BASELINE config 5's loop, on SEVERAL GPUs from this ONE Python process -- the way pypmc scripts run
(no mpirun, no torchrun):

    python examples/pmc_devices.py                      # every visible GPU
    python examples/pmc_devices.py 0,1,2,3              # these devices
    python examples/pmc_devices.py 0,0,0,0 2000000      # four virtual shards on one GPU, 2e6 samples per iteration (config 5)

The sampler draws the component counts on the host (``rng.multinomial``: counts and origins bit-exact), generates contiguous
blocks of the samples on the devices, weights them there against the mixture target, and the Rao-Blackwellised update reduces
them where they are: every device its block, the K-sized statistics added in device order on the first device
(pypmc_amd.devices.DeviceGroup over the C ABI's pmc_init_devices).  Nothing N-sized returns to the host.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pypmc_amd as pypmc   # noqa: E402
from pypmc_amd.devices import DeviceGroup   # noqa: E402
from pypmc_amd.density.gauss import Gauss   # noqa: E402
from pypmc_amd.density.mixture import MixtureDensity, create_gaussian_mixture   # noqa: E402
from pypmc_amd.mix_adapt.pmc import gaussian_pmc   # noqa: E402
from pypmc_amd.sampler.importance_sampling import ImportanceSampler   # noqa: E402
from pypmc_amd.tools.convergence import perp_from_sums   # noqa: E402

devices = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else None
big = int(sys.argv[2]) if len(sys.argv) > 2 else 0
np.random.seed(0)

with DeviceGroup(devices) as group:
    print("devices:", group.devices)
    if not big:
        # ---- examples/pmc.py:15-73
        target = create_gaussian_mixture([np.array([5., 0.01]), np.array([-4., 1.])],
                                         [np.array([[.01, .003], [.003, .0025]]), np.array([[.1, 0.], [0., .02]])],
                                         [.3, .7])
        proposal = MixtureDensity([Gauss(m, np.eye(2)) for m in ([-5., 0.], [0., 0.], [5., 0.])])
        sampler = ImportanceSampler(target.evaluate, proposal, devices=group)
        for step in range(10):
            sampler.run(10 ** 3, trace_sort=True)
            run = sampler.last_run
            gaussian_pmc(run, sampler.proposal, run.weights, 'origin', mincount=20, rb=True, copy=False)
            print("step %2d  perplexity %.3f  weights %s" % (step, perp_from_sums(sampler.last_weight_sums[0], sampler.last_weight_sums[1], 10 ** 3),
                                                               np.round(sampler.proposal.weights, 3)))
    else:
        # ---- BASELINE config 5: D = 40, K = 128 proposal, K_t = 4 target, `big` samples per iteration over the devices
        def mk(K, D, seed, spread):
            rs = np.random.RandomState(seed)
            mu = rs.normal(0, spread, size=(K, D))
            cov = np.array([a.dot(a.T) / D + 0.5 * np.eye(D) for a in rs.normal(0, 1, size=(K, D, D))])
            w = rs.uniform(0.5, 1.5, size=K)
            return mu, cov, w / w.sum()
        K, D, KT = 128, 40, 4
        tmu, tcov, tw = mk(KT, D, 11, 1.0)
        target = create_gaussian_mixture(tmu, tcov, tw)
        which = np.arange(K) % KT
        proposal = create_gaussian_mixture(tmu[which] + np.random.RandomState(5).normal(0, 0.15, (K, D)), 1.5 * tcov[which])
        sampler = ImportanceSampler(target.evaluate, proposal, devices=group)
        for it in range(5):
            t0 = time.time()
            sampler._run_group(big, trace_sort=False, store=False)      # (no host copies of the samples: config 5's loop)
            run = sampler.last_run
            gaussian_pmc(run, sampler.proposal, run.weights, copy=False)
            dt = time.time() - t0
            print("iteration %d: %.1f ms, %.3e samples/s, perplexity %.4f" %
                  (it, dt * 1e3, big / dt, perp_from_sums(sampler.last_weight_sums[0], sampler.last_weight_sums[1], big)))
