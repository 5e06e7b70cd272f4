"""The reference's examples/pmc.py on the MI355X path: importance sampling of a bimodal 2-D
Gaussian target with a 3-component proposal that is adapted by Rao-Blackwellised PMC after every run.
Same API calls as the reference script; only the import root differs.

    python examples/pmc.py [samples_per_step]
"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pypmc_amd as pypmc   # noqa: E402

n_per_step = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 3

component_weights = np.array([0.3, 0.7])
mean0 = np.array([5.0, 0.01])
covariance0 = np.array([[0.01, 0.003], [0.003, 0.0025]])
mean1 = np.array([-4.0, 1.0])
covariance1 = np.array([[0.1, 0.], [0., 0.02]])
target_mixture = pypmc.density.mixture.create_gaussian_mixture([mean0, mean1], [covariance0, covariance1],
                                                               component_weights)
log_target = target_mixture.evaluate

initial_prop_means = [np.array([4.0, 0.0]), np.array([-5.0, 0.0]), np.array([0.0, 0.0])]
initial_proposal = pypmc.density.mixture.MixtureDensity(
    [pypmc.density.gauss.Gauss(m, np.eye(2)) for m in initial_prop_means])

sampler = pypmc.sampler.importance_sampling.ImportanceSampler(log_target, initial_proposal)

np.random.seed(42)
t0 = time.time()
for i in range(10):
    origin = sampler.run(n_per_step, trace_sort=True)
    samples = sampler.samples[-1]
    weights = sampler.weights[-1][:, 0]
    pypmc.mix_adapt.pmc.gaussian_pmc(samples, sampler.proposal, weights, origin, mincount=20, rb=True, copy=False)
    print("step %d: perplexity %.4f  ess %.4f" % (i, pypmc.tools.convergence.perp(weights),
                                                  pypmc.tools.convergence.ess(weights)))
print("10 x %d samples in %.2f s" % (n_per_step, time.time() - t0))
print('initial component weights:', initial_proposal.weights)
print('final   component weights:', sampler.proposal.weights)
print('target  component weights:', component_weights)
for k, m in enumerate([mean0, mean1]):
    print('final mean of component %i:' % k, sampler.proposal.components[k].mu, ' target:', m)
    print('final covariance of component %i:\n' % k, sampler.proposal.components[k].sigma)
