"""The reference's examples/pmc_mpi.py (its data-parallel PMC example) the way this package shards it.

    python examples/pmc_torchrun.py [samples_per_step]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/pmc_torchrun.py [samples_per_step]

The reference (examples/pmc_mpi.py:63-128, pypmc/tools/parallel_sampler.py:58-71) lets every MPI process draw its share,
GATHERS all samples, weights and latent variables on the master, adapts the proposal there and BROADCASTS the new
proposal.  Here one process per GPU draws and weights its shard on its device, `gaussian_pmc` forms the shard's
sufficient statistics, ONE all-reduce (RCCL under torchrun) of K (1 + D + D(D+1)/2) + a few doubles joins them, and
every rank does the same K-sized update: no N-sized array ever leaves its GPU, there is no master and no broadcast.
Same target, same start proposal and the same calls as the reference script otherwise.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pypmc_amd as pypmc   # noqa: E402
from pypmc_amd import parallel   # noqa: E402

rank, world, _ = parallel.init_from_env()        # torchrun: one process per GPU; a plain python run is one rank

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

# every process its own random numbers (pmc_mpi.py:70-76 broadcasts a seed and adds the rank)
np.random.seed(4711 + rank)
lo, hi = parallel.shard_bounds(n_per_step)       # this rank's share of a step's samples
t0 = time.time()
for i in range(10):
    origin = sampler.run(hi - lo, trace_sort=True)
    samples = sampler.samples[-1]
    weights = sampler.weights[-1][:, 0]
    # the shard's statistics, all-reduced inside: every rank ends up with the same proposal
    pypmc.mix_adapt.pmc.gaussian_pmc(samples, sampler.proposal, weights, origin, mincount=20, rb=True, copy=False)
    # perplexity of ALL ranks' weights from three all-reduced sums (tools/convergence.py:6-39 on the joined array)
    s = parallel.all_reduce_sum(np.array([weights.sum(), (weights[weights > 0] * np.log(weights[weights > 0])).sum(),
                                          float(len(weights))]))
    perp = np.exp(-(s[1] / s[0] - np.log(s[0]))) / s[2]
    if rank == 0:
        print("step %d: perplexity of the %d samples of %d rank(s) %.4f" % (i, int(s[2]), world, perp))

# the adapted proposal is identical on every rank: check instead of trusting
flat = np.concatenate([sampler.proposal.weights] + [np.r_[c.mu, c.sigma.ravel()] for c in sampler.proposal.components])
rank0s = parallel.broadcast_from_rank0(flat)
differing = parallel.all_reduce_sum(np.array([float(np.count_nonzero(flat != rank0s))]))[0]
assert differing == 0, "ranks disagree about the proposal in %d numbers" % differing
if rank == 0:
    print("10 x %d samples on %d rank(s) in %.2f s" % (n_per_step, world, time.time() - t0))
    print('final   component weights:', sampler.proposal.weights)
    print('target  component weights:', component_weights)
    live = [k for k in range(len(sampler.proposal)) if sampler.proposal.weights[k] > 0]
    for k, m in zip(sorted(live, key=lambda k: -sampler.proposal.components[k].mu[0]), [mean0, mean1]):
        print('final mean of component %i:' % k, sampler.proposal.components[k].mu, ' target:', m)
if world > 1 or parallel.active():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
