"""BASELINE config 5: the full PMC adapt loop (propose -> weight -> Rao-Blackwell update) with every
N-sized array resident on the GPU.  D=40, K=128 Gaussian proposal, K_t=4 Gaussian target.

    python examples/pmc_device_loop.py [N_per_iteration] [iterations] [evaluate-twice]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/pmc_device_loop.py ...

Per iteration: counts on the host (rng.multinomial), samples + origins on the device (pmc_propose),
log P and log q + importance weights + perplexity sums in one pass that also leaves the Rao-Blackwell
responsibilities of the update behind (pmc_importance_weights_emit), the sufficient statistics
(pmc_estep_from_u), one all-reduce when several ranks run, K-sized update on the host.
With a third argument the update evaluates the components again, as the reference does
(pmc_responsibilities + the statistics): same result to rounding, 1.5x the time at this shape.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pypmc_amd as pypmc   # noqa: E402
from pypmc_amd import parallel   # noqa: E402
from pypmc_amd.tools.convergence import perp_from_sums   # noqa: E402

import torch   # noqa: E402

rank, world, _ = parallel.init_from_env()        # torchrun: one process per GPU, RCCL; no-op for a single process

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reuse = len(sys.argv) <= 3
K, D, K_T = 128, 40, 4


def mk(K, D, seed, spread=3.0):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, spread, size=(K, D))
    cov = np.empty((K, D, D))
    for k in range(K):
        A = rs.normal(0, 1, size=(D, D))
        cov[k] = A.dot(A.T) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K)
    return mu, cov, w / w.sum()


tmu, tcov, tw = mk(K_T, D, 11, spread=1.0)
target = pypmc.density.mixture.create_gaussian_mixture(tmu, tcov, tw)
# proposal: K components, 32 per target mode, slightly displaced and 1.5 x too wide (the state an
# adapt loop is in after its first rough iterations)
rs = np.random.RandomState(5)
which = np.arange(K) % K_T
pmu = tmu[which] + rs.normal(0, 0.15, (K, D))
proposal = pypmc.density.mixture.create_gaussian_mixture(pmu, 1.5 * tcov[which])
np.random.seed(100 + parallel.rank())
sampler = pypmc.sampler.importance_sampling.ImportanceSampler(target.evaluate, proposal)

for it in range(iters):
    torch.cuda.synchronize()
    t0 = time.time()
    run = sampler.run_device(N, trace_sort=True, prepare_update=reuse)
    torch.cuda.synchronize()
    t1 = time.time()
    pypmc.mix_adapt.pmc.gaussian_pmc(run["samples"], sampler.proposal, run["weights"], run["origin"],
                                     mincount=0, rb=True, copy=False, mahalanobis=run["mahalanobis"],
                                     responsibilities=run["responsibilities"])
    torch.cuda.synchronize()
    t2 = time.time()
    S, L, Q = run["weight_sums"]
    if parallel.rank() == 0:
        print("iteration %d: propose+weight %.3f s, update %.3f s, %.2e samples/s/rank, perplexity %.4f, live K %d"
              % (it, t1 - t0, t2 - t1, N / (t2 - t0), perp_from_sums(S, L, N),
                 int((sampler.proposal.weights > 0).sum())))
if world > 1:
    import torch.distributed as dist
    dist.destroy_process_group()
