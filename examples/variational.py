"""Variational-Bayes mixture fit of importance samples on the GPU (cf. the reference's
examples/variational.py): N weighted samples from a 3-mode target, K = 8 start components, pruning.

    python examples/variational.py [N]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/variational.py [N]
With several ranks every rank holds N/world samples; the only communication is the all-reduce of the
K-sized statistics per update.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pypmc_amd as pypmc   # noqa: E402
from pypmc_amd import parallel   # noqa: E402

rank, world, _ = parallel.init_from_env()        # torchrun: one process per GPU, RCCL; no-op for a single process

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
D = 5
rs = np.random.RandomState(1)
means = rs.normal(0, 4, (3, D))
covs = np.array([np.eye(D) * s for s in (0.5, 1.0, 2.0)])
target = pypmc.density.mixture.create_gaussian_mixture(means, covs, [0.5, 0.3, 0.2])

lo, hi = parallel.shard_bounds(N)
np.random.seed(10 + parallel.rank())
data = target.propose(hi - lo)

t0 = time.time()
vb = pypmc.mix_adapt.variational.GaussianInference(data, components=8, alpha0=1e-3,
                                                   m=rs.normal(0, 4, (8, D)))
nit = vb.run(100, prune=0.5 * N / 100, rel_tol=1e-8)
if parallel.rank() == 0:
    print("N = %d on %d rank(s): converged after %s updates in %.2f s, K = %d" % (N, world, nit, time.time() - t0, vb.K))
    mix = vb.make_mixture()
    print("weights:", np.round(mix.weights, 4))
    for c in mix.components:
        print("mean:", np.round(c.mu, 3))
if world > 1:
    import torch.distributed as dist
    dist.destroy_process_group()
