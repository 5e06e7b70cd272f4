"""Worker for the world_size-2 gloo tests (spawned by tests/test_distributed_cpu.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, world, port, workdir):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        from pypmc_amd import parallel
        from pypmc_amd.density.mixture import create_gaussian_mixture
        from pypmc_amd.mix_adapt.variational import GaussianInference
        from pypmc_amd.mix_adapt.pmc import gaussian_pmc, PMC
        be = OracleBackend()
        z = np.load(os.path.join(workdir, "inputs.npz"))
        lo, hi = parallel.shard_bounds(len(z["data"]))
        assert parallel.world_size() == world and parallel.rank() == rank
        guess = create_gaussian_mixture(z["mu"], z["cov"], z["w"])
        # VB: every rank holds a shard, statistics are all-reduced, host update is replicated
        vb = GaussianInference(z["data"][lo:hi], initial_guess=guess, weights=z["sw"][lo:hi], backend=be)
        out = dict(vb_N=vb.N, vb_N_comp0=vb.N_comp.copy(), vb_bound0=vb.likelihood_bound())
        nit = vb.run(6, prune=1.)
        out.update(vb_nit=-1 if nit is None else nit, vb_m=vb.m, vb_W=vb.W, vb_alpha=vb.alpha,
                   vb_bound=vb.likelihood_bound(), vb_r_rows=len(vb.r))
        # PMC: sharded samples / weights / latent
        prop = create_gaussian_mixture(z["mu"], z["cov"], z["w"])
        prop._backend = be
        res = gaussian_pmc(z["data"][lo:hi], prop, weights=z["iw"][lo:hi], latent=z["latent"][lo:hi],
                           mincount=5, backend=be)
        out.update(pmc_w=res.weights, pmc_mu=np.array([c.mu for c in res.components]),
                   pmc_sigma=np.array([c.sigma for c in res.components]))
        drv = PMC(z["data"][lo:hi], prop, weights=z["iw"][lo:hi], backend=be)
        out["pmc_ll"] = drv.log_likelihood()
        np.savez(os.path.join(workdir, "rank%d.npz" % rank), **out)
    finally:
        dist.destroy_process_group()
