"""Worker and shared case of the world_size-2 tests (spawned by tests/test_distributed_cpu.py with the
oracle-backed checker and by tests/test_gpu_distributed.py with the HipBackend on one GPU)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_inputs(seed=5, K=3, D=4, N=601):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, 3, (K, D))
    cov = np.array([np.eye(D) * (0.5 + k) for k in range(K)])
    w = np.array([0.2, 0.5, 0.3])[:K]
    w = w / w.sum()
    latent = np.sort(rs.choice(K, size=N, p=w))
    data = mu[latent] + np.einsum('nij,nj->ni', np.linalg.cholesky(cov)[latent], rs.normal(size=(N, D)))
    perm = rs.permutation(N)
    return dict(data=data[perm], mu=mu, cov=cov, w=w, sw=rs.uniform(0.5, 1.5, N), iw=rs.uniform(0.2, 2.0, N),
                latent=latent[perm])


def case(be, z, lo, hi):
    """Everything a rank computes on its shard [lo, hi) -- or a single process on all rows.
    Returns K-sized results only (they must agree between the two)."""
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    from pypmc_amd.mix_adapt.variational import GaussianInference
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc, student_t_pmc, PMC
    data, sw, iw, latent = z["data"][lo:hi], z["sw"][lo:hi], z["iw"][lo:hi], z["latent"][lo:hi]
    K = len(z["w"])
    out = {}
    # VB from a mixture guess: every rank holds a shard, statistics are all-reduced, host update replicated
    guess = create_gaussian_mixture(z["mu"], z["cov"], z["w"])
    vb = GaussianInference(data, initial_guess=guess, weights=sw, backend=be)
    out.update(vb_N=vb.N, vb_N_comp0=vb.N_comp.copy(), vb_bound0=vb.likelihood_bound())
    nit = vb.run(6, prune=1.)
    out.update(vb_nit=-1 if nit is None else nit, vb_m=vb.m, vb_W=vb.W, vb_alpha=vb.alpha,
               vb_bound=vb.likelihood_bound(), vb_r_rows=len(vb.r))
    # VB with the start means taken from the data: the first K rows of the GLOBAL array / K random ones
    vf = GaussianInference(data, components=K + 1, initial_guess="first", backend=be)
    out.update(vbf_m0=vf.m.copy(), vbf_N_comp0=vf.N_comp.copy())
    vf.run(4, prune=0.)
    out.update(vbf_m=vf.m, vbf_W=vf.W, vbf_bound=vf.likelihood_bound())
    np.random.seed(1234)                         # every process seeds alike; rank 0's draw is used
    vr = GaussianInference(data, components=K, initial_guess="random", backend=be)
    out.update(vbr_m0=vr.m.copy(), vbr_N_comp0=vr.N_comp.copy())
    # Gaussian PMC: sharded samples / weights / latent, mincount pruning on the GLOBAL histogram
    prop = create_gaussian_mixture(z["mu"], z["cov"], z["w"])
    prop._backend = be
    res = gaussian_pmc(data, prop, weights=iw, latent=latent, mincount=5, backend=be)
    out.update(pmc_w=res.weights, pmc_mu=np.array([c.mu for c in res.components]),
               pmc_sigma=np.array([c.sigma for c in res.components]))
    res = gaussian_pmc(data, prop, weights=iw, latent=latent, rb=False, backend=be)
    out.update(pmcl_w=res.weights, pmcl_mu=np.array([c.mu for c in res.components]))
    drv = PMC(data, prop, weights=iw, backend=be)
    out["pmc_ll"] = drv.log_likelihood()
    out["pmc_run"] = -1 if drv.run(3) is None else 1
    out["pmc_run_mu"] = np.array([c.mu for c in drv.density.components])
    # Student-t PMC incl. the degree-of-freedom condition
    tprop = create_t_mixture(z["mu"], z["cov"], np.full(K, 5.), z["w"])
    tprop._backend = be
    res = student_t_pmc(data, tprop, weights=iw, backend=be)
    out.update(tpmc_w=res.weights, tpmc_mu=np.array([c.mu for c in res.components]),
               tpmc_sigma=np.array([c.sigma for c in res.components]),
               tpmc_dof=np.array([c.dof for c in res.components]))
    return out


# results that are bitwise identical on all ranks (replicated K-sized update, no broadcast)
REPLICATED = ("vb_m", "vb_W", "vbf_m0", "vbf_m", "vbf_W", "vbr_m0", "pmc_mu", "pmc_sigma", "pmc_w", "pmcl_mu",
              "pmc_run_mu", "tpmc_mu", "tpmc_sigma", "tpmc_dof")


def collectives(be=None):
    """The plumbing of pypmc_amd.parallel on whatever process group is up: numpy and device buffers through
    all_reduce_sum, scalars, shard offsets, rows of the global sample array, rank 0's vector."""
    from pypmc_amd import parallel
    world, rank = parallel.world_size(), parallel.rank()
    out = {}
    a = np.arange(5, dtype=np.float64) + rank
    out["ar_numpy"] = parallel.all_reduce_sum(a.copy())
    out["ar_numpy_expected"] = world * np.arange(5, dtype=np.float64) + sum(range(world))
    if be is not None and be.name == "hip":
        t = be.asdevice(a.copy())
        out["ar_device"] = be.tohost(parallel.all_reduce_sum(t))
    out["scalars"] = np.array(parallel.all_reduce_scalars(1.0, 2.5 * (rank + 1)))
    out["scalars_expected"] = np.array([float(world), 2.5 * sum(r + 1 for r in range(world))])
    n_local = 7 + rank
    off, total = parallel.shard_offset(n_local)
    out["offset"] = np.array([off, total])
    out["offset_expected"] = np.array([sum(7 + r for r in range(rank)), sum(7 + r for r in range(world))])
    rows = np.arange(n_local * 3, dtype=np.float64).reshape(n_local, 3) + 1000 * rank
    out["rows"] = parallel.global_rows([0, total - 1, 3], n_local, lambda loc: rows[loc], 3)
    out["bcast"] = parallel.broadcast_from_rank0(np.array([3.0, 4.0]) + rank)
    return out


def _rendezvous(workdir):
    """a file in the run's own temporary directory: no TCP port to pick and lose to another process before it is bound
    (a full-suite run once died of EADDRINUSE on a port that had just been probed free)"""
    return "file://" + os.path.join(workdir, "rendezvous")


def run(rank, world, port, workdir, backend_kind="oracle", pg_backend="gloo"):
    import torch.distributed as dist
    if backend_kind == "hip":
        import torch
        torch.cuda.set_device(0)                     # both ranks share the one GPU of the box
    if pg_backend == "nccl":                         # RCCL: one rank per device (a single rank on a one-GPU box)
        import torch
        dist.init_process_group("nccl", init_method=_rendezvous(workdir), rank=rank, world_size=world,
                                device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", init_method=_rendezvous(workdir), rank=rank, world_size=world)
    try:
        from pypmc_amd import parallel
        if os.environ.get("PMC_NATIVE_COLLECTIVE", "0") not in ("", "0"):
            parallel.enable_native_collective(0)
        if os.environ.get("PMC_P2P_COLLECTIVE", "0") not in ("", "0"):
            parallel.enable_p2p_collective(device=0)
        if backend_kind == "hip":
            from pypmc_amd.backend import HipBackend
            be = HipBackend(0)
        else:
            from oracle_backend import OracleBackend
            be = OracleBackend()
        z = dict(np.load(os.path.join(workdir, "inputs.npz")))
        lo, hi = parallel.shard_bounds(len(z["data"]))
        assert parallel.world_size() == world and parallel.rank() == rank and parallel.active()
        out = case(be, z, lo, hi)
        out["backend"] = np.array(be.name)
        out["pg_backend"] = np.array(dist.get_backend())
        out["collective"] = np.array(str(parallel.collective_name()))
        out.update({"coll_" + k: v for k, v in collectives(be).items()})
        np.savez(os.path.join(workdir, "rank%d.npz" % rank), **out)
    finally:
        from pypmc_amd import parallel as _p
        _p.disable_native_collective()
        _p.disable_p2p_collective()
        dist.destroy_process_group()
