"""N > 1 path on CPU: two gloo ranks, samples sharded by rank, one all-reduce of the statistics
vector per update -- must reproduce the single-process result on every rank."""
import os
import socket
import tempfile

import numpy as np

import dist_worker
from oracle_backend import OracleBackend
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_torchrun(nproc, script_args, env, timeout=900, attempts=4):
    """`python -m torch.distributed.run --nproc-per-node nproc ...` on a port probed free, again on another one if the
    port was taken between the probe and torchrun's bind (EADDRINUSE) -- other jobs share the node's port space"""
    import subprocess
    import sys
    for attempt in range(attempts):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + list(script_args)
        r = subprocess.run(cmd, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        if r.returncode != 0 and ("EADDRINUSE" in r.stderr or "address already in use" in r.stderr.lower()) \
                and attempt + 1 < attempts:
            continue
        return r
    return r


def test_shard_bounds():
    from pypmc_amd.parallel import shard_bounds
    for N, world in ((10, 3), (7, 8), (1000, 8), (0, 2)):
        cuts = [shard_bounds(N, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == N
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


def check_two_ranks(single, ranks, N, rtol):
    assert sum(int(r["vb_r_rows"]) for r in ranks) == N        # r stays sharded
    for got in ranks:
        for key, val in single.items():
            if key == "vb_r_rows":
                continue
            np.testing.assert_allclose(got[key], val, rtol=rtol, atol=1e-12, err_msg=key)
    # the start means are rows of the data: exact, whatever the shard they came from
    np.testing.assert_array_equal(ranks[0]["vbf_m0"], single["vbf_m0"])
    np.testing.assert_array_equal(ranks[0]["vbr_m0"], single["vbr_m0"])
    # all ranks hold bitwise identical parameters (replicated update, no broadcast)
    for key in dist_worker.REPLICATED:
        np.testing.assert_array_equal(ranks[0][key], ranks[1][key], err_msg=key)


def spawn_ranks(z, backend_kind, world=2, pg_backend="gloo"):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "inputs.npz"), **z)
        mp.spawn(dist_worker.run, args=(world, _free_port(), tmp, backend_kind, pg_backend), nprocs=world, join=True)
        return [dict(np.load(os.path.join(tmp, "rank%d.npz" % r))) for r in range(world)]


def spawn_two_ranks(z, backend_kind):
    return spawn_ranks(z, backend_kind, 2)


def check_collectives(ranks):
    """pypmc_amd.parallel's plumbing as each rank saw it (dist_worker.collectives)"""
    for r, got in enumerate(ranks):
        np.testing.assert_array_equal(got["coll_ar_numpy"], got["coll_ar_numpy_expected"])
        if "coll_ar_device" in got:
            np.testing.assert_array_equal(got["coll_ar_device"], got["coll_ar_numpy_expected"])
        np.testing.assert_array_equal(got["coll_scalars"], got["coll_scalars_expected"])
        np.testing.assert_array_equal(got["coll_offset"], got["coll_offset_expected"])
        np.testing.assert_array_equal(got["coll_bcast"], [3.0, 4.0])
        total = int(got["coll_offset"][1])
        # rows 0, total-1, 3 of the global array: rank 0 holds 7 rows (values 0..20), the last rank the last row
        last_rank = len(ranks) - 1
        n_last = 7 + last_rank
        expect = np.array([[0., 1., 2.], np.arange((n_last - 1) * 3, n_last * 3) + 1000. * last_rank, [9., 10., 11.]])
        np.testing.assert_array_equal(got["coll_rows"], expect)
        assert total == sum(7 + q for q in range(len(ranks)))


def test_two_rank_vb_and_pmc_match_single_process():
    z = dist_worker.make_inputs()
    single = dist_worker.case(OracleBackend(), z, 0, len(z["data"]))
    ranks = spawn_two_ranks(z, "oracle")
    check_two_ranks(single, ranks, len(z["data"]), rtol=1e-10)
    check_collectives(ranks)


def test_one_rank_group_takes_the_sharded_paths():
    """a process group of ONE rank (what `torchrun --nproc-per-node 1` sets up) runs the collectives and the
    sharded code paths and reproduces the run without a group"""
    z = dist_worker.make_inputs(seed=3, N=257)
    single = dist_worker.case(OracleBackend(), z, 0, len(z["data"]))
    ranks = spawn_ranks(z, "oracle", world=1)
    check_collectives(ranks)
    for key, val in single.items():
        np.testing.assert_allclose(ranks[0][key], val, rtol=1e-12, atol=1e-13, err_msg=key)


def test_first_rows_spanning_ranks():
    """initial_guess='first' with more start means than rank 0 holds rows"""
    z = dist_worker.make_inputs(seed=8, N=9)
    be = OracleBackend()
    from pypmc_amd.mix_adapt.variational import GaussianInference
    single = GaussianInference(z["data"], components=6, initial_guess="first", backend=be).m
    np.testing.assert_array_equal(single, z["data"][:6])
    with tempfile.TemporaryDirectory() as tmp:
        import torch.multiprocessing as mp
        np.savez(os.path.join(tmp, "inputs.npz"), **z)
        mp.spawn(_first_rows_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
        for r in range(2):
            np.testing.assert_array_equal(np.load(os.path.join(tmp, "m%d.npy" % r)), single)


def _first_rows_worker(rank, world, port, workdir):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=dist_worker._rendezvous(workdir), rank=rank, world_size=world)
    try:
        from pypmc_amd import parallel
        from pypmc_amd.mix_adapt.variational import GaussianInference
        z = np.load(os.path.join(workdir, "inputs.npz"))
        lo, hi = parallel.shard_bounds(len(z["data"]))             # 5 + 4 rows
        vb = GaussianInference(z["data"][lo:hi], components=6, initial_guess="first", backend=OracleBackend())
        np.save(os.path.join(workdir, "m%d.npy" % rank), vb.m)
    finally:
        dist.destroy_process_group()


def test_init_from_env_refuses_a_world_without_a_rank():
    """advice r3: WORLD_SIZE > 1 with no RANK (mpirun / srun name their variables differently) used to fall into a private
    one-rank group -- every process unsharded, every all-reduce over its own data only.  Now it is an error; a forced
    one-rank group of a plain process still works and removes its rendezvous directory at exit."""
    code = r'''
import glob, os, sys, tempfile
sys.path.insert(0, %r)
os.environ.pop("RANK", None)
os.environ["WORLD_SIZE"] = "4"
from pypmc_amd import parallel
try:
    parallel.init_from_env()
    print("NO ERROR")
except RuntimeError as exc:
    print("refused:", "RANK is not set" in str(exc))
os.environ.pop("WORLD_SIZE")
os.environ["PMC_DIST_BACKEND"] = "gloo"
before = set(glob.glob(os.path.join(tempfile.gettempdir(), "pmc_rdzv_*")))
r, w, _ = parallel.init_from_env(force=True)
mine = set(glob.glob(os.path.join(tempfile.gettempdir(), "pmc_rdzv_*"))) - before
print("group:", r, w, parallel.active(), len(mine))
open(os.path.join(%r, "rdzv.txt"), "w").write("\n".join(mine))
import torch.distributed as dist
dist.destroy_process_group()
'''
    import subprocess
    import sys
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMC_FORCE_DIST"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, "-c", code % (ROOT, tmp)], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "refused: True" in out.stdout and "group: 0 1 True 1" in out.stdout, out.stdout
        left = [d for d in open(os.path.join(tmp, "rdzv.txt")).read().split("\n") if d]
        assert left and not any(os.path.exists(d) for d in left), "the rendezvous directory was not removed at exit"
