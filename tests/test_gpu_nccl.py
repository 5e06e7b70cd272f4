"""The RCCL path for real, on the one GPU of the box: a process group of ONE rank with backend "nccl"
(torch.distributed's name for RCCL on ROCm) runs every collective the sharded front-end issues -- the joined
statistics / bookkeeping buffer of the PMC updates, the VB E-step's statistics vector, the rows of the global
sample array, scalars staged from numpy -- through ncclAllReduce on the device, and bench.py under
``torch.distributed.run --nproc-per-node 1`` reports what the group really was.  An 8-GPU run issues exactly
these calls with world_size 8 (reference: pypmc/tools/parallel_sampler.py:58-71 gathers with mpi4py instead)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import dist_worker
from test_distributed_cpu import check_collectives, spawn_ranks, run_torchrun

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_nccl_group_runs_the_sharded_surface():
    from pypmc_amd.backend import HipBackend
    z = dist_worker.make_inputs(seed=21, N=4099)
    single = dist_worker.case(HipBackend(), z, 0, len(z["data"]))           # no process group in this process
    ranks = spawn_ranks(z, "hip", world=1, pg_backend="nccl")
    got = ranks[0]
    assert str(got["backend"]) == "hip" and str(got["pg_backend"]) == "nccl"
    check_collectives(ranks)
    for key, val in single.items():
        # same kernels on the same rows; the reduced buffer is the sum over ONE rank: nothing may move
        np.testing.assert_allclose(got[key], val, rtol=1e-13, atol=1e-14, err_msg=key)
    np.testing.assert_array_equal(got["vbf_m0"], single["vbf_m0"])
    np.testing.assert_array_equal(got["vbr_m0"], single["vbr_m0"])


def test_one_rank_through_the_librarys_own_communicator(monkeypatch):
    """PMC_NATIVE_COLLECTIVE=1: the all-reduce is pmc_comm_allreduce_sum (ncclAllReduce issued by libpmc_hip on the
    launch stream); torch.distributed only carries the unique id"""
    from pypmc_amd.backend import HipBackend
    z = dist_worker.make_inputs(seed=22, N=3001)
    single = dist_worker.case(HipBackend(), z, 0, len(z["data"]))
    monkeypatch.setenv("PMC_NATIVE_COLLECTIVE", "1")
    ranks = spawn_ranks(z, "hip", world=1, pg_backend="nccl")
    got = ranks[0]
    assert str(got["collective"]) == "rccl:libpmc_hip"
    check_collectives(ranks)
    for key, val in single.items():
        np.testing.assert_allclose(got[key], val, rtol=1e-13, atol=1e-14, err_msg=key)


def _bench(launcher, extra, env=None):
    """bench.py on a small batch: `launcher` = [python] for a plain process, "torchrun" for one rank under
    torch.distributed.run (retried on another port should the probed one be taken in between)"""
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
            "--samples-per-gpu", "400000", "--no-cpu-baseline", "--no-configs"] + extra
    if launcher == "torchrun":
        r = run_torchrun(1, args, env or dict(os.environ), timeout=900)
    else:
        r = subprocess.run(launcher + args, cwd=ROOT, env=env or dict(os.environ), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_under_torchrun_one_rank_is_rccl(scaling):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("PMC_DIST_BACKEND", None)
    line = _bench("torchrun", ["--scaling", scaling], env)
    assert line["dist"]["backend"] == "nccl" and line["dist"]["world_size"] == 1
    assert line["dist"]["allreduce_ms"] > 0.0 and line["dist"]["allreduce_doubles"] > 0
    assert line["scaling"] == scaling and line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["N_total"] == 400000 and line["config"]["N_per_gpu"] == 400000


def test_bench_plain_python_has_no_group_unless_forced():
    line = _bench([sys.executable], [])
    assert line["dist"]["backend"] is None and line["dist"]["world_size"] == 1
    line = _bench([sys.executable], ["--force-dist"])
    assert line["dist"]["backend"] == "nccl" and line["dist"]["world_size"] == 1 and line["dist"]["allreduce_ms"] > 0.0
    line = _bench([sys.executable], ["--force-dist"], dict(os.environ, PMC_NATIVE_COLLECTIVE="1"))
    assert line["dist"]["backend"] == "rccl:libpmc_hip" and line["dist"]["allreduce_ms"] > 0.0
    assert line["dist"]["diagnostics"] is None          # one rank, not asked for
    # --diagnose: the self-diagnosis of a multi-GPU run on the one rank a one-GPU box has -- RCCL for real, through
    # torch.distributed and through the library's own communicator, and the one-shot exchange with itself
    line = _bench([sys.executable], ["--force-dist", "--diagnose"])
    diag = line["dist"]["diagnostics"]
    assert diag["world_size"] == 1 and diag["backend"] == "nccl"
    for name in ("default", "rccl_native"):
        assert diag[name]["ok"] is True and diag[name]["ms_per_round"] > 0, (name, diag[name])
    assert diag["p2p"].get("enabled") is False or diag["p2p"]["ok"] is True, diag["p2p"]
    assert diag.get("per_rank_errors") in ({}, None) or diag["p2p"].get("enabled") is False


def test_bench_single_process_over_virtual_shards():
    """bench.py --gpus 2 --single-process --devices 0,0: the driver's contract line from ONE process that owns both shards
    (pmc_init_devices; the same ordinal twice on a one-GPU box)"""
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0,0", "--steps", "3", "--warmup", "1",
            "--samples-per-gpu", "300000", "--no-cpu-baseline", "--no-configs", "--no-traffic"]
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and line["unit"].startswith("samples/s")
    assert line["config"]["N_total"] == 600000 and line["dtype"] == "f64"
    assert 0 < line["roofline"]["frac"] < 1 and len(line["step_ms"]["all"]) == 3
    assert "ordered_sum_check" not in line["dist"]      # virtual shards, not asked for
    # --diagnose: the check a run over several REAL devices makes by itself -- the devices' ordered sum against the same
    # shards as virtual shards of the first device, bit for bit
    r = subprocess.run([sys.executable] + args + ["--diagnose"], cwd=ROOT, env=dict(os.environ), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    chk = line["dist"]["ordered_sum_check"]
    assert chk.get("matches_virtual_shards_bitwise") is True and len(chk["shards"]) == 2, chk
