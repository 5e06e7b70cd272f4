"""N > 1 path on the product backend: two ranks share the box's one GPU (process group "gloo",
because RCCL refuses two ranks on one device), each runs the HIP kernels on its shard and the
statistics buffer is all-reduced -- results must equal the single-process HIP run and be bitwise
identical across ranks.  The driver's 8-GPU run uses the same code with backend "nccl"."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import dist_worker
from test_distributed_cpu import check_two_ranks, spawn_two_ranks, run_torchrun

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_hip_backend_match_single_process():
    from pypmc_amd.backend import HipBackend
    z = dist_worker.make_inputs(N=5003)
    single = dist_worker.case(HipBackend(), z, 0, len(z["data"]))
    ranks = spawn_two_ranks(z, "hip")
    assert all(str(r["backend"]) == "hip" for r in ranks)
    # the two shards' statistics are summed in a different order than the single run's chunks
    check_two_ranks(single, ranks, len(z["data"]), rtol=1e-9)


def test_two_ranks_vs_oracle():
    """the sharded HIP run against the single-process oracle run"""
    from oracle_backend import OracleBackend
    z = dist_worker.make_inputs(seed=11, N=1201)
    single = dist_worker.case(OracleBackend(), z, 0, len(z["data"]))
    ranks = spawn_two_ranks(z, "hip")
    for got in ranks:
        for key in ("vb_N_comp0", "vb_bound0", "vbf_m0", "vbf_N_comp0", "vbr_m0", "pmc_w", "pmc_mu", "pmc_sigma",
                    "pmcl_mu", "pmc_ll", "tpmc_mu", "tpmc_sigma", "tpmc_dof"):
            np.testing.assert_allclose(got[key], single[key], rtol=1e-9, atol=1e-11, err_msg=key)


def test_bench_two_ranks_one_gpu():
    """bench.py --gpus 2 end to end (torch.distributed.run, 2 ranks on the one GPU via gloo)"""
    env = dict(os.environ, PMC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = run_torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                         "--samples-per-gpu", "300000"], env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "cpu_baseline" not in line                  # rank 0 at N=1 only
    # more than one rank: the self-diagnosis ran (verdict r5 #5) -- the sum through every collective of the package, checked
    # bit for bit; here two processes share ONE GPU: torch.distributed's gloo, the library's RCCL communicator (two ranks
    # on one device: RCCL refuses that -- reported, not fatal) and the one-shot exchange (HIP IPC between the two processes)
    diag = line["dist"]["diagnostics"]
    assert diag["world_size"] == 2 and diag["backend"] == "gloo" and len(diag["ranks"]) == 2
    assert diag["default"]["ok"] is True and diag["default"]["ms_per_round"] > 0
    assert "rccl_native" in diag and ("ok" in diag["rccl_native"])
    p2p = diag["p2p"]
    assert p2p.get("enabled") is False or (p2p["ok"] is True and p2p["ms_per_round"] > 0 and "selftest=passed" in p2p["info"]), p2p


def test_bench_line_survives_a_diagnosis_that_never_returns():
    """the watchdog around the multi-GPU self-diagnosis: with a collective that hangs (PMC_DIAG_TEST_HANG) the line still goes
    out -- with the stage the diagnosis was stuck in -- and every rank leaves with status 0"""
    env = dict(os.environ, PMC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", PMC_DIAG_TEST_HANG="1")
    r = run_torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm", "2",
                         "--samples-per-gpu", "200000", "--diagnose-timeout", "3"], env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    diag = line["dist"]["diagnostics"]
    assert "timed out" in diag["error"] and diag["stage"] == "test hang"


@pytest.mark.parametrize("script,args", [("pmc_device_loop.py", ["100000", "2"]), ("variational.py", ["60000"]),
                                         ("pmc_torchrun.py", ["4000"])])
def test_examples_under_torchrun(script, args):
    """the multi-rank examples end to end: 4 ranks sharing the one GPU (PMC_DIST_BACKEND=gloo)"""
    env = dict(os.environ, PMC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = run_torchrun(4, [os.path.join(ROOT, "examples", script)] + args, env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    want = {"pmc_device_loop.py": "iteration 1", "variational.py": "converged after",
            "pmc_torchrun.py": "10 x 4000 samples on 4 rank(s)"}[script]
    assert want in r.stdout, r.stdout[-2000:]
