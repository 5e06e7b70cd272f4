"""The one-shot exchange (csrc/pmc_p2p.hip, SURVEY section 5: one-shot P2P all-gather + ordered local sum) with 2 and 4
processes sharing the box's one GPU: every rank writes its vector into every rank's mailbox through HIP IPC, waits for
the flags and adds the vectors in RANK ORDER -- bit-equal to that sum computed on the host, identical on all ranks,
round after round (two alternating slot sets), on a second stream, and under the sharded front-end."""
import os
import tempfile

import numpy as np
import pytest

import dist_worker
import p2p_worker
from test_distributed_cpu import check_two_ranks

pytestmark = pytest.mark.gpu
SIZES = [7464, 110336, 1, 33]                              # the statistics vectors of K = 32, D = 20 and K = 128, D = 40


@pytest.mark.parametrize("world", [2, 4])
def test_ordered_sum_bit_for_bit(world):
    import torch.multiprocessing as mp
    rounds = 24
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(p2p_worker.run, args=(world, tmp, SIZES, rounds), nprocs=world, join=True)
        ranks = [dict(np.load(os.path.join(tmp, "p2p_rank%d.npz" % r))) for r in range(world)]

    def expect(n, tag):
        s = p2p_worker.vector(0, n, tag)
        for r in range(1, world):
            s = s + p2p_worker.vector(r, n, tag)           # in rank order, one addition at a time
        return s
    for r in range(rounds):
        ref = expect(SIZES[r % len(SIZES)], r)
        for got in ranks:
            np.testing.assert_array_equal(got["round%d" % r], ref)
    for r in range(6):
        ref = expect(SIZES[0], 100 + r)
        for got in ranks:
            np.testing.assert_array_equal(got["burst%d" % r], ref)
    for got in ranks:
        np.testing.assert_array_equal(got["big"], np.full(3, world * (world + 1) / 2.0))


def test_a_failed_self_test_leaves_the_default_collective_on_every_rank():
    """verdict r4 #2: the self-test of ONE rank is forced to fail; nobody uses the exchange, everybody knows why, and the
    sums still come out (through the process group's own collective)"""
    import torch.multiprocessing as mp
    world = 3
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(p2p_worker.run_selftest_failure, args=(world, tmp), nprocs=world, join=True)
        ranks = [dict(np.load(os.path.join(tmp, "fail_rank%d.npz" % r))) for r in range(world)]
    ref = p2p_worker.vector(0, 777, 5)
    for r in range(1, world):
        ref = ref + p2p_worker.vector(r, 777, 5)
    for i, got in enumerate(ranks):
        assert not bool(got["used"]) and not bool(got["enabled"]) and str(got["collective"]) == "gloo", (i, got)
        assert "self-test" in str(got["reason"]) or "peer" in str(got["reason"]), str(got["reason"])
        np.testing.assert_allclose(got["sum"], ref, rtol=1e-15)
    assert "self-test" in str(ranks[1]["reason"])               # the rank that was made to fail names its own finding


def test_a_round_that_times_out_is_nan_and_an_error_never_a_local_sum():
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(p2p_worker.run_timeout, args=(2, tmp), nprocs=2, join=True)
        r0, r1 = [dict(np.load(os.path.join(tmp, "timeout_rank%d.npz" % r))) for r in range(2)]
    ref = p2p_worker.vector(0, 100, 1) + p2p_worker.vector(1, 100, 1)
    np.testing.assert_array_equal(r0["good"], ref)
    np.testing.assert_array_equal(r1["good"], ref)
    assert bool(r0["raised"]) and bool(r0["again"])
    assert np.isnan(r0["after"]).all(), "a timed-out round must not leave the rank's own numbers behind"


def test_sharded_front_end_over_the_one_shot_exchange():
    """the two-rank HIP run of tests/test_gpu_distributed.py with PMC_P2P_COLLECTIVE=1: same results as one process"""
    import torch.multiprocessing as mp
    from pypmc_amd.backend import HipBackend
    from test_distributed_cpu import _free_port
    z = dist_worker.make_inputs(N=5003)
    single = dist_worker.case(HipBackend(), z, 0, len(z["data"]))
    old = os.environ.get("PMC_P2P_COLLECTIVE")
    os.environ["PMC_P2P_COLLECTIVE"] = "1"
    try:
        with tempfile.TemporaryDirectory() as tmp:
            np.savez(os.path.join(tmp, "inputs.npz"), **z)
            mp.spawn(dist_worker.run, args=(2, _free_port(), tmp, "hip", "gloo"), nprocs=2, join=True)
            ranks = [dict(np.load(os.path.join(tmp, "rank%d.npz" % r))) for r in range(2)]
    finally:
        if old is None:
            os.environ.pop("PMC_P2P_COLLECTIVE")
        else:
            os.environ["PMC_P2P_COLLECTIVE"] = old
    assert all(str(r["collective"]) == "p2p:libpmc_hip" for r in ranks)
    check_two_ranks(single, ranks, len(z["data"]), rtol=1e-9)


def test_handle_layer_sharded_over_the_one_shot_exchange():
    """pmc_ctx_p2p_open / pmc_ctx_p2p_connect: three processes, each with a context and its block of the samples;
    calculate_mean / calculate_covariance (importance_sampling.py:46-83) of ALL samples on every rank, bit-identical
    across the ranks"""
    import torch.multiprocessing as mp
    world, N, D = 3, 30011, 5
    rs = np.random.RandomState(4)
    x = rs.normal(size=(N, D)) * np.arange(1, D + 1) + 3.0
    w = rs.uniform(0.1, 2.0, N)
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "ctx_inputs.npz"), x=x, w=w)
        mp.spawn(p2p_worker.run_ctx, args=(world, tmp), nprocs=world, join=True)
        ranks = [dict(np.load(os.path.join(tmp, "ctx_rank%d.npz" % r))) for r in range(world)]
    mean = (w[:, None] * x).sum(axis=0) / w.sum()
    d = x - mean
    cov = np.einsum('n,ni,nj->ij', w, d, d) / w.sum() * (w.sum() ** 2 / (w.sum() ** 2 - (w ** 2).sum()))
    for got in ranks:
        np.testing.assert_allclose(got["mean"], mean, rtol=1e-12)
        np.testing.assert_allclose(got["cov"], cov, rtol=1e-10, atol=1e-12)
        np.testing.assert_array_equal(got["mean"], ranks[0]["mean"])
        np.testing.assert_array_equal(got["cov"], ranks[0]["cov"])
