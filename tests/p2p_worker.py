"""One rank of tests/test_gpu_p2p.py: several processes share the box's one GPU, the process group (gloo, a file
rendezvous) only carries the mailbox handles; the sums go through pmc_p2p_allreduce_sum."""
import os
import sys

import numpy as np


def vector(rank, n, round_):
    rs = np.random.RandomState(1000 * rank + round_)
    return rs.normal(size=n) * 10.0 ** rs.randint(-3, 4)


def run(rank, world, workdir, sizes, rounds):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(workdir, "rendezvous"), rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from pypmc_amd import parallel
        assert parallel.enable_p2p_collective(max_doubles=max(sizes), device=0) is True
        assert parallel.collective_name() == "p2p:libpmc_hip"
        st = parallel.p2p_status()
        assert st["enabled"] and "selftest=passed" in st["info"], st
        out = {"info": np.array(st["info"])}
        for r in range(rounds):
            n = sizes[r % len(sizes)]
            t = torch.from_numpy(vector(rank, n, r)).cuda()
            got = parallel.all_reduce_sum(t)
            assert got is t
            out["round%d" % r] = t.cpu().numpy()
        # a second stream, back to back without a host synchronisation in between
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ts = [torch.from_numpy(vector(rank, sizes[0], 100 + r)).cuda() for r in range(6)]
            for t in ts:
                parallel.all_reduce_sum(t)
        s.synchronize()
        for r, t in enumerate(ts):
            out["burst%d" % r] = t.cpu().numpy()
        # larger than the mailbox: the process group's own all-reduce takes over
        big = torch.ones(max(sizes) + 5, dtype=torch.float64, device="cuda") * (rank + 1)
        parallel.all_reduce_sum(big)
        out["big"] = big.cpu().numpy()[:3]
        np.savez(os.path.join(workdir, "p2p_rank%d.npz" % rank), **out)
    finally:
        from pypmc_amd import parallel as _p
        _p.disable_p2p_collective()
        dist.destroy_process_group()


def run_selftest_failure(rank, world, workdir):
    """verdict r4 #2: the connect-time self-test is forced to fail on ONE rank (PMC_P2P_SELFTEST_CORRUPT: that rank expects
    another sum) -- every rank must come back without the exchange, with the reason, and the default collective carries on"""
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(workdir, "rendezvous"), rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from pypmc_amd import parallel
        os.environ["PMC_P2P_SELFTEST_CORRUPT"] = "1"
        used = parallel.enable_p2p_collective(max_doubles=5000, device=0)
        st = parallel.p2p_status()
        t = torch.from_numpy(vector(rank, 777, 5)).cuda()
        parallel.all_reduce_sum(t)
        np.savez(os.path.join(workdir, "fail_rank%d.npz" % rank), used=used, enabled=st["enabled"], reason=np.array(str(st["reason"])),
                 collective=np.array(str(parallel.collective_name())), sum=t.cpu().numpy())
    finally:
        dist.destroy_process_group()


def run_timeout(rank, world, workdir):
    """advice r4: a rank whose peer does not arrive within PMC_P2P_TIMEOUT_S must not keep its own unreduced numbers: the
    buffer is NaN, the call raises, and the exchange refuses further rounds"""
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(workdir, "rendezvous"), rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from pypmc_amd import parallel
        from pypmc_amd._lib import HipLibraryError
        os.environ["PMC_P2P_TIMEOUT_S"] = "abc"            # (does not parse: ignored, the default of 20 s stays)
        assert parallel.enable_p2p_collective(max_doubles=5000, device=0) is True
        t = torch.from_numpy(vector(rank, 100, 1)).cuda()
        parallel.all_reduce_sum(t)                          # a good round first (with the unparsable timeout)
        good = t.cpu().numpy()
        os.environ["PMC_P2P_TIMEOUT_S"] = "1.5"
        raised, again, after = False, False, None
        if rank == 0:                                       # rank 1 stays away from this round
            t2 = torch.from_numpy(vector(rank, 100, 2)).cuda()
            try:
                parallel.all_reduce_sum(t2)
            except HipLibraryError as exc:
                raised = "gave up waiting" in str(exc)
            after = t2.cpu().numpy()
            try:
                parallel.all_reduce_sum(t2)
            except HipLibraryError:
                again = True
        dist.barrier()
        np.savez(os.path.join(workdir, "timeout_rank%d.npz" % rank), good=good, raised=raised, again=again,
                 after=after if after is not None else np.zeros(0))
    finally:
        from pypmc_amd import parallel as _p
        _p.disable_p2p_collective()
        dist.destroy_process_group()


def run_ctx(rank, world, workdir):
    """the handle layer (include/pmc_ctx.h) sharded over `world` processes with pmc_ctx_p2p_open / _connect: every rank
    uploads its block of the samples, pmc_weighted_moments returns the moments of ALL ranks' samples"""
    import ctypes as C
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pypmc_amd import _lib
    lib = _lib.load()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    z = np.load(os.path.join(workdir, "ctx_inputs.npz"))
    x, w = z["x"], z["w"]
    N, D = x.shape
    lo, hi = rank * N // world, (rank + 1) * N // world
    ctx = C.c_void_p()
    assert lib.pmc_init(0, C.byref(ctx)) == 0, lib.pmc_last_error()
    mine = (C.c_char * 128)()                             # PMC_P2P_HANDLE_BYTES
    assert lib.pmc_ctx_p2p_open(ctx, rank, world, 8 + 1 + D + D * (D + 1) // 2 + 2, C.cast(mine, C.c_void_p)) == 0, lib.pmc_last_error()
    with open(os.path.join(workdir, "handle%d.tmp" % rank), "wb") as f:
        f.write(bytes(mine))
    os.rename(os.path.join(workdir, "handle%d.tmp" % rank), os.path.join(workdir, "handle%d.bin" % rank))
    t0 = time.time()
    while not all(os.path.exists(os.path.join(workdir, "handle%d.bin" % r)) for r in range(world)):
        assert time.time() - t0 < 120, "the other ranks' handles did not arrive"
        time.sleep(0.01)
    allh = b"".join(open(os.path.join(workdir, "handle%d.bin" % r), "rb").read() for r in range(world))
    buf = (C.c_char * len(allh)).from_buffer_copy(allh)
    assert lib.pmc_ctx_p2p_connect(ctx, C.cast(buf, C.c_void_p)) == 0, lib.pmc_last_error()
    # (every rank must have mapped every mailbox before anybody writes: a second file barrier)
    open(os.path.join(workdir, "mapped%d" % rank), "w").close()
    while not all(os.path.exists(os.path.join(workdir, "mapped%d" % r)) for r in range(world)):
        assert time.time() - t0 < 120
        time.sleep(0.01)
    s = C.c_void_p()
    xs = np.ascontiguousarray(x[lo:hi])
    assert lib.pmc_samples_upload(ctx, dp(xs), hi - lo, D, C.byref(s)) == 0, lib.pmc_last_error()
    mean, cov = np.empty(D), np.empty((D, D))
    for _ in range(3):
        assert lib.pmc_weighted_moments(ctx, s, dp(np.ascontiguousarray(w[lo:hi])), 0, dp(mean), dp(cov)) == 0, lib.pmc_last_error()
    np.savez(os.path.join(workdir, "ctx_rank%d.npz" % rank), mean=mean, cov=cov)
    open(os.path.join(workdir, "done%d" % rank), "w").close()
    while not all(os.path.exists(os.path.join(workdir, "done%d" % r)) for r in range(world)):
        assert time.time() - t0 < 120
        time.sleep(0.01)                                   # nobody unmaps a mailbox a peer may still write to
    lib.pmc_samples_free(s)
    assert lib.pmc_shutdown(ctx) == 0
