"""Multi-GPU plumbing: one process per GPU (torchrun), samples sharded by rank, and exactly one
collective on the hot path -- a sum all-reduce of the K-sized statistics vector
(torch.distributed: backend "nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU test-suite).

The reference's only distributed code gathers whole pickled sample histories to rank 0 with
mpi4py (pypmc/tools/parallel_sampler.py:58-71) and broadcasts the adapted proposal back
(examples/pmc_mpi.py:119-131).  Here samples never leave their GPU: every rank reduces its shard to
[scalars | K x (1 + D + D(D+1)/2) | K x 2 | bookkeeping tail] doubles, the ranks all-reduce that
buffer, and every rank runs the identical K-sized host update -- no broadcast is needed.

Everything that must be identical on all ranks but is derived from data (the start means of
GaussianInference(initial_guess='first'/'random'), the global sample count) goes through a sum
all-reduce as well -- the owner contributes the value, everybody else zeros -- so the result is
bitwise the same everywhere.
"""
import numpy as np


import os as _os

# RCCL shares device memory between the ranks of a node through hipIpcGetMemHandle / hipIpcOpenMemHandle; the host driver
# of the MI355X boxes supports only the dmabuf IPC mode, and with the legacy mode (this variable unset or 1) the first of
# the two calls fails with "invalid argument" as soon as a second process is involved -- measured on a box of the pool
# with two processes sharing one allocation: profiles/r04_ipc_mode.txt (scripts/ipc_mode_probe.py).  The image exports
# the variable already; this is a setdefault (a value the launcher exported wins) for ranks started from an environment
# that lost it, and it has to happen before the HIP runtime starts (it is read at initialisation).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def world_size():
    d = _dist()
    return d.get_world_size() if d else 1


def rank():
    d = _dist()
    return d.get_rank() if d else 0


def active():
    """True when a process group exists -- the sharded code paths (joined exchange buffer, global rows, one
    collective per update) are taken then, for a group of one rank as well: a single GPU under ``torchrun
    --nproc-per-node 1`` runs exactly what eight do."""
    return _dist() is not None


def shard_bounds(N, r=None, world=None):
    """[begin, end) of rank ``r``'s contiguous block of N samples (sizes differ by at most 1)."""
    r = rank() if r is None else r
    world = world_size() if world is None else world
    base, extra = divmod(int(N), world)
    begin = r * base + min(r, extra)
    return begin, begin + base + (1 if r < extra else 0)


def init_from_env(force=False):
    """Join the process group ``torchrun`` set up (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*): one process per
    GPU, backend "nccl" (= RCCL over xGMI).  ``PMC_DIST_BACKEND=gloo`` is the development aid used by the
    tests: several ranks share the GPUs that exist (RCCL refuses two ranks on one device).
    Returns (rank, world_size, local device index).

    A launch under ``torchrun`` (RANK and WORLD_SIZE in the environment) always gets its group, a single rank
    included -- ``torchrun --nproc-per-node 1`` runs the very collectives an 8-GPU run issues, on one GPU.
    A plain ``python`` process gets none (0, 1, current device) unless ``force`` (or ``PMC_FORCE_DIST=1``) asks
    for a one-rank group, which meets itself through a rendezvous file of its own (removed at exit).
    WORLD_SIZE > 1 without RANK -- a launcher that names its variables differently (mpirun, srun) -- is an error: a
    private one-rank group there would make every process run unsharded and all-reduce only its own data, silently."""
    import os
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    force = force or os.environ.get("PMC_FORCE_DIST", "0") not in ("", "0")
    if world > 1 and not launched:
        raise RuntimeError("WORLD_SIZE=%d but RANK is not set: launch with torch.distributed.run (torchrun), or export "
                           "RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the launcher's own variables "
                           "(e.g. OMPI_COMM_WORLD_RANK, SLURM_PROCID)" % world)
    if world <= 1 and not launched and not force:
        return 0, 1, torch.cuda.current_device() if torch.cuda.is_available() else 0
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("PMC_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    if backend == "nccl" or torch.cuda.is_available():      # (a gloo group on a box without a GPU: the CPU test-suite)
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        # a forced one-rank group of a plain process meets itself through a file of its own: no TCP port to probe and
        # lose to another process before it is bound; a launched group uses what the launcher set up (env://)
        extra = {}
        if not launched:                                    # (force, one rank: checked above)
            import atexit
            import shutil
            import tempfile
            rdzv = tempfile.mkdtemp(prefix="pmc_rdzv_")
            atexit.register(shutil.rmtree, rdzv, ignore_errors=True)
            extra = dict(init_method="file://" + os.path.join(rdzv, "store"), rank=0, world_size=1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), **extra)
        else:
            dist.init_process_group(backend, **extra)
    if os.environ.get("PMC_NATIVE_COLLECTIVE", "0") not in ("", "0"):
        enable_native_collective(local)
    if os.environ.get("PMC_P2P_COLLECTIVE", "0") not in ("", "0"):
        enable_p2p_collective(device=local)
    return dist.get_rank(), dist.get_world_size(), local


# ---- the library's own RCCL communicator (include/pmc_hip.h: pmc_comm_*), optional ---------------------------
_native = None            # (ctypes handle of the pmc_comm, the library) once enable_native_collective() has run


def enable_native_collective(device=None):
    """Run the path's all-reduce through libpmc_hip's own RCCL communicator (``pmc_comm_allreduce_sum``:
    ncclAllReduce on the caller's stream) instead of ``torch.distributed.all_reduce``.  torch.distributed is then
    only the bootstrap: rank 0 draws the RCCL unique id, the existing process group carries it to the ranks, every
    rank joins.  Opt-in (``PMC_NATIVE_COLLECTIVE=1`` makes ``init_from_env`` call this): it is exercised on one
    rank by the GPU tests; torch.distributed stays the default for multi-GPU runs."""
    global _native
    d = _dist()
    if d is None:
        raise RuntimeError("enable_native_collective needs an initialised torch.distributed process group")
    if _native is not None:
        return
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    dev = torch.cuda.current_device() if device is None else int(device)
    ident = (C.c_char * 128)()
    if d.get_rank() == 0:
        _lib.check(lib.pmc_comm_unique_id(C.cast(ident, C.c_void_p)), "pmc_comm_unique_id")
    where = torch.device("cuda", dev) if d.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device=where)
    d.broadcast(t, src=0)
    ident = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().tolist()))
    comm = C.c_void_p()
    _lib.check(lib.pmc_comm_init(d.get_rank(), d.get_world_size(), C.cast(ident, C.c_void_p), dev, C.byref(comm)),
               "pmc_comm_init")
    _native = (comm, lib)


def disable_native_collective():
    global _native
    if _native is not None:
        comm, lib = _native
        _native = None
        lib.pmc_comm_destroy(comm)


def collective_name():
    """what sums the statistics buffer over the ranks: None (no group), 'nccl' / 'gloo' (torch.distributed's
    backend), 'rccl:libpmc_hip' (the library's own RCCL communicator) or 'p2p:libpmc_hip' (its one-shot exchange)"""
    d = _dist()
    if d is None:
        return None
    if _p2p is not None:
        return "p2p:libpmc_hip"                          # (device buffers that fit the mailbox; the rest as below)
    return "rccl:libpmc_hip" if _native is not None else d.get_backend()


def _native_all_reduce(t):
    """in-place sum of a float64 CUDA tensor over the ranks through pmc_comm_allreduce_sum, on the current stream"""
    import ctypes as C
    import torch
    from . import _lib
    comm, lib = _native
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    _lib.check(lib.pmc_comm_allreduce_sum(comm, C.c_void_p(t.data_ptr()), t.numel(), stream), "pmc_comm_allreduce_sum")


# ---- the one-shot exchange among the ranks of a node (include/pmc_hip.h: pmc_p2p_*), optional --------------------
_p2p = None               # (ctypes handle of the pmc_p2p, the library, capacity in doubles)
_p2p_reason = None        # why the exchange was asked for and is NOT in use (the RCCL / gloo collective runs instead)
_P2P_HANDLE_BYTES = 128   # PMC_P2P_HANDLE_BYTES


def _agree(d, ok, where):
    """True only if ``ok`` on EVERY rank (one sum all-reduce of the failure flags): the ranks must all use the exchange or
    all stay with the default collective"""
    import torch
    t = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=where)
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return float(t.cpu()[0]) == 0.0


def enable_p2p_collective(max_doubles=1 << 18, device=None):
    """Run the path's all-reduce of DEVICE buffers of up to ``max_doubles`` doubles as the library's one-shot exchange
    (``pmc_p2p_allreduce_sum``: every rank writes its vector into a mailbox of every peer through HIP IPC and adds the
    G vectors in rank order -- one hop instead of a ring's 2 (G - 1), bit-identical on all ranks and from run to run).
    The ranks of ONE node; the existing process group only carries the mailbox handles at set-up.  Opt-in
    (``PMC_P2P_COLLECTIVE=1`` makes ``init_from_env`` call this).

    Returns True if the exchange is in use.  It is NOT when the ranks are not on one host, are more than 16, a peer's
    device is not reachable, the mapping fails or the connect-time self-test round (a known pattern, compared bit for
    bit on every rank) does not come back right on ANY rank: the reason is logged and kept (``p2p_status()``), nothing
    stays half-open, and the default collective (RCCL / gloo) keeps running -- on all ranks alike."""
    global _p2p, _p2p_reason
    d = _dist()
    if d is None:
        raise RuntimeError("enable_p2p_collective needs an initialised torch.distributed process group")
    if _p2p is not None:
        return True
    import ctypes as C
    import logging
    import socket
    import zlib
    import torch
    from . import _lib
    log = logging.getLogger(__name__)
    lib = _lib.load()
    dev = torch.cuda.current_device() if device is None else int(device)
    world, me = d.get_world_size(), d.get_rank()
    where = torch.device("cuda", dev) if d.get_backend() == "nccl" else torch.device("cpu")

    def give_up(reason):
        global _p2p_reason
        _p2p_reason = reason
        log.warning("one-shot exchange not used (%s): the %s collective stays", reason, d.get_backend())
        return False

    # one node, at most 16 ranks: known before anything is allocated (advice r4)
    hosts = torch.zeros(world, dtype=torch.float64, device=where)
    hosts[me] = float(zlib.crc32(socket.gethostname().encode()) + 1)
    d.all_reduce(hosts, op=d.ReduceOp.SUM)
    hosts = hosts.cpu().tolist()
    if world > 16:
        return give_up("%d ranks: the mailboxes serve at most 16" % world)
    if len(set(hosts)) != 1:
        return give_up("the ranks run on %d different hosts" % len(set(hosts)))
    h = C.c_void_p()
    err = None
    if lib.pmc_p2p_create(me, world, int(max_doubles), dev, C.byref(h)) < 0:
        err = "rank %d: %s" % (me, _lib.last_error())
        h = None
    mine = (C.c_char * _P2P_HANDLE_BYTES)()
    if h is not None and lib.pmc_p2p_handle(h, C.cast(mine, C.c_void_p)) < 0:
        err = "rank %d: %s" % (me, _lib.last_error())
    n = _P2P_HANDLE_BYTES
    allh = torch.zeros(world * n, dtype=torch.float64, device=where)      # (a sum all-reduce: every backend has one)
    allh[me * n:(me + 1) * n] = torch.tensor(list(bytes(mine)), dtype=torch.float64, device=where)
    d.all_reduce(allh, op=d.ReduceOp.SUM)
    if not _agree(d, err is None, where):
        if h is not None:
            lib.pmc_p2p_destroy(h)
        return give_up(err or "a peer could not create its mailbox")
    raw = bytes(int(v) for v in allh.cpu().tolist())
    buf = (C.c_char * len(raw)).from_buffer_copy(raw)
    # (collective: mapping + the self-test round; a rank that fails makes its peers' self-test time out)
    if lib.pmc_p2p_connect(h, C.cast(buf, C.c_void_p)) < 0:
        err = "rank %d: %s" % (me, _lib.last_error())
    ok = _agree(d, err is None, where)
    d.barrier()                                          # every mailbox is mapped everywhere before the first round / nobody
    if not ok:                                           # unmaps one a peer's self-test may still write to
        lib.pmc_p2p_destroy(h)
        return give_up(err or "a peer's connect / self-test failed")
    _p2p = (h, lib, int(max_doubles))
    _p2p_reason = None
    return True


def p2p_status():
    """dict(enabled, info, reason): whether the one-shot exchange is in use, pmc_p2p_info's line
    ("memory=finegrained world=4 ... selftest=passed"), or why it is not"""
    if _p2p is None:
        return dict(enabled=False, info=None, reason=_p2p_reason)
    import ctypes as C
    h, lib, _ = _p2p
    buf = C.create_string_buffer(256)
    lib.pmc_p2p_info(h, buf, 256)
    return dict(enabled=True, info=buf.value.decode(), reason=None)


def disable_p2p_collective():
    global _p2p
    if _p2p is not None:
        h, lib, _ = _p2p
        _p2p = None
        d = _dist()
        if d is not None:
            d.barrier()                                  # nobody unmaps a mailbox a peer may still write to
        lib.pmc_p2p_destroy(h)


def _p2p_all_reduce(t):
    import ctypes as C
    import torch
    from . import _lib
    h, lib, _ = _p2p
    stream = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    _lib.check(lib.pmc_p2p_allreduce_sum(h, C.c_void_p(t.data_ptr()), t.numel(), stream), "pmc_p2p_allreduce_sum")
    # The caller reads the sum on the host next (it synchronises anyway): a round in which a peer did not arrive within
    # PMC_P2P_TIMEOUT_S -- the buffer is NaN then, never this rank's own numbers -- is an error here, not a silent
    # divergence of the ranks' mixtures (advice r4)
    _lib.check(lib.pmc_p2p_status(h, stream), "pmc_p2p_allreduce_sum")


def diagnose(n_doubles, rounds=100, device=None, progress=None):
    """First contact with several GPUs, self-diagnosing (verdict r5 #5): the builder has never had more than one.  Outside any
    timed region, COLLECTIVELY (every rank calls it), never raising: the sum of one statistics-sized device buffer over
    the ranks through every way this package has --

      * ``default``: torch.distributed's backend ("nccl" = RCCL on ROCm; "gloo" in the tests),
      * ``rccl_native``: the library's own communicator (pmc_comm_*: ncclAllReduce on the caller's stream),
      * ``p2p``: the one-shot exchange (pmc_p2p_*: mailboxes mapped through HIP IPC, peer stores over xGMI) -- its connect
        ends with a bit-exact self-test round, and a failure on any rank leaves all ranks where they were --

    each checked bit for bit against the sum known in closed form (rank-dependent integers: every partial sum is exact)
    and timed over ``rounds`` rounds.  Returns, on every rank, dict(world_size, backend, ranks_hosts, default | rccl_native |
    p2p = dict(ok, ms_per_round | error, ...)); the per-rank error texts are gathered so that rank 0's dict shows them
    all.  The stages stay in lockstep: a stage that fails on one rank is skipped by all (one agreement all-reduce in front
    of every collective part).  ``progress`` (a dict): its key "stage" names the stage that is running -- a caller that runs
    this in a watchdog thread (bench.py) can say where a collective that never returned was stuck.
    Reference role: the gather of pypmc/tools/parallel_sampler.py:58-71."""
    progress = progress if progress is not None else {}
    progress["stage"] = "start"
    if _os.environ.get("PMC_DIAG_TEST_HANG"):             # (test switch: a collective that never returns -- the caller's
        import time as _t                                 #  watchdog has to get the headline out without this function)
        progress["stage"] = "test hang"
        _t.sleep(1e9)
    import socket
    import time
    d = _dist()
    if d is None:
        return dict(world_size=1, backend=None, note="no process group")
    import torch
    dev = torch.cuda.current_device() if device is None else int(device)
    cuda = torch.device("cuda", dev)
    where = _collective_device(d)
    world, me = d.get_world_size(), d.get_rank()
    out = dict(world_size=world, backend=d.get_backend(), device=dev)
    try:
        names = [None] * world
        d.all_gather_object(names, "%s:cuda%d" % (socket.gethostname(), dev))
        out["ranks"] = names
    except Exception as exc:                              # (reported, never fatal)
        out["ranks_error"] = repr(exc)
    n = int(n_doubles)
    base = (torch.arange(n, dtype=torch.float64, device=cuda) % 97.0 + 1.0) / 128.0
    expect = base * (world * (world + 1) // 2)            # sum over ranks of (rank + 1) * base: exact in fp64

    def fresh():
        return (base * float(me + 1)).contiguous()

    def stage(name, reduce_fn, setup=None, teardown=None):
        rec = {}
        err = None
        progress["stage"] = name
        try:
            if setup is not None:
                extra = setup()
                if isinstance(extra, dict):
                    rec.update(extra)
                if rec.get("enabled") is False:
                    out[name] = rec
                    return
        except Exception as exc:
            err = repr(exc)
        if not _agree(d, err is None, where):             # (every rank or none)
            rec.update(ok=False, error=err or "a peer failed to set this stage up")
            out[name] = rec
            return
        try:
            t = fresh()
            reduce_fn(t)
            torch.cuda.synchronize(cuda)
            rec["ok"] = bool(torch.equal(t, expect))
            if not rec["ok"]:
                rec["max_abs_error"] = float((t - expect).abs().max())
            torch.cuda.synchronize(cuda)
            d.barrier()
            t0 = time.perf_counter()
            for _ in range(int(rounds)):
                reduce_fn(t)
            torch.cuda.synchronize(cuda)
            rec["ms_per_round"] = (time.perf_counter() - t0) / max(int(rounds), 1) * 1e3
            rec["doubles"] = n
        except Exception as exc:
            rec.update(ok=False, error=repr(exc))
        finally:
            try:
                if teardown is not None:
                    teardown()
            except Exception as exc:
                rec["teardown_error"] = repr(exc)
        out[name] = rec

    def default_reduce(t):
        if where.type == "cuda":
            d.all_reduce(t, op=d.ReduceOp.SUM)
        else:
            h = t.cpu()
            d.all_reduce(h, op=d.ReduceOp.SUM)
            t.copy_(h)
    stage("default", default_reduce)
    had_native, had_p2p = _native is not None, _p2p is not None
    stage("rccl_native", _native_all_reduce, setup=lambda: enable_native_collective(dev),
          teardown=None if had_native else disable_native_collective)

    def p2p_setup():
        on = enable_p2p_collective(max_doubles=max(n, 1 << 12), device=dev)
        st = p2p_status()
        return dict(enabled=bool(on), info=st.get("info"), reason=st.get("reason"))
    stage("p2p", _p2p_all_reduce, setup=p2p_setup, teardown=None if had_p2p else disable_p2p_collective)
    progress["stage"] = "gathering the ranks' errors"
    try:
        errs = [None] * world
        mine = {k: v.get("error") or v.get("reason") for k, v in out.items() if isinstance(v, dict) and (v.get("error") or v.get("reason"))}
        d.all_gather_object(errs, mine)
        out["per_rank_errors"] = {str(r): e for r, e in enumerate(errs) if e}
    except Exception as exc:
        out["per_rank_errors_error"] = repr(exc)
    progress["stage"] = "done"
    return out


def _collective_device(d):
    """device the default process group's backend reduces on"""
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if d.get_backend() == "nccl" else torch.device("cpu")


def all_reduce_sum(buf):
    """In-place sum over ranks of a float64 buffer; returns ``buf``.  A no-op without a process group; a group
    of ONE rank still runs the collective (that is how the RCCL path is exercised on a one-GPU box).

    The buffer is reduced where the process group's backend works -- RCCL ("nccl") on the GPU, gloo
    on the host -- and staged through the other memory when it lives there: a numpy / CPU buffer
    under nccl goes through a device copy, a device tensor under gloo (the CPU test-suite and
    `PMC_DIST_BACKEND=gloo`, several ranks on one GPU) through a host copy."""
    d = _dist()
    if d is None:
        return buf
    import torch
    if _p2p is not None and not isinstance(buf, np.ndarray) and buf.is_cuda and buf.is_contiguous() \
            and buf.dtype == torch.float64 and buf.numel() <= _p2p[2]:
        _p2p_all_reduce(buf)                            # the one-shot exchange (device buffers that fit the mailbox)
        return buf
    if _native is not None:                             # the library's own communicator: always on the device
        cuda = torch.device("cuda", torch.cuda.current_device())
        if isinstance(buf, np.ndarray):
            t = torch.from_numpy(buf)
            g = t.to(cuda)
            _native_all_reduce(g)
            t.copy_(g)
        elif buf.is_cuda and buf.is_contiguous():
            _native_all_reduce(buf)
        else:
            g = buf.to(cuda).contiguous()
            _native_all_reduce(g)
            buf.copy_(g)
        return buf
    dev = _collective_device(d)
    if isinstance(buf, np.ndarray):
        t = torch.from_numpy(buf)                       # shares memory with buf
        if dev.type == "cpu":
            d.all_reduce(t, op=d.ReduceOp.SUM)
        else:
            g = t.to(dev)
            d.all_reduce(g, op=d.ReduceOp.SUM)
            t.copy_(g)
        return buf
    if buf.device.type == dev.type:
        d.all_reduce(buf, op=d.ReduceOp.SUM)
    else:
        g = buf.to(dev)
        d.all_reduce(g, op=d.ReduceOp.SUM)
        buf.copy_(g)
    return buf


def all_reduce_scalars(*values):
    """Sum a few python floats over ranks (setup-time bookkeeping: global N, global sum of weights)."""
    d = _dist()
    if d is None:
        return tuple(float(v) for v in values)
    a = np.array([float(v) for v in values], dtype=np.float64)
    all_reduce_sum(a)
    return tuple(float(v) for v in a)


def shard_offset(n_local):
    """Global index of this rank's first sample when the ranks hold consecutive blocks of
    ``n_local`` rows each (rank order = sample order), and the global row count."""
    d = _dist()
    if d is None:
        return 0, int(n_local)
    sizes = np.zeros(d.get_world_size(), dtype=np.float64)
    sizes[d.get_rank()] = float(n_local)
    all_reduce_sum(sizes)
    sizes = np.rint(sizes).astype(np.int64)
    return int(sizes[:d.get_rank()].sum()), int(sizes.sum())


def global_rows(indices, n_local, fetch, dim):
    """Rows ``indices`` (global sample numbers) of the sharded sample array, identical on every
    rank: the owner of a row contributes it, everybody else zeros, one sum all-reduce.
    ``fetch(local_indices)`` returns this rank's rows as a float64 array."""
    indices = np.asarray(indices, dtype=np.int64)
    offset, _ = shard_offset(n_local)
    out = np.zeros((len(indices), dim), dtype=np.float64)
    mine = (indices >= offset) & (indices < offset + n_local)
    if mine.any():
        out[mine] = np.asarray(fetch(indices[mine] - offset), dtype=np.float64).reshape(-1, dim)
    return all_reduce_sum(out)


def broadcast_from_rank0(values):
    """A float64 vector as rank 0 holds it, on every rank (sum all-reduce of zeros elsewhere)."""
    a = np.array(values, dtype=np.float64)
    if rank() != 0:
        a[...] = 0.0
    return all_reduce_sum(a)
