"""Multi-GPU plumbing: one process per GPU (torchrun), samples sharded by rank, and exactly one
collective on the hot path -- a sum all-reduce of the K-sized statistics vector
(torch.distributed: backend "nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU test-suite).

The reference's only distributed code gathers whole pickled sample histories to rank 0 with
mpi4py (pypmc/tools/parallel_sampler.py:58-71) and broadcasts the adapted proposal back
(examples/pmc_mpi.py:119-131).  Here samples never leave their GPU: every rank reduces its shard to
[scalars | K x (1 + D + D(D+1)/2) | K x 2] doubles, the ranks all-reduce that buffer, and every rank
runs the identical K-sized host update -- no broadcast is needed.
"""
import numpy as np


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


def shard_bounds(N, r=None, world=None):
    """[begin, end) of rank ``r``'s contiguous block of N samples (sizes differ by at most 1)."""
    r = rank() if r is None else r
    world = world_size() if world is None else world
    base, extra = divmod(int(N), world)
    begin = r * base + min(r, extra)
    return begin, begin + base + (1 if r < extra else 0)


def all_reduce_sum(buf):
    """In-place sum over ranks of a float64 buffer (CUDA tensor -> RCCL, CPU tensor / numpy ->
    gloo).  Returns ``buf``.  A no-op for a single process."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return buf
    import torch
    if isinstance(buf, np.ndarray):
        t = torch.from_numpy(buf)          # shares memory with buf
        d.all_reduce(t, op=d.ReduceOp.SUM)
        return buf
    d.all_reduce(buf, op=d.ReduceOp.SUM)
    return buf


def all_reduce_scalars(*values):
    """Sum a few python floats over ranks (setup-time bookkeeping: global N, global sum of weights).
    Uses a tensor on the device the process group's backend expects."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return tuple(float(v) for v in values)
    import torch
    dev = "cuda" if d.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return tuple(float(v) for v in t.cpu())
