#!/usr/bin/env python3
"""Which memory the one-shot exchange's mailbox gets, and what a round costs with each kind (verdict r4 #2: "profiles/ note on
what memory type was chosen and why").  Two / four processes share the box's one GPU (gloo carries the handles); for each of
PMC_P2P_MEMORY = (default order) | finegrained | uncached | coarse: pmc_p2p_info's line and the mean time of an
all-reduce of the K = 32, D = 20 statistics vector (7464 doubles) and of the K = 128, D = 40 one (110336), host-synchronised.

    python scripts/p2p_memory_probe.py            (GPU box)
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def worker(rank, world, workdir, kind):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if kind != "default":
        os.environ["PMC_P2P_MEMORY"] = kind
    dist.init_process_group("gloo", init_method="file://" + os.path.join(workdir, "rdzv_" + kind), rank=rank, world_size=world)
    from pypmc_amd import parallel
    used = parallel.enable_p2p_collective(max_doubles=1 << 17, device=0)
    st = parallel.p2p_status()
    line = "world %d  %-12s used=%s  %s" % (world, kind, used, st["info"] or st["reason"])
    if used:
        for n in (7464, 110336):
            t = torch.ones(n, dtype=torch.float64, device="cuda") * (rank + 1)
            for _ in range(20):
                parallel.all_reduce_sum(t)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(200):
                parallel.all_reduce_sum(t)
            torch.cuda.synchronize()
            line += "  | %6d doubles: %.1f us / round" % (n, (time.perf_counter() - t0) / 200 * 1e6)
    if rank == 0:
        print(line, flush=True)
    parallel.disable_p2p_collective()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    for world in (2, 4):
        for kind in ("default", "finegrained", "uncached", "coarse"):
            with tempfile.TemporaryDirectory() as tmp:
                try:
                    mp.spawn(worker, args=(world, tmp, kind), nprocs=world, join=True)
                except Exception as exc:                    # (a kind the runtime refuses for IPC)
                    print("world %d  %-12s FAILED: %s" % (world, kind, str(exc).strip().splitlines()[-1][:160]), flush=True)
