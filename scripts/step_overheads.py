#!/usr/bin/env python3
"""What the measuring itself costs a bench step (GPU box): the step of bench.py (N = 1e7, K = 32, D = 20) with and without
the library's kernel-timing events, the five phase events, and with the K-sized result copied by `.cpu()` (pageable,
synchronous) or into a pinned buffer (asynchronous + one stream synchronisation).

    python scripts/step_overheads.py [N]
"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from bench import mk, gauss_params, vb_params, K, D, K_T
    from pypmc_amd.backend import HipBackend, ComponentSet
    from pypmc_amd import parallel
    be = HipBackend()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    mu, cov, w = mk(K, D, 1)
    tmu, tcov, tw = mk(K_T, D, 11)
    inv, ln = gauss_params(mu, cov)
    tinv, tln = gauss_params(tmu, tcov)
    W, beta, nu, ln_pi, ln_lambda = vb_params(mu, cov, w, N)
    prop = ComponentSet(0, mu, inv, c0=ln, weight=w)
    tgt = ComponentSet(0, tmu, tinv, c0=tln, weight=tw)
    post = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    pp, pt, pv = be.pack(prop), be.pack(tgt), be.pack(post)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g) * 1.1
    x += torch.tensor(mu, device="cuda")[torch.randint(0, K, (N,), device="cuda", generator=g)]
    stats = be.zeros(be.stats_len(K, D))
    pinned = torch.empty(stats.numel(), dtype=torch.float64).pin_memory()

    def run(lib_timing, phase_events, pinned_copy, steps=20):
        def step():
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if phase_events else None
            if evs: evs[0].record()
            be.importance_weights(x, prop, tgt, pack=pp, target_pack=pt)
            if evs: evs[1].record()
            e = be.estep(x, post, 0, pack=pv, out=stats)
            if evs: evs[2].record()
            flat = parallel.all_reduce_sum(e["stats"])
            if evs: evs[3].record()
            if pinned_copy:
                pinned.copy_(flat, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            else:
                flat.cpu()
            if evs: evs[4].record()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        be.kernel_timings()
        be.kernel_timing(lib_timing)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        be.kernel_timing(False)
        be.kernel_timings()
        return dt * 1e3

    print("N = %d; ms per step" % N)
    print("%-14s %-14s %-14s %8s" % ("library events", "phase events", "result copy", "ms"))
    for lt, pe, pc in itertools.product((True, False), (True, False), (False, True)):
        ms = min(run(lt, pe, pc) for _ in range(2))
        print("%-14s %-14s %-14s %8.3f" % ("on" if lt else "off", "on" if pe else "off", "pinned async" if pc else ".cpu()", ms))


if __name__ == "__main__":
    main()
