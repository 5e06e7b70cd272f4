#!/usr/bin/env python3
"""Can a caller capture the step in a hipGraph, and what does replaying it save? (GPU box)

    python scripts/graph_probe.py [N]

The bench step (importance-weight pass + VB E-step, K = 32, D = 20) launched eagerly through the library against the
same launches captured once with torch.cuda.CUDAGraph (hipStreamBeginCapture on the launch stream) and replayed:
results must be bit-equal; prints the time per step of both at the given N (default: one GPU's share of a
strong-scaled N = 1e7 over 8, and the full 1e7).
"""
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
    be = HipBackend()
    sizes = [int(a) for a in sys.argv[1:]] or [1_250_000, 10_000_000]
    for N in sizes:
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
        host = torch.empty(stats.numel(), dtype=torch.float64).pin_memory()
        st = torch.cuda.Stream()

        def step():
            r = be.importance_weights(x, prop, tgt, pack=pp, target_pack=pt)
            e = be.estep(x, post, 0, pack=pv, out=stats)
            host.copy_(e["stats"], non_blocking=True)
            return r

        with torch.cuda.stream(st):
            for _ in range(3):
                r = step()
            st.synchronize()
            ref_w, ref_stats = r["weights"].clone(), stats.clone()
            t0 = time.perf_counter()
            for _ in range(30):
                step()
                st.synchronize()
            eager = (time.perf_counter() - t0) / 30
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph, stream=st):
                    rg = step()
            except Exception as exc:                        # noqa: BLE001
                print("N = %d: capture failed: %r" % (N, exc))
                continue
            stats.zero_()
            graph.replay()
            st.synchronize()
            torch.cuda.synchronize()
            same = torch.equal(rg["weights"], ref_w) and torch.equal(stats, ref_stats)
            t0 = time.perf_counter()
            for _ in range(30):
                graph.replay()
                torch.cuda.synchronize()
            replay = (time.perf_counter() - t0) / 30
        print("N = %8d: eager %.3f ms per step, graph replay %.3f ms (%+.1f %%), results bit-equal: %s"
              % (N, eager * 1e3, replay * 1e3, (replay / eager - 1) * 100, same))


if __name__ == "__main__":
    main()
