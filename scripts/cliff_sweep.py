#!/usr/bin/env python3
"""Cost per (sample, component) pair of the log-pdf pass and of the E-step across the (D, K) plane (GPU box):
looks for cliffs -- shapes that cost much more per pair than their neighbours -- rather than for peak numbers.

    python scripts/cliff_sweep.py [--N 1000000]

Prints picoseconds per pair (library timing, best of 3) for pmc_mixture_logpdf and pmc_estep (VB kind), and flags
entries more than 1.35 x the cheaper of their two K-neighbours of the same D (K >= 4: below that the fixed per-sample
work -- the row load, the log-sum-exp tail, the scalars -- dominates by construction).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.kbench import mk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--dims", default="2,5,8,10,12,16,20,24,30,32,40")
    ap.add_argument("--ks", default="1,2,3,4,6,8,12,16,17,20,24,32,33,40,48,64,65,96,128")
    args = ap.parse_args()
    import torch
    from pypmc_amd.backend import HipBackend, ComponentSet
    be = HipBackend()
    N = args.N
    dims = [int(d) for d in args.dims.split(",")]
    ks = [int(k) for k in args.ks.split(",")]
    table = {}
    for D in dims:
        g = torch.Generator(device="cuda").manual_seed(D)
        x = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g) * 3.0
        for K in ks:
            mu, cov, w = mk(K, D, 1)
            inv = np.linalg.inv(cov)
            inv = 0.5 * (inv + inv.transpose(0, 2, 1))
            ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
            cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
            nu = D + 2. + np.arange(K) * 0.1
            W = inv / nu[:, None, None]
            vb = ComponentSet(2, mu, W, c0=D / (1. + np.arange(K)), c1=nu, c2=np.log(w), c3=np.linalg.slogdet(W)[1] + 3.)
            pack, vpack = be.pack(cs), be.pack(vb)
            out = be.zeros(be.stats_len(K, D))

            def best(fn, names):
                fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    be.kernel_timings()
                    be.kernel_timing(True)
                    fn()
                    torch.cuda.synchronize()
                    be.kernel_timing(False)
                    kt = be.kernel_timings()
                    ts.append(sum(v["ms"] for k_, v in kt.items() if k_ in names))
                return min(ts)
            t_l = best(lambda: be.logpdf(x, cs, pack=pack), ("k_logpdf", "finishing reductions"))
            t_e = best(lambda: be.estep(x, vb, 0, pack=vpack, out=out),
                       ("k_resp", "k_stats", "k_estep_fused", "finishing reductions"))
            table[(D, K)] = (t_l * 1e9 / (N * K), t_e * 1e9 / (N * K))
            del pack, vpack, out
        del x
        torch.cuda.empty_cache()
    for which, name in ((0, "log-pdf"), (1, "E-step")):
        print("%s, ps per (sample, component), N = %d" % (name, N))
        print("   D \\ K " + "".join("%8d" % k for k in ks))
        for D in dims:
            row = ""
            for i, K in enumerate(ks):
                v = table[(D, K)][which]
                nb = [table[(D, ks[j])][which] for j in (i - 1, i + 1) if 0 <= j < len(ks)]
                flag = "*" if (K >= 4 and nb and v > 1.35 * min(nb)) else " "
                row += "%7.1f%s" % (v, flag)
            print("%8d " % D + row)
        print()


if __name__ == "__main__":
    main()
