#!/usr/bin/env python3
"""Timings of the run-time-dimension unit (sample dimensions beyond 64, csrc/pmc_big.hip): mixture log-pdf,
responsibilities, statistics and propose per dimension, with the compiled D = 64 unit beside them.

    python scripts/bigdim_bench.py [--N 1000000] [--K 32] > profiles/r02_big_dims.txt
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--K", type=int, default=32)
    ap.add_argument("--dims", type=int, nargs="*", default=[64, 72, 96, 100, 128, 200, 256, 512])
    args = ap.parse_args()
    import torch
    from pypmc_amd.backend import HipBackend
    be = HipBackend()
    print("# N = %d samples, K = %d components; ms per call (best of 3), algorithmic TFLOP/s in brackets" % (args.N, args.K))
    print("# %4s %18s %18s %18s %12s" % ("D", "mixture log-pdf", "responsibilities", "statistics", "propose"))
    for D in args.dims:
        N = args.N if D <= 256 else args.N // 4
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kbench.py"), "--N", str(N), "--K", str(args.K),
                              "--D", str(D), "--reps", "3"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        r = json.loads(out[out.index("{"):])
        pair = D * D + 4 * D + 40
        stat = 1 + 2 * D + D * (D + 1)
        tf = lambda ms, fl: N * args.K * fl / ms * 1e-9
        # propose
        rs = np.random.RandomState(0)
        mu = rs.normal(size=(args.K, D))
        A = rs.normal(size=(args.K, D, D))
        chol = np.linalg.cholesky(np.einsum('kij,klj->kil', A, A) / D + 0.5 * np.eye(D))
        counts = np.full(args.K, N // args.K)
        x = be.empty((int(counts.sum()), D))
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            be.propose(mu, chol, None, counts, seed=1, out=x, want_origin=False)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print("  %4d %9.2f (%5.1f) %9.2f (%5.1f) %9.2f (%5.1f) %9.2f   N = %d" % (
            D, r["logpdf"]["ms"], tf(r["logpdf"]["ms"], pair), r["vb_resp_only"]["ms"], tf(r["vb_resp_only"]["ms"], pair),
            r["vb_stats_only"]["ms"], tf(r["vb_stats_only"]["ms"], stat), min(ts[1:]), N), flush=True)


if __name__ == "__main__":
    main()
