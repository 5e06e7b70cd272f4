#!/usr/bin/env python3
"""Run-to-run determinism of the hot path: every kernel reduces in a fixed order (per-tile partials, summed
by the finishing kernels in index order), so the same inputs must give the same BITS on every launch.  A
missing barrier or an unordered reduction shows up here as a count of launches that differ from the first.

    python scripts/determinism_stress.py [--reps 300]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


CASES = [(4, 5, 50_000), (4, 3, 50_000), (2, 7, 33_333), (7, 32, 100_001), (5, 9, 1_000_000),
         (8, 6, 77_777), (12, 16, 100_000), (20, 32, 300_000), (32, 31, 123_457), (40, 128, 200_000),
         (70, 5, 40_000), (130, 3, 20_001)]          # the last two: the run-time-dimension unit


def sweep(reps, cases=CASES, verbose=True):
    """Number of launches whose result differed bitwise from the first launch on the same inputs."""
    import torch
    import pypmc_amd as pypmc
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd._lib import PMC_KIND_GAUSS, PMC_KIND_STUDENT_T, PMC_KIND_VB, PMC_RESP_VB, PMC_RESP_PMC_RB
    be = pypmc.backend.get_backend()
    bad = 0
    for D, K, N in cases:
        rs = np.random.RandomState(D * 1000 + K)
        mu = rs.normal(0, 2, (K, D))
        prec = np.empty((K, D, D))
        for k in range(K):
            A = rs.normal(0, 1, (D, D))
            prec[k] = np.linalg.inv(A.dot(A.T) / D + 0.5 * np.eye(D))
        x = torch.as_tensor(rs.normal(0, 2.5, (N, D)), device=be.device)
        sw = torch.as_tensor(rs.uniform(0.5, 1.5, N), device=be.device)
        ln = rs.normal(-3, 1, K)
        for kind, mode, label in ((PMC_KIND_VB, PMC_RESP_VB, "vb"), (PMC_KIND_GAUSS, PMC_RESP_PMC_RB, "pmc"),
                                  (PMC_KIND_STUDENT_T, PMC_RESP_PMC_RB, "pmc-t")):
            if kind == PMC_KIND_VB:
                cs = ComponentSet(kind, mu, prec, np.full(K, D / 50.), np.full(K, D + 3.), rs.normal(-2, .3, K), ln)
            elif kind == PMC_KIND_GAUSS:
                cs = ComponentSet(kind, mu, prec, ln, weight=np.full(K, 1.0 / K))
            else:
                cs = ComponentSet(kind, mu, prec, ln, np.full(K, -.5 * (5. + D)), np.full(K, .2), np.full(K, 5.),
                                  weight=np.full(K, 1.0 / K))
            first = None
            differ = 0
            for rep in range(reps):
                out = be.estep(x, cs, mode, sample_w=sw)["stats"]
                got = out.clone()
                if first is None:
                    first = got
                elif not torch.equal(first.view(torch.int64), got.view(torch.int64)):
                    differ += 1
            fused = bool(be.lib.pmc_estep_is_fused(K, D, kind, int(mode)))
            if verbose:
                print("D=%-3d K=%-4d N=%-8d %-6s %-6s launches differing from the first: %d / %d"
                      % (D, K, N, label, "fused" if fused else "2-kern", differ, reps - 1), flush=True)
            bad += differ
            # the log-pdf too
            first = None
            differ = 0
            for rep in range(0 if kind == PMC_KIND_VB else reps // 3):
                got = be.logpdf(x, cs)["out"].clone()
                if first is None:
                    first = got
                elif not torch.equal(first.view(torch.int64), got.view(torch.int64)):
                    differ += 1
            if differ:
                print("   logpdf differs: %d" % differ)
            bad += differ
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    bad = sweep(args.reps)
    print("TOTAL differing launches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
