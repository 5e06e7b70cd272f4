#!/usr/bin/env python3
"""Where a PMC / VB iteration's wall time goes at mid sizes: device kernels vs the K-sized host update.

    python scripts/hostshare_bench.py [--profile]      # --profile: cProfile of the D=40, K=128 case
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    import torch
    from pypmc_amd.backend import get_backend
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    from pypmc_amd.mix_adapt.variational import GaussianInference
    be = get_backend()
    for N, D, K in ((1_000_000, 20, 32), (1_000_000, 40, 128), (100_000, 10, 8)):
        mu, cov, w = bench.mk(K, D, 1)
        mix = create_gaussian_mixture(mu, cov, w)
        np.random.seed(1)
        x = mix.propose(N, device=True)
        iw = torch.rand(N, dtype=torch.float64, device=be.device) + 0.5

        def pmc():
            gaussian_pmc(x, mix, iw, copy=True)
        vb = GaussianInference(x, initial_guess=mix)
        for name, fn in (("gaussian_pmc", pmc), ("VB update", vb.update), ("VB bound", vb.likelihood_bound)):
            fn()
            torch.cuda.synchronize()
            be.kernel_timings()
            be.kernel_timing(True)
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 5 * 1e3
            be.kernel_timing(False)
            dev = sum(v["ms"] for v in be.kernel_timings().values()) / 5
            print("N=%-8d D=%-3d K=%-4d %-13s wall %7.2f ms   device kernels %7.2f ms   host %7.2f ms"
                  % (N, D, K, name, wall, dev, wall - dev), flush=True)
            if args.profile and K == 128 and name != "VB bound":
                import cProfile
                import pstats
                pr = cProfile.Profile()
                pr.enable()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                pr.disable()
                pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
