#!/usr/bin/env python3
"""run() as one library call per stretch between prunings (pmc_vb_state_run) against the loop in variational.py: ms per iteration."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bench import mk
from pypmc_amd.density.mixture import create_gaussian_mixture
from pypmc_amd.mix_adapt.variational import GaussianInference
for K, D, N in ((64, 20, 1_250_000), (64, 20, 156_250), (32, 20, 20_000), (8, 5, 4096)):
    mix = create_gaussian_mixture(*mk(K, D, 3))
    np.random.seed(9)
    x = mix.propose(N, device=True)
    row = []
    for in_lib in (False, True, False, True):
        vb = GaussianInference(x, components=K, initial_guess='first')      # (far from the fit: the iterations do not end early)
        vb.run_in_library = in_lib
        vb.run(3, prune=0.)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = vb.run(40, prune=0.)
        torch.cuda.synchronize()
        row.append("%s %.1f us (%s)" % ("library" if in_lib else "python ", (time.perf_counter() - t0) / (n or 40) * 1e6, n))
    print("K=%d D=%d N=%d:  %s" % (K, D, N, "   ".join(row)), flush=True)
