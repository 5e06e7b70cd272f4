#!/usr/bin/env python3
"""Host overhead of one GaussianInference.run() iteration (update + bound + prune) with the K-sized state on the device:
a batch so small that the kernels do not matter, profiled by function."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_vb_state import _data, _fit
K, D, N = 64, 20, 4096
x = _data(N, D, 8, 1)
vb = _fit(x, K, True)


def iteration():
    vb.update()
    vb.likelihood_bound()
    vb.prune(0.0001)


for _ in range(20):
    iteration()
t0 = time.perf_counter()
for _ in range(200):
    iteration()
print("us per iteration at N = %d: %.1f" % (N, (time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    iteration()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
