#!/usr/bin/env python3
"""GaussianInference(devices=[...]) -- one process, the data sharded over the devices of a group (here: virtual shards of the one
GPU): one run() iteration (update + bound + prune) with the K-sized state on the group's first device against the K-sized work
on the host (pmc_vb_estep per E-step, LAPACK / scipy / numpy between two of them)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bench import mk
from pypmc_amd.density.mixture import create_gaussian_mixture
from pypmc_amd.mix_adapt.variational import GaussianInference
from pypmc_amd.devices import DeviceGroup
K, D = 64, 20
mix = create_gaussian_mixture(*mk(K, D, 3))
for N, devices in ((1_250_000, [0]), (1_250_000, [0, 0]), (1_250_000, [0, 0, 0, 0]), (10_000_000, [0, 0, 0, 0, 0, 0, 0, 0])):
    np.random.seed(9)
    x = mix.propose(N)
    with DeviceGroup(devices) as g:
        row = []
        for on_device in (True, False, True, False):
            vb = GaussianInference.__new__(GaussianInference)
            vb.device_update = on_device
            vb.__init__(x, initial_guess=mix, devices=g)

            def iteration():
                vb.update()
                vb.likelihood_bound()
                vb.prune()
            for _ in range(5):
                iteration()
            t0 = time.perf_counter()
            for _ in range(20):
                iteration()
            row.append("%s %.3f ms" % ("state on the first device" if on_device else "K-sized work on the host", (time.perf_counter() - t0) / 20 * 1e3))
            del vb
        print("N=%d over %d (virtual) devices:  %s" % (N, len(devices), "   ".join(row)), flush=True)
