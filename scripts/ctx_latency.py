#!/usr/bin/env python3
"""Latency of an E-step at an 8-GPU shard size through the two host sides (GPU box): GaussianInference.E_step of the Python
front-end against pmc_vb_estep of the handle layer (C++ host code), same samples, same parameters, D = 20.

    python scripts/ctx_latency.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import mk  # noqa: E402
from pypmc_amd import _lib  # noqa: E402
from pypmc_amd.density.mixture import create_gaussian_mixture  # noqa: E402
from pypmc_amd.mix_adapt.variational import GaussianInference  # noqa: E402

lib = _lib.load()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
D = 20
for K, N in ((64, 1_250_000), (32, 1_250_000), (8, 100_000), (64, 10_000_000)):
    mix = create_gaussian_mixture(*mk(K, D, 3))
    np.random.seed(9)
    x = mix.propose(N)
    vb = GaussianInference(x, initial_guess=mix)
    for _ in range(5):
        vb.E_step()
    torch.cuda.synchronize()
    reps = 40 if N <= 2_000_000 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        vb.E_step()
    torch.cuda.synchronize()
    t_py = (time.perf_counter() - t0) / reps
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in
            (vb.m, vb.W, vb.nu, vb.beta, vb.expectation_ln_pi, vb.expectation_det_ln_lambda, vb.x_mean_comp)]
    ctx, s = C.c_void_p(), C.c_void_p()
    assert lib.pmc_init(0, C.byref(ctx)) == 0
    assert lib.pmc_samples_upload(ctx, dp(x), N, D, C.byref(s)) == 0
    Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)

    def call():
        assert lib.pmc_vb_estep(ctx, s, None, K, dp(arrs[0]), dp(arrs[1]), dp(arrs[2]), dp(arrs[3]), dp(arrs[4]), dp(arrs[5]),
                                dp(arrs[6]), dp(Nk), dp(xbar), dp(S), dp(elq), None, None) == 0
    for _ in range(5):
        call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    t_c = (time.perf_counter() - t0) / reps
    np.testing.assert_allclose(Nk, vb.N_comp, rtol=1e-11)
    print("D = %d K = %3d N = %8d: GaussianInference.E_step %.3f ms   pmc_vb_estep %.3f ms" % (D, K, N, t_py * 1e3, t_c * 1e3))
    lib.pmc_samples_free(s)
    lib.pmc_shutdown(ctx)
    del vb, x
