#!/usr/bin/env python3
"""Host <-> device rates of the handle layer (include/pmc_ctx.h) for a large sample array: pmc_samples_upload /
pmc_samples_download / pmc_mix_logpdf with the page-locked view on and off (PMC_CTX_PIN_BYTES), GPU box.

    python scripts/ctx_transfer_bench.py [N]
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[2] == "child":
    from pypmc_amd import _lib
    lib = _lib.load()
    N, D, K = int(sys.argv[1]), 20, 16
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rs = np.random.RandomState(0)
    x = rs.normal(size=(N, D))
    mu = rs.normal(0, 3, (K, D))
    inv = np.ascontiguousarray(np.tile(np.eye(D), (K, 1, 1)))
    ln = np.full(K, -0.5 * D * np.log(2 * np.pi))
    w = np.full(K, 1. / K)
    ctx, mix, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.pmc_init(0, C.byref(ctx)) == 0
    assert lib.pmc_mixture_create(ctx, 0, K, D, dp(w), dp(mu), dp(inv), dp(ln), None, C.byref(mix)) == 0
    out = np.empty(N)
    for rep in range(3):
        t0 = time.perf_counter()
        assert lib.pmc_samples_upload(ctx, dp(x), N, D, C.byref(s)) == 0
        t1 = time.perf_counter()
        assert lib.pmc_mix_logpdf(mix, s, dp(out), None) == 0
        t2 = time.perf_counter()
        back = np.empty_like(x)
        assert lib.pmc_samples_download(s, dp(back)) == 0
        t3 = time.perf_counter()
        lib.pmc_samples_free(s)
        gb = x.nbytes * 1e-9
        print("  rep %d: upload %.1f ms (%.1f GB/s)  logpdf + %d MB back %.1f ms  download %.1f ms (%.1f GB/s)"
              % (rep, (t1 - t0) * 1e3, gb / (t1 - t0), out.nbytes >> 20, (t2 - t1) * 1e3, (t3 - t2) * 1e3, gb / (t3 - t2)))
    assert np.array_equal(back, x)
    lib.pmc_mixture_destroy(mix)
    # the headline's step through the handle layer: importance weights against a mixture target (nothing N-sized comes
    # back), then the VB E-step -- host arrays in, K-sized host arrays out, synchronous
    sys.path.insert(0, ROOT)
    from bench import mk, gauss_params, vb_params
    K, KT = 32, 4
    mu, cov, w = mk(K, D, 1)
    tmu, tcov, tw = mk(KT, D, 11)
    inv, ln = gauss_params(mu, cov)
    tinv, tln = gauss_params(tmu, tcov)
    W, beta, nu, ln_pi, ln_lambda = [np.ascontiguousarray(a) for a in vb_params(mu, cov, w, N)]
    x = np.ascontiguousarray(mu[rs.randint(0, K, N)] + rs.normal(size=(N, D)))
    q, t = C.c_void_p(), C.c_void_p()
    assert lib.pmc_mixture_create(ctx, 0, K, D, dp(w), dp(mu), dp(np.ascontiguousarray(inv)), dp(ln), None, C.byref(q)) == 0
    assert lib.pmc_mixture_create(ctx, 0, KT, D, dp(tw), dp(tmu), dp(np.ascontiguousarray(tinv)), dp(tln), None, C.byref(t)) == 0
    assert lib.pmc_samples_upload(ctx, dp(x), N, D, C.byref(s)) == 0
    sums, Nk, xbar, S, elq = np.empty(3), np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
    for rep in range(4):
        t0 = time.perf_counter()
        assert lib.pmc_is_weights(q, s, None, t, None, None, dp(sums)) == 0
        t1 = time.perf_counter()
        assert lib.pmc_vb_estep(ctx, s, None, K, dp(mu), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lambda), None,
                                dp(Nk), dp(xbar), dp(S), dp(elq), None, None) == 0
        t2 = time.perf_counter()
        print("  rep %d: pmc_is_weights %.2f ms  pmc_vb_estep %.2f ms  -> %.3g samples/s through both (N = %d, K = %d, D = %d)"
              % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, N / (t2 - t0), N, K, D))
    assert abs(Nk.sum() - N) < 1e-6 * N
    lib.pmc_samples_free(s)
    lib.pmc_mixture_destroy(q)
    lib.pmc_mixture_destroy(t)
    lib.pmc_shutdown(ctx)
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "10000000"
    for label, env in (("page-locked view (default)", {}), ("plain pageable copies", {"PMC_CTX_PIN_BYTES": str(1 << 62)})):
        print(label)
        subprocess.run([sys.executable, os.path.abspath(__file__), N, "child"], env=dict(os.environ, **env), check=True)
