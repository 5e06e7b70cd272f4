#!/usr/bin/env python3
"""The matrix-product form of the Mahalanobis forms (csrc/pmc_mgemm.hip) against the exact kernels on one GPU:
differences, the guard's error / bound ratio, kernel times.

    python scripts/mgemm_check.py [quick]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from pypmc_amd.backend import HipBackend  # noqa: E402
from test_gpu_kernels import mk, gauss_set, student_set  # noqa: E402

be = HipBackend()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def device_samples(mu, cov, w, N, seed):
    """samples of the mixture, drawn on the device"""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    mix = create_gaussian_mixture(mu, cov, w)
    return mix.propose(N, np.random.mtrand.RandomState(seed), device=True)


def both(fn):
    be.configure("maha_gemm_tolerance", 0.0)
    exact = fn()
    be.configure("maha_gemm_tolerance", 5e-11)
    gemm = fn()
    return exact, gemm


for D, K, N in ([(40, 128, 400000)] if quick else [(32, 32, 2000000), (32, 128, 1000000), (40, 32, 2000000),
                                                    (40, 128, 2000000), (40, 96, 1000000), (48, 64, 1000000), (44, 32, 1000000), (36, 64, 1000000)]):
    mu, cov, w = mk(K, D, 5)
    x = device_samples(mu, cov, w, N, 7)
    for fam in ("gauss", "student"):
        comps = gauss_set(mu, cov, w)[0] if fam == "gauss" else student_set(mu, cov, w, np.full(K, 8.))[0]
        ex, ge = both(lambda: be.tohost(be.logpdf(x, comps, want_scalars=True)["out"]))
        err = np.abs(ex - ge)
        rep = be.maha_gemm_report(N, K, D)
        if rep:
            # the guard's bound per sample (host restatement) against the difference actually seen
            cen = 0.5 * (mu.min(axis=0) + mu.max(axis=0))
            dn = np.linalg.norm(be.tohost(x[:200000]) - cen, axis=1)
            bound = 1e-15 * (rep["norms"][0] * dn ** 2 + rep["norms"][1] * dn + rep["norms"][2])
            print("   guard: norms %s  refused %d of %d workgroups;  bound (first 2e5 samples) median %.2e max %.2e;  "
                  "largest |difference| / bound %.3f" % (np.array2string(rep["norms"], precision=3), rep["refused"],
                                                        rep["workgroups"], np.median(bound), bound.max(),
                                                        (err[:200000] / bound).max()), flush=True)
        t_ex, t_ge = both(lambda: timeit(lambda: be.logpdf(x, comps, want_scalars=True)))
        print("D=%d K=%d N=%d %-7s log q: max |gemm - exact| %.3e (rel %.3e)   exact %.3f ms  gemm %.3f ms  (%.2f -> %.2f ps/pair)"
              % (D, K, N, fam, err.max(), (err / np.abs(ex)).max(), t_ex, t_ge, t_ex * 1e9 / (N * K), t_ge * 1e9 / (N * K)),
              flush=True)
    # importance weights + emitted responsibilities + statistics (configuration 5's pair of calls)
    prop = gauss_set(mu, cov, w)[0]
    target = gauss_set(*mk(4, D, 11))[0]

    def step():
        em = be.importance_weights(x, prop, target, emit=True)
        st = be.estep_from_u(x, prop, em["responsibilities"])
        return em, st
    (em_e, st_e), (em_g, st_g) = both(step)
    we, wg = be.tohost(em_e["weights"]), be.tohost(em_g["weights"])
    se, sg = be.tohost(st_e["stats"]), be.tohost(st_g["stats"])
    ps = 1 + D + D * (D + 1) // 2
    a, b = se[8:8 + K * ps].reshape(K, ps), sg[8:8 + K * ps].reshape(K, ps)
    scale = np.abs(a).max(axis=1, keepdims=True) + 1e-300
    print("   IS weights rel diff %.3e   scalars rel diff %.3e   statistics (per component scale) %.3e"
          % ((np.abs(we - wg) / np.abs(we)).max(), (np.abs(be.tohost(em_e["scalars"]) - be.tohost(em_g["scalars"]))[:4]
                                                    / np.abs(be.tohost(em_e["scalars"]))[:4]).max(),
             (np.abs(a - b) / scale).max()), flush=True)
    ue, ug = em_e["responsibilities"].host_matrix(be)[:20000], em_g["responsibilities"].host_matrix(be)[:20000]
    big = ue > 1e-200
    print("   u = w rho (first 20000 samples): rel diff %.3e" % (np.abs(ue - ug)[big] / ue[big]).max(), flush=True)
    be.kernel_timing(True)
    for tol in (0.0, 5e-11):
        be.configure("maha_gemm_tolerance", tol)
        step()
        be.kernel_timings()
        for _ in range(3):
            step()
        tm = be.kernel_timings()
        print("   tol %g: " % tol + "  ".join("%s %.3f ms" % (k, v["ms"] / 3) for k, v in tm.items()), flush=True)
    be.kernel_timing(False)
    # the E-step (VB) through pmc_estep
    from scipy.special import digamma
    inv = np.linalg.inv(cov)
    inv = 0.5 * (inv + inv.transpose(0, 2, 1))
    rs = np.random.RandomState(3)
    nu, beta, alpha = D + 2. + rs.uniform(0, 3, K), 1. + rs.uniform(0, 3, K), 1. + rs.uniform(0, 3, K)
    W = inv / nu[:, None, None]
    ln_lambda = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + np.linalg.slogdet(W)[1]
    ln_pi = digamma(alpha) - digamma(alpha.sum())
    from pypmc_amd.backend import ComponentSet
    cs = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    ee, eg = both(lambda: be.tohost(be.estep(x, cs, 0)["stats"]))
    a, b = ee[8:8 + K * ps].reshape(K, ps), eg[8:8 + K * ps].reshape(K, ps)
    scale = np.abs(a).max(axis=1, keepdims=True) + 1e-300
    rep = be.maha_gemm_report(N, K, D)
    if rep:
        print("   VB guard: norms %s refused %d of %d" % (np.array2string(rep["norms"], precision=3), rep["refused"], rep["workgroups"]))
    t_ex, t_ge = both(lambda: timeit(lambda: be.estep(x, cs, 0)))
    print("   VB E-step: E[log q(Z)] rel diff %.3e  statistics %.3e   exact %.3f ms  gemm %.3f ms"
          % (abs(ee[0] - eg[0]) / abs(ee[0]), (np.abs(a - b) / scale).max(), t_ex, t_ge), flush=True)
