#!/usr/bin/env python3
"""Kernel micro-benchmark used during development (not the driver's bench.py):
times the three hot kernels on synthetic device-resident data with torch CUDA events."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mk(K, D, seed):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, 3, size=(K, D))
    cov = np.empty((K, D, D))
    for k in range(K):
        A = rs.normal(0, 1, size=(D, D))
        cov[k] = A.dot(A.T) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K)
    return mu, cov, w / w.sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=10_000_000)
    ap.add_argument("--K", type=int, default=32)
    ap.add_argument("--D", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--student", action="store_true")
    args = ap.parse_args()
    import torch
    from pypmc_amd.backend import HipBackend, ComponentSet
    be = HipBackend()
    N, K, D = args.N, args.K, args.D
    mu, cov, w = mk(K, D, 1)
    inv = np.linalg.inv(cov)
    inv = 0.5 * (inv + inv.transpose(0, 2, 1))
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
    if args.student:
        dof = np.full(K, 8.)
        cs = ComponentSet(1, mu, inv, c0=ln, c1=-.5 * (dof + D), c2=1. / dof, c3=dof, weight=w)
    else:
        cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
    g = torch.Generator(device="cuda").manual_seed(1)
    comp = torch.multinomial(torch.tensor(w, device="cuda"), N, replacement=True, generator=g)
    L = torch.tensor(np.linalg.cholesky(cov), device="cuda")
    z = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g)
    x = torch.tensor(mu, device="cuda")[comp]
    for k in range(K):                      # x = mu_k + L_k z, component by component
        sel = (comp == k).nonzero().squeeze(1)
        x[sel] += z[sel] @ L[k].T
    del z
    nu = D + 2. + np.arange(K) * 0.1
    W = inv / nu[:, None, None]
    vb = ComponentSet(2, mu, W, c0=D / (1. + np.arange(K)), c1=nu, c2=np.log(w),
                      c3=np.linalg.slogdet(W)[1] + 3.)
    pack, vpack = be.pack(cs), be.pack(vb)
    lt = torch.zeros(N, dtype=torch.float64, device="cuda")

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return min(ts), float(np.median(ts))

    flops_pair = D * D + 4 * D + 40
    res = {}
    t, tm = timeit(lambda: be.logpdf(x, cs, pack=pack))
    res["logpdf"] = dict(ms=t, ms_median=tm, samples_per_s=N / t * 1e3,
                         tflops=N * K * flops_pair / t * 1e-9)
    t, tm = timeit(lambda: be.logpdf(x, cs, pack=pack, log_target=lt, want_scalars=True))
    res["logpdf+is"] = dict(ms=t, ms_median=tm, samples_per_s=N / t * 1e3)
    out = be.zeros(be.stats_len(K, D))
    t, tm = timeit(lambda: be.estep(x, vb, 0, pack=vpack, out=out))
    fl_vb = K * (D * D + 4 * D) + K * (1 + 2 * D + D * (D + 1)) + K * 40
    res["vb_estep"] = dict(ms=t, ms_median=tm, samples_per_s=N / t * 1e3, tflops=N * fl_vb / t * 1e-9,
                           fused=bool(be.lib.pmc_estep_is_fused(K, D, 2, 0)))
    # the statistics half as pmc_estep runs it (the component x monomial form where it applies): library timing
    be.kernel_timings()
    be.kernel_timing(True)
    for _ in range(args.reps):
        be.estep(x, vb, 0, pack=vpack, out=out)
    torch.cuda.synchronize()
    be.kernel_timing(False)
    kt = be.kernel_timings()
    for name in ("k_resp", "k_stats", "k_estep_fused", "finishing reductions"):
        if name in kt:
            ms = kt[name]["ms"] / max(kt[name]["calls"], 1)
            res["estep:" + name] = dict(ms=ms, ms_median=ms, tflops=kt[name]["flops"] / max(kt[name]["calls"], 1) / ms * 1e-9)
    # split: responsibilities alone
    lib = be.lib
    u = be._tilebuf("u", N, K)
    ws = be._workspace(N, K, D)
    P = be._p
    t, tm = timeit(lambda: lib.pmc_responsibilities(P(x), N, D, P(vpack), K, 2, 0, 0, P(None), P(None), P(u),
                                                    P(None), P(None), P(None), P(None), P(None), K, P(out), P(ws),
                                                    be._stream()))
    res["vb_resp_only"] = dict(ms=t, ms_median=tm)
    t, tm = timeit(lambda: lib.pmc_sufficient_stats(P(x), N, D, P(vpack), K, P(u), P(out[8:]), P(ws),
                                                    be._stream()))
    res["vb_stats_only"] = dict(ms=t, ms_median=tm)
    print(json.dumps(dict(N=N, K=K, D=D, student=args.student, **res), indent=1))


if __name__ == "__main__":
    main()
