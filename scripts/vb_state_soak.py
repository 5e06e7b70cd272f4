#!/usr/bin/env python3
"""Soak of the device-resident VB state: random problems, GaussianInference.run() on the device path and on the host path
(K-sized work through LAPACK / scipy / numpy as in rounds 1-5); iteration counts, surviving K, final bound and posterior."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_vb_state import _data, _fit
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = dict(bound=0.0, m=0.0, W=0.0, alpha=0.0)
bad = 0
t_dev = t_host = 0.0
for case in range(n_cases):
    rng = np.random.RandomState(5000 + case)
    D = int(rng.choice([1, 2, 3, 5, 8, 13, 20, 32, 40, 64]))
    K = int(rng.randint(2, 30))
    N = int(rng.randint(max(4 * K, 200), 60000))
    x = _data(N, D, int(rng.randint(1, 6)), case, spread=float(rng.uniform(3, 12)))
    kw = {}
    if case % 4 == 0:
        kw["weights"] = rng.uniform(0.05, 3.0, size=N)
    if case % 5 == 0:
        kw.update(alpha0=float(rng.uniform(1e-3, 1)), beta0=float(rng.uniform(1e-3, 1)), nu0=D + float(rng.uniform(0, 3)))
    dev, host = _fit(x, K, True, **kw), _fit(x, K, False, **kw)
    t0 = time.perf_counter(); nd = dev.run(60, verbose=False); t_dev += time.perf_counter() - t0
    t0 = time.perf_counter(); nh = host.run(60, verbose=False); t_host += time.perf_counter() - t0
    ok = dev.K == host.K and (nd is None) == (nh is None) and (nd is None or abs(nd - nh) <= 2)
    bd, bh = dev.likelihood_bound(), host.likelihood_bound()
    eb = abs(bd - bh) / abs(bh)
    errs = dict(bound=eb)
    if dev.K == host.K:
        for name in ("m", "W", "alpha"):
            a, b = np.asarray(getattr(dev, name)), np.asarray(getattr(host, name))
            errs[name] = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
    for k_, v in errs.items():
        worst[k_] = max(worst[k_], v)
    ok = ok and all(v < 1e-7 for v in errs.values())
    bad += not ok
    print("case %2d  D=%2d K=%2d->%2d/%2d N=%6d  iterations %s / %s  %s  %s" % (
        case, D, K, dev.K, host.K, N, nd, nh, "  ".join("%s %.1e" % kv for kv in errs.items()), "ok" if ok else "DIFFERENT"), flush=True)
print("cases %d, different %d; worst relative differences %s" % (n_cases, bad, worst))
print("run() wall time, all cases: device state %.2f s, K-sized work on the host %.2f s" % (t_dev, t_host))
