#!/usr/bin/env python3
"""Oracle-vs-reference timing on the host (SURVEY.md 8(d) "CPU baseline", step 1).

BUILD CONTAINER ONLY: needs /root/reference.  Builds the reference's Cython extensions in a scratch
copy (the recipe of tests/golden/make_golden.py), then times the REAL reference and the C oracle
(oracle/libpmc_oracle.so, single thread) on the same inputs at the SURVEY section 6 shapes and writes
the ratio table to profiles/r06_cpu_ratio.json (round 2's: r02_cpu_ratio.json).  The ratio ties bench.py's `cpu_baseline` (the oracle
timed on the GPU box's host, where the reference cannot travel) to the reference itself.

    python scripts/cpu_ratio.py [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def mk(K, D, seed):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, 3, size=(K, D))
    cov = np.empty((K, D, D))
    for k in range(K):
        A = rs.normal(0, 1, size=(D, D))
        cov[k] = A.dot(A.T) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K)
    return mu, cov, w / w.sum()


def draw(mu, cov, w, N, seed):
    rs = np.random.RandomState(seed)
    k = rs.choice(len(w), size=N, p=w)
    L = np.linalg.cholesky(cov)
    return mu[k] + np.einsum('nij,nj->ni', L[k], rs.normal(size=(N, mu.shape[1])))


def gauss_ln(mix):
    """log normalisation of the reference's Gauss components (gauss.pyx:56; not a public attribute)"""
    return np.array([-0.5 * c.dim * np.log(2 * np.pi) - 0.5 * c.log_det_sigma for c in mix.components])


def best(fn, repeat=3):
    t = []
    for _ in range(repeat):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-build", default="/tmp/pypmc_ref")
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies every N")
    ap.add_argument("--out", default="r06_cpu_ratio.json", help="file name under profiles/")
    args = ap.parse_args()
    import make_golden
    make_golden.ensure_reference(args.ref_build)
    warnings.simplefilter("ignore")
    import pypmc
    from oracle import oracle as orc
    from scipy.special import gammaln
    orc.build()
    rows = []

    def record(name, shape, N, t_ref, t_orc):
        rows.append(dict(case=name, shape=shape, N=N, reference_s=t_ref, oracle_s=t_orc,
                         reference_samples_per_s=N / t_ref, oracle_samples_per_s=N / t_orc,
                         oracle_over_reference=t_ref / t_orc))
        print("%-34s N=%-7d reference %.3f s  oracle %.3f s  oracle is %.2fx the reference's speed"
              % (name, N, t_ref, t_orc, t_ref / t_orc), flush=True)

    # 1. Gauss mixture multi_evaluate  (mixture.pyx:112-156)
    K, D, N = 16, 20, int(100000 * args.scale)
    mu, cov, w = mk(K, D, 1)
    x = draw(mu, cov, w, N, 7)
    mix = pypmc.density.mixture.create_gaussian_mixture(mu, cov, w)
    inv = np.array([c.inv_sigma for c in mix.components])
    ln = gauss_ln(mix)
    record("gauss mixture multi_evaluate", "K=16 D=20", N, best(lambda: mix.multi_evaluate(x)),
           best(lambda: orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)))

    # 2. Student-t mixture multi_evaluate  (student_t.pyx:135-166)
    K, D, N = 32, 30, int(30000 * args.scale)
    mu, cov, w = mk(K, D, 2)
    x = draw(mu, cov, w, N, 8)
    dof = np.full(K, 8.)
    tmix = pypmc.density.mixture.create_t_mixture(mu, cov, dof, w)
    inv = np.array([c.inv_sigma for c in tmix.components])
    ln = gammaln(.5 * (dof + D)) - gammaln(.5 * dof) - 0.5 * D * np.log(dof * np.pi) \
        - 0.5 * np.array([c.log_det_sigma for c in tmix.components])
    record("student-t mixture multi_evaluate", "K=32 D=30 nu=8", N, best(lambda: tmix.multi_evaluate(x)),
           best(lambda: orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, -.5 * (dof + D), 1. / dof)))

    # 3. VB E-step  (variational.pyx:116-127)
    K, D, N = 32, 20, int(50000 * args.scale)
    mu, cov, w = mk(K, D, 3)
    x = draw(mu, cov, w, N, 9)
    guess = pypmc.density.mixture.create_gaussian_mixture(mu, cov, w)
    vb = pypmc.mix_adapt.variational.GaussianInference(x, initial_guess=guess)
    ln_lambda = np.array(vb.expectation_det_ln_lambda)
    ln_pi = np.array(vb.expectation_ln_pi)
    m, W, beta, nu = np.array(vb.m), np.array(vb.W), np.array(vb.beta), np.array(vb.nu)
    record("VB E_step", "K=32 D=20", N, best(vb.E_step),
           best(lambda: orc.vb_estep(x, None, m, W, beta, nu, ln_pi, ln_lambda)))

    # 4. bench.py's step: IS weights vs a K_t=4 target + VB E-step, reference classes vs oracle
    tmu, tcov, tw = mk(4, D, 11)
    target = pypmc.density.mixture.create_gaussian_mixture(tmu, tcov, tw)
    prop = pypmc.density.mixture.create_gaussian_mixture(mu, cov, w)
    pinv = np.array([c.inv_sigma for c in prop.components])
    pln = gauss_ln(prop)
    tinv = np.array([c.inv_sigma for c in target.components])
    tln = gauss_ln(target)

    def ref_step():
        lt = target.multi_evaluate(x)
        lq = prop.multi_evaluate(x)
        wts = np.exp(lt - lq)
        pypmc.tools.convergence.perp(wts), pypmc.tools.convergence.ess(wts)
        vb.E_step()

    def orc_step():
        lt, _ = orc.mixture_multi_evaluate(0, x, tw, tmu, tinv, tln)
        lq, _ = orc.mixture_multi_evaluate(0, x, w, mu, pinv, pln)
        wts = orc.is_weights(lt, lq)
        orc.perp(wts), orc.ess(wts)
        orc.vb_estep(x, None, m, W, beta, nu, ln_pi, ln_lambda)
    record("bench step (IS pass + VB E-step)", "K=32+4 D=20", N, best(ref_step), best(orc_step))

    # 5. gaussian_pmc Rao-Blackwell, weighted  (pmc.pyx:120-246): reference end to end vs the oracle's
    #    N-sized part (rho + reductions; the K-sized update is microseconds)
    N = int(20000 * args.scale)
    xs, iw = x[:N], np.random.RandomState(1).uniform(0.5, 1.5, N)
    live = list(range(K))

    def orc_pmc():
        rho = orc.rho_rb(0, xs, w, mu, pinv, pln, None, None, live)
        orc.pmc_reductions(xs, rho, None, iw, live)
    record("gaussian_pmc rb weighted", "K=32 D=20", N,
           best(lambda: pypmc.mix_adapt.pmc.gaussian_pmc(xs, prop, iw), repeat=2), best(orc_pmc, repeat=2))

    import datetime
    out = dict(host=os.uname().nodename, cpu=open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
               measured=datetime.date.today().isoformat(), threads=1, note="reference = pypmc 1.2.6 built from /root/reference (Cython, gcc -O2 default flags); "
               "oracle = oracle/pmc_oracle.c (gcc -O3 -ffp-contract=off); best of 3, single thread",
               rows=rows)
    path = os.path.join(ROOT, "profiles", args.out)
    json.dump(out, open(path, "w"), indent=1)
    print(path)


if __name__ == "__main__":
    main()
