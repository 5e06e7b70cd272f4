#!/usr/bin/env python3
"""Randomised sweep of the HIP path against the oracle over EVERY sample dimension 1..64 (exact and
padded kernel units), ragged N and odd K.  tests/test_gpu_fuzz.py runs it with fixed seeds under
`-m gpu`; by hand on the GPU box:

    python tests/fuzz_gpu.py [seed] [rounds]
"""
import os
import sys

import numpy as np
from scipy.special import digamma

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def mk(K, D, rs):
    mu = rs.normal(0, 3, size=(K, D))
    cov = np.empty((K, D, D))
    for k in range(K):
        A = rs.normal(0, 1, size=(D, D))
        cov[k] = A.dot(A.T) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K)
    return mu, cov, w / w.sum()


def rel(a, b, floor=1e-300):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def stats_err(one, two, K, D, split_stats):
    """Two forms of the same statistics vector against each other, every block on ITS scale: counts and scalars
    relative, first moments against sqrt(count * largest second moment), second moments against the component's
    largest (an off-diagonal entry that nearly cancels carries the rounding of its terms).  A value that is not finite in
    one form only -- or a NaN that is not 0 / 0 of two all-zero vectors -- counts as an infinite error."""
    a, b = split_stats(one, K, D), split_stats(two, K, D)
    tiny = 1e-300

    def worst(diff, scale):
        with np.errstate(invalid="ignore", divide="ignore"):
            q = np.abs(diff) / scale
        q = np.where((diff == 0) & ~np.isfinite(q), 0.0, q)          # 0 / 0: both forms hold an exact zero on a zero scale
        return float("inf") if np.isnan(q).any() else float(np.max(q))

    floor0 = np.full(len(b[0]), 1e-9 * np.abs(b[0]).max())
    # scalar 0 (VB: E[log q(Z)] = sum_n w_n sum_k r log r): every sample's term is formed to a few ulps of 1, so the sum carries an
    # ABSOLUTE error of (sum of the sample weights) x eps however small it is -- nearly one-hot responsibilities make it tiny
    # (the oracle check in _sweep uses the same floor)
    floor0[0] = max(floor0[0], 1e-4 * np.abs(b[1]).sum())
    e = [worst(a[0] - b[0], np.abs(b[0]) + floor0 + tiny)]
    cnt = np.abs(b[1]) + 1e-12 * np.abs(b[1]).max() + tiny               # (components nobody belongs to: on the total's scale)
    e.append(worst(a[1] - b[1], cnt))
    m2 = np.abs(b[3]).reshape(K, -1).max(axis=1)
    m2 = m2 + 1e-12 * m2.max() + tiny
    e.append(worst(a[2] - b[2], np.sqrt(cnt * m2)[:, None]))
    e.append(worst(a[3] - b[3], m2[:, None, None]))
    if np.any(b[4]):
        e.append(worst(a[4] - b[4], np.abs(b[4]) + 1e-12 * np.abs(b[4]).max() + tiny))
    return max(e)


def sweep(seed=0, rounds=1, be=None, dims=range(1, 65), verbose=True, kmax=40, nmax=3000, fast_paths=False):
    """Returns {quantity: worst relative error}; raises AssertionError naming the failing shape.
    ``fast_paths``: the large-N forms of pmc_estep -- common-shift statistics (k_stats_gemm), responsibilities in
    groups (k_resp_groups) -- are switched on at any N / K fill (pmc_configure), and the emitting weighting pass
    (pmc_importance_weights_emit -> pmc_estep_from_u) is checked against the oracle as well."""
    from oracle import oracle as orc
    from pypmc_amd.backend import HipBackend, ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    be = be or HipBackend()
    rs = np.random.RandomState(seed)
    worst = {}
    if fast_paths:
        be.configure("stats_common_shift_min_n", 0)
        be.configure("stats_common_shift_min_fill", 0)
        be.configure("stats_common_shift_min_k", 2)
        be.configure("estep_grouped_responsibilities", 2)
        be.configure("maha_gemm_min_n", 256)                 # the matrix-product form of the Mahalanobis forms from one workgroup on
    else:
        be.configure("maha_gemm_min_n", 2 ** 40)             # this sweep holds the exact kernels to bitwise identities
    try:
        return _sweep(seed, rounds, be, dims, verbose, kmax, nmax, fast_paths, rs, worst, orc, ComponentSet, split_stats,
                      centred_moments)
    finally:
        if fast_paths:
            be.configure("stats_common_shift_min_n", 524288)
            be.configure("stats_common_shift_min_fill", 0.63)
            be.configure("stats_common_shift_min_k", 17)
            be.configure("estep_grouped_responsibilities", 1)
        be.reset_option("maha_gemm_min_n")


def _sweep(seed, rounds, be, dims, verbose, kmax, nmax, fast_paths, rs, worst, orc, ComponentSet, split_stats, centred_moments):
    from pypmc_amd.mix_adapt._stats import shift_is_far

    def note(name, v, tol, ctx):
        worst[name] = max(worst.get(name, 0.0), v)
        assert v < tol, "%s: %.3g >= %.3g at %s" % (name, v, tol, ctx)

    floor2 = 1e-9

    def two_forms(one, two, K, D):
        # (K > 16: a small batch's E-step forms its responsibilities in groups of 16, the groups in pieces, since round 6 -- u
        #  to rounding, not bitwise: sums that cancel are compared on the scale of their component, not element by element)
        if fast_paths or K > 16:
            return stats_err(one, two, K, D, split_stats)
        return float(np.max(np.abs(one - two) / (np.abs(two) + floor2 * np.abs(two).max() + 1e-300)))
    for rnd in range(rounds):
        for D in dims:
            K = int(rs.randint(1, kmax + 1))
            N = int(rs.choice([1, 2, 63, 64, 65, 127, 129, rs.randint(1, nmax)]))
            if fast_paths:
                N = int(rs.choice([16384, 16385, 16447, 20000 + rs.randint(0, 300)]))   # the forms start at 16384 samples
            ctx = dict(D=D, K=K, N=N, seed=seed, round=rnd)
            mu, cov, w = mk(K, D, rs)
            k = rs.choice(K, size=N, p=w)
            L = np.linalg.cholesky(cov)
            x = mu[k] + np.einsum('nij,nj->ni', L[k], rs.normal(size=(N, D)))
            inv = np.linalg.inv(cov)
            inv = 0.5 * (inv + inv.transpose(0, 2, 1))
            ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
            # log-pdf + importance weights against a second mixture
            cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
            ref_q, ref_ind = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
            res = be.logpdf(x, cs, want_individual=True)
            note("logpdf", rel(be.tohost(res["out"]), ref_q), 1e-10, ctx)
            note("individual", rel(be.tohost(res["individual"]), ref_ind), 1e-10, ctx)
            KT = int(rs.randint(1, 5))
            tmu, tcov, tw = mk(KT, D, rs)
            tinv = np.linalg.inv(tcov)
            tinv = 0.5 * (tinv + tinv.transpose(0, 2, 1))
            tln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(tcov)[1]
            ref_t = orc.mixture_multi_evaluate(0, x, tw, tmu, tinv, tln)[0]
            iw = be.importance_weights(x, cs, ComponentSet(0, tmu, tinv, c0=tln, weight=tw))
            with np.errstate(over='ignore'):
                ref_w = np.exp(ref_t - ref_q)
            fin = np.isfinite(ref_w)
            note("weights", rel(be.tohost(iw["weights"])[fin], ref_w[fin]), 1e-10, ctx)
            # Student-t mixture log-pdf
            from scipy.special import gammaln
            dof = rs.uniform(1.0, 12.0, K)
            tln_ = gammaln(.5 * (dof + D)) - gammaln(.5 * dof) - 0.5 * D * np.log(dof * np.pi) - \
                0.5 * np.linalg.slogdet(cov)[1]
            scs = ComponentSet(1, mu, inv, c0=tln_, c1=-.5 * (dof + D), c2=1. / dof, c3=dof, weight=w)
            ref_s = orc.mixture_multi_evaluate(1, x, w, mu, inv, tln_, -.5 * (dof + D), 1. / dof)[0]
            note("student logpdf", rel(be.tohost(be.logpdf(x, scs)["out"]), ref_s), 1e-10, ctx)
            # VB E-step
            sw = rs.uniform(0.5, 1.5, N) if rs.rand() < 0.5 else None
            nu = D + 2. + rs.uniform(0, 5, K)
            beta = 1. + rs.uniform(0, 5, K)
            alpha = 1. + rs.uniform(0, 5, K)
            W = inv / nu[:, None, None]
            m = mu + 0.1 * rs.normal(size=mu.shape)
            ln_lambda = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + \
                np.linalg.slogdet(W)[1]
            ln_pi = digamma(alpha) - digamma(alpha.sum())
            ref = orc.vb_estep(x, sw, m, W, beta, nu, ln_pi, ln_lambda)
            vcs = ComponentSet(2, m, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
            out = be.estep(x, vcs, 0, sample_w=sw, want_r=True)
            note("vb r", rel(be.tohost(out["r"]), ref["r"]), 1e-10, ctx)
            sc, S0, M1, M2, _, _ = split_stats(be.tohost(out["stats"]), K, D)
            note("vb N_comp", rel(S0, ref["N_comp"], 1e-30), 1e-10, ctx)
            live = ref["N_comp"] > 1e-3
            xm, S = centred_moments(S0, M1, M2, m)
            if shift_is_far(S0, M1, M2):
                # what GaussianInference.E_step does then: the moments once more, about the means just found
                _, S0b, M1b, M2b, _, _ = split_stats(be.tohost(be.estep(x, vcs, 0, sample_w=sw, shift=xm)["stats"]), K, D)
                xm, S = centred_moments(S0b, M1b, M2b, xm)
            if live.any():
                # the reference's own normalisation (variational.pyx:806-932: x-bar and S are divided by N_k), on the
                # components' own scale: a mean against its standard deviation, S_ij against sqrt(S_ii S_jj)
                # (a handful of samples: no spread to speak of -- then the coordinate's own size)
                sd = np.maximum(np.sqrt(np.einsum('kii->ki', ref["S"]))[live], 1e-3 * (1. + np.abs(ref["x_mean_comp"][live])))
                note("vb x_mean", float(np.max(np.abs(xm[live] - ref["x_mean_comp"][live]) / sd)), 1e-10, ctx)
                note("vb S", float(np.max(np.abs(S[live] - ref["S"][live]) / (sd[:, :, None] * sd[:, None, :]))), 1e-10, ctx)
            # E[log q(Z)] = sum_n w_n sum_k r log r (variational.pyx:1003-1013): every sample's term lies in [-log K, 0] and
            # is formed to a few ulps of 1 on both sides, so the sum carries an ABSOLUTE error of N eps however small it is
            # (responsibilities that are nearly one-hot make it tiny); 1e-10 relative with that floor of 50 eps per sample
            elq_ref = ref["expectation_log_q_Z"]
            note("vb elq", abs(sc[0] - elq_ref) / (abs(elq_ref) + 1e-4 * N), 1e-10, ctx)
            # the same E-step through pmc_estep (small D: ONE kernel, register or LDS form) against the two kernels
            one = be.tohost(be.estep(x, vcs, 0, sample_w=sw)["stats"])
            two = be.tohost(out["stats"])
            note("vb one-kernel stats", two_forms(one, two, K, D), 1e-9, ctx)
            # PMC Rao-Blackwell responsibilities + statistics
            iwts = rs.uniform(0.1, 2.0, N)
            out = be.estep(x, cs, 1, sample_w=iwts, want_r=True)
            rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
            got = be.tohost(out["r"])
            normal = ref_ind > -690                          # exp(log q_k) a normal number in the reference
            note("pmc rho", rel(got[normal], rho[normal]), 1e-10, ctx)
            # below, the reference's numerator is denormal (a few bits) or zero: follow it loosely
            note("pmc rho (denormal numerator)", rel(got[~normal], rho[~normal], 1e-200), 5e-2, ctx)
            sc, S0, M1, M2, _, _ = split_stats(be.tohost(out["stats"]), K, D)
            # alpha, mu, Sigma of the update in the reference's own normalisation (pmc.pyx:188-222), from the oracle's loops
            o_alpha, o_mu, o_cov = orc.pmc_reductions(x, rho, None, iwts, list(range(K)))
            note("pmc alpha", rel(S0, o_alpha, 1e-30), 1e-10, ctx)
            held = o_alpha > 1e-6 * o_alpha.sum()            # (a component nobody belongs to has no scale of its own)
            if held.any():
                pm, pc = centred_moments(S0, M1, M2, mu)
                if shift_is_far(S0, M1, M2):                # (gaussian_pmc's second pass)
                    _, S0b, M1b, M2b, _, _ = split_stats(be.tohost(be.estep(x, cs, 1, sample_w=iwts, shift=pm)["stats"]), K, D)
                    pm, pc = centred_moments(S0b, M1b, M2b, pm)
                sd = np.maximum(np.sqrt(np.einsum('kii->ki', o_cov))[held], 1e-3 * (1. + np.abs(o_mu[held])))
                note("pmc mu", float(np.max(np.abs(pm[held] - o_mu[held]) / sd)), 1e-10, ctx)
                note("pmc Sigma", float(np.max(np.abs(pc[held] - o_cov[held]) / (sd[:, :, None] * sd[:, None, :]))), 1e-10, ctx)
            # (beyond D ~ 500 every exp(log q_k) underflows in the reference too: rho = 0, M2 = 0)
            one = be.tohost(be.estep(x, cs, 1, sample_w=iwts)["stats"])
            two = be.tohost(out["stats"])
            note("pmc one-kernel stats", two_forms(one, two, K, D), 1e-9, ctx)
            # the evaluate-once iteration: Mahalanobis forms kept by the weighting pass -> the same statistics, bitwise
            kept = be.importance_weights(x, cs, ComponentSet(0, tmu, tinv, c0=tln, weight=tw), keep=True)
            if not fast_paths:
                assert np.array_equal(be.tohost(kept["weights"]), be.tohost(iw["weights"])), ("kept weights", ctx)
            else:                                            # (D >= 32: the pass that keeps nothing may be the matrix product)
                note("kept weights", rel(be.tohost(kept["weights"])[fin], be.tohost(iw["weights"])[fin]), 1e-10, ctx)
            pre = be.tohost(be.estep_from_tiles(x, cs, kept["tiles"], sample_w=iwts)["stats"])
            if not fast_paths:
                assert np.array_equal(pre, two), ("estep_from_tiles differs from the two kernels", ctx)
            else:                                            # (the statistics may take either form: rounding)
                note("kept-forms stats", two_forms(pre, two, K, D), 1e-9, ctx)
            if fast_paths:
                # the weighting pass that emits u = w rho itself: same weights; statistics of the reference's
                # importance-weighted Rao-Blackwell update (weights = the pass's own importance weights)
                tcs = ComponentSet(0, tmu, tinv, c0=tln, weight=tw)
                em = be.importance_weights(x, cs, tcs, emit=True)
                resp = em.get("responsibilities")
                if resp is not None:
                    from pypmc_amd.backend import NSCALARS as NSC
                    PS = 1 + D + D * (D + 1) // 2
                    # (D <= 24: only the pass that emits nothing may be the matrix product; everywhere: only that pass walks
                    #  the components of a block in pieces -- the rounding of the merge)
                    note("emit weights", rel(be.tohost(em["weights"])[fin], be.tohost(iw["weights"])[fin]),
                         1e-12 if D > 24 else 1e-10, ctx)
                    wts = be.tohost(iw["weights"])
                    viaw = be.tohost(be.estep(x, cs, 1, sample_w=wts, want_r=True)["stats"])
                    got = be.tohost(be.estep_from_u(x, cs, resp)["stats"])
                    got[:NSC] = viaw[:NSC]                 # (the emitting pass leaves its scalars with the weights)
                    note("emit stats", two_forms(got, viaw, K, D), 1e-9, ctx)
        if verbose:
            print("round %d ok" % rnd, flush=True)
    return worst


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for k_, v in sorted(sweep(seed, rounds).items()):
        print("worst %-12s %.3g" % (k_, v))


if __name__ == "__main__":
    main()
