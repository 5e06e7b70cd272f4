"""pmc_importance_weights_emit + pmc_estep_from_u: a weighting pass that knows the Rao-Blackwell update follows
leaves u_nk = w_n rho_nk (pmc.pyx:23-43, :188) behind and the update is the statistics kernel alone -- against the
oracle, against the path that keeps the Mahalanobis forms (pmc_estep_from_tiles) and through the front-end."""
import numpy as np
import pytest

from test_gpu_kernels import mk, draw, gauss_set, student_set, assert_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    return HipBackend()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def untile(be, t, N, K):
    t = be.tohost(t)[:((N + 63) // 64) * K * 64].reshape(-1, K, 64)
    return np.concatenate([t[i].T for i in range(t.shape[0])])[:N]


@pytest.mark.parametrize("D,K,N", [(1, 2, 70), (2, 3, 1000), (5, 9, 257), (8, 17, 4097), (20, 32, 40000), (23, 5, 129),
                                   (32, 6, 640), (40, 24, 20000), (64, 3, 200), (9, 33, 20001)])
def test_emit_matches_oracle_and_the_kept_forms(be, orc, D, K, N):
    mu, cov, w = mk(K, D, 900 + D + K)
    x, _ = draw(mu, cov, w, N, 31)
    tmu, tcov, tw = mk(3, D, 78)
    tmu = 0.5 * tmu
    prop, inv, ln = gauss_set(mu, cov, w)
    target = gauss_set(tmu, tcov, tw)[0]
    plain = be.importance_weights(x, prop, target, want_out=True)
    em = be.importance_weights(x, prop, target, want_out=True, emit=True)
    resp = em["responsibilities"]
    assert resp is not None and resp.N == N and resp.K == K
    # the weighting pass itself is unchanged -- to the rounding of the merge: the pass that emits nothing walks the components of
    # a block in pieces at these sizes (round 6, k_logpdf_split), the emitting one does not
    for key in ("weights", "out", "scalars"):
        np.testing.assert_allclose(be.tohost(em[key]), be.tohost(plain[key]), rtol=1e-12, atol=1e-300)
    # u = w rho against the oracle's rho (pmc.pyx:23-43) and the weights just formed
    wts = be.tohost(em["weights"])
    rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
    u = resp.host_matrix(be)
    ref = wts[:, None] * rho
    normal = ref > 1e-290
    assert_rel(u[normal], ref[normal], rtol=1e-10, what="u = w rho")
    assert np.all(u[~normal] <= 1e-280)
    # the statistics: what the update that keeps the Mahalanobis forms computes, to rounding
    kept = be.importance_weights(x, prop, target, keep=True)
    a = be.tohost(be.estep_from_u(x, prop, resp)["stats"])[8:8 + K * (1 + D + D * (D + 1) // 2)]
    b = be.tohost(be.estep_from_tiles(x, prop, kept["tiles"], sample_w=kept["weights"])["stats"])[8:8 + len(a)]
    a, b = a.reshape(K, -1), b.reshape(K, -1)
    scale = np.abs(b).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(a - b) / scale).max() < 1e-11
    assert resp.matches(prop, em["weights"]) and not resp.matches(prop, kept["weights"])
    other, _, _ = gauss_set(mu + 1e-3, cov, w)
    assert not resp.matches(other, em["weights"])
    # advice r3: another sample array of the same length, and weights modified in place, are refused too
    xd = be.asdevice(x)
    em2 = be.importance_weights(xd, prop, target, emit=True)
    r2 = em2["responsibilities"]
    assert r2.matches(prop, em2["weights"], xd) and not r2.matches(prop, em2["weights"], xd.clone())
    em3 = be.importance_weights(xd, prop, target, emit=True)
    r3 = em3["responsibilities"]
    assert r3.matches(prop, em3["weights"], xd)
    em3["weights"].mul_(2.0)
    assert not r3.matches(prop, em3["weights"], xd)


@pytest.mark.parametrize("D,K,N", [(2, 3, 1000), (5, 9, 257), (20, 32, 30000), (30, 8, 5000), (40, 24, 20000), (64, 3, 200)])
def test_emit_student_t(be, orc, D, K, N):
    """Student-t proposals: u = w rho gamma and the two degree-of-freedom sums, against the kept-forms path"""
    mu, cov, w = mk(K, D, 950 + D + K)
    x, _ = draw(mu, cov, w, N, 33)
    dof = 3. + np.arange(K) % 5
    prop = student_set(mu, cov, w, dof)[0]
    target = gauss_set(*mk(3, D, 79))[0]
    plain = be.importance_weights(x, prop, target, want_out=True)
    em = be.importance_weights(x, prop, target, want_out=True, emit=True)
    resp = em["responsibilities"]
    assert resp is not None and resp.vsums is not None
    for key in ("weights", "out", "scalars"):                         # (to the rounding of the merge: see above)
        np.testing.assert_allclose(be.tohost(em[key]), be.tohost(plain[key]), rtol=1e-12, atol=1e-300)
    kept = be.importance_weights(x, prop, target, keep=True)
    a = be.tohost(be.estep_from_u(x, prop, resp)["stats"])
    b = be.tohost(be.estep_from_tiles(x, prop, kept["tiles"], sample_w=kept["weights"])["stats"])
    ps = 1 + D + D * (D + 1) // 2
    sa, sb = a[8:8 + K * ps].reshape(K, ps), b[8:8 + K * ps].reshape(K, ps)
    scale = np.abs(sb).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(sa - sb) / scale).max() < 1e-11
    np.testing.assert_allclose(a[8 + K * ps:], b[8 + K * ps:], rtol=1e-11, atol=1e-300)     # the dof sums
    assert np.all(a[8 + K * ps:] != 0.)


def test_emit_falls_back_where_it_does_not_apply(be):
    D, K, N = 6, 4, 500
    mu, cov, w = mk(K, D, 5)
    x, _ = draw(mu, cov, w, N, 6)
    target = gauss_set(*mk(2, D, 7))[0]
    wd = w.copy()
    wd[1] = 0.
    dead = gauss_set(mu, cov, wd)[0]
    # a pruned component: emitted since round 6 (columns for the live components only: test_emit_with_pruned_components)
    r = be.importance_weights(x, dead, target, emit=True).get("responsibilities")
    assert r is not None and r.K == K - 1 and r.live == [0, 2, 3]
    # no live component at all, a negative weight: the pass keeps nothing behind (callers ask can_emit first)
    assert not be.can_emit(gauss_set(mu, cov, np.zeros(K))[0])
    wn = w.copy()
    wn[2] = -0.1
    assert not be.can_emit(gauss_set(mu, cov, wn)[0])
    assert be.importance_weights(x, gauss_set(mu, cov, wn)[0], target, emit=True).get("responsibilities") is None


def test_front_end_iteration_without_a_responsibility_kernel(be):
    """ImportanceSampler.run_device(prepare_update=True) + gaussian_pmc(responsibilities=...) against the same
    iteration with the kept Mahalanobis forms"""
    import pypmc_amd as pypmc
    from pypmc_amd.density.mixture import create_gaussian_mixture
    D, K, N = 12, 20, 60000
    tmu, tcov, tw = mk(3, D, 11)
    target = create_gaussian_mixture(tmu / 3., tcov, tw)
    rs = np.random.RandomState(5)
    which = np.arange(K) % 3
    start = create_gaussian_mixture(tmu[which] / 3. + rs.normal(0, 0.2, (K, D)), 1.5 * tcov[which])
    results = []
    for form in ("tiles", "emit"):
        sampler = pypmc.sampler.importance_sampling.ImportanceSampler(target.evaluate, start,
                                                                      rng=np.random.RandomState(100))
        be.kernel_timings()
        be.kernel_timing(True)
        run = sampler.run_device(N, trace_sort=True, keep_mahalanobis=form == "tiles", prepare_update=form == "emit")
        if form == "emit":
            assert run["responsibilities"] is not None and run["mahalanobis"] is None
            new = pypmc.mix_adapt.pmc.gaussian_pmc(run["samples"], sampler.proposal, run["weights"], run["origin"],
                                                   responsibilities=run["responsibilities"])
        else:
            new = pypmc.mix_adapt.pmc.gaussian_pmc(run["samples"], sampler.proposal, run["weights"], run["origin"],
                                                   mahalanobis=run["mahalanobis"])
        be.kernel_timing(False)
        kernels = be.kernel_timings()
        assert ("k_resp" in kernels) == (form == "tiles")              # no responsibility kernel in the emitting form
        results.append(new)
        if form == "emit":
            with pytest.raises(ValueError):                            # other weights than the pass formed
                pypmc.mix_adapt.pmc.gaussian_pmc(run["samples"], sampler.proposal, run["weights"].clone(), run["origin"],
                                                 responsibilities=run["responsibilities"])
            # advice r4: an equal COPY of the samples (a history reallocated by a later append looks like this) is not an
            # error -- nothing proves the values stale -- the update just forms its responsibilities itself
            be.kernel_timings()
            be.kernel_timing(True)
            again = pypmc.mix_adapt.pmc.gaussian_pmc(run["samples"].clone(), sampler.proposal, run["weights"], run["origin"],
                                                     responsibilities=run["responsibilities"])
            be.kernel_timing(False)
            assert "k_resp" in be.kernel_timings()
            np.testing.assert_allclose(again.weights, new.weights, rtol=1e-11)
    a, b = results
    np.testing.assert_allclose(b.weights, a.weights, rtol=1e-11)
    for ca, cb in zip(a.components, b.components):
        np.testing.assert_allclose(cb.mu, ca.mu, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(cb.sigma, ca.sigma, rtol=1e-9, atol=1e-12)


def test_front_end_student_t_iteration(be):
    """student_t_pmc(responsibilities=...) against student_t_pmc(mahalanobis=...): weights, means, covariances and the
    adapted degrees of freedom"""
    import pypmc_amd as pypmc
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    D, K, N = 6, 8, 80000
    tmu, tcov, tw = mk(3, D, 12)
    target = create_gaussian_mixture(tmu / 3., tcov, tw)
    rs = np.random.RandomState(6)
    which = np.arange(K) % 3
    start = create_t_mixture(tmu[which] / 3. + rs.normal(0, 0.2, (K, D)), 1.5 * tcov[which], np.full(K, 6.))
    results = []
    for form in ("tiles", "emit"):
        sampler = pypmc.sampler.importance_sampling.ImportanceSampler(target.evaluate, start,
                                                                      rng=np.random.RandomState(101))
        run = sampler.run_device(N, trace_sort=True, keep_mahalanobis=form == "tiles", prepare_update=form == "emit")
        kw = dict(responsibilities=run["responsibilities"]) if form == "emit" else dict(mahalanobis=run["mahalanobis"])
        assert (run["responsibilities"] is not None) == (form == "emit")
        results.append(pypmc.mix_adapt.pmc.student_t_pmc(run["samples"], sampler.proposal, run["weights"], run["origin"], **kw))
    a, b = results
    np.testing.assert_allclose(b.weights, a.weights, rtol=1e-11)
    for ca, cb in zip(a.components, b.components):
        np.testing.assert_allclose(cb.mu, ca.mu, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(cb.sigma, ca.sigma, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(cb.dof, ca.dof, rtol=1e-8)


@pytest.mark.parametrize("D,K,N", [(2, 5, 3000), (5, 12, 5000), (7, 32, 4097), (12, 9, 2000), (20, 32, 30000), (24, 40, 17001)])
def test_estep_about_other_points(be, orc, D, K, N):
    """pmc_estep_about: the moments about points of the caller's choice (one-kernel forms, two kernels, common-shift
    statistics) -- centred results equal to the E-step about the components' own means, and the oracle's"""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    from test_gpu_stats_gemm import vb_set
    be.configure("stats_common_shift_min_n", 0)
    be.configure("stats_common_shift_min_fill", 0)
    try:
        mu, cov, w = mk(K, D, 70 + D)
        x, _ = draw(mu, cov, w, N, 71)
        cs, (m, W, beta, nu, ln_pi, ln_lambda) = vb_set(mu, cov, D, K, 72)
        rs = np.random.RandomState(73)
        shift = m + 0.3 * rs.normal(size=m.shape)
        ref = orc.vb_estep(x, None, m, W, beta, nu, ln_pi, ln_lambda)
        own = be.tohost(be.estep(x, cs, 0)["stats"]).copy()
        other = be.tohost(be.estep(x, cs, 0, shift=shift)["stats"]).copy()
        np.testing.assert_allclose(other[:8], own[:8], rtol=1e-12, atol=1e-300)          # scalars do not depend on it
        a, b = split_stats(own, K, D), split_stats(other, K, D)
        np.testing.assert_allclose(b[1], a[1], rtol=1e-12)
        xa, Sa = centred_moments(a[1], a[2], a[3], m)
        xb, Sb = centred_moments(b[1], b[2], b[3], shift)
        live = ref["N_comp"] > 1e-6
        np.testing.assert_allclose(xb[live], xa[live], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(Sb[live], Sa[live], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(xb[live], ref["x_mean_comp"][live], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(Sb[live], ref["S"][live], rtol=1e-8, atol=1e-10)
        # about the mean itself the first moments vanish and the second are the covariance
        again = split_stats(be.tohost(be.estep(x, cs, 0, shift=xb)["stats"]), K, D)
        assert np.abs(again[2][live] / again[1][live][:, None]).max() < 1e-10
    finally:
        be.configure("stats_common_shift_min_n", 524288)
        be.configure("stats_common_shift_min_fill", 0.63)


@pytest.mark.parametrize("D,K,N,student,mgemm", [(2, 5, 1000, False, False), (8, 17, 4097, False, False), (20, 32, 20000, False, False),
                                                 (20, 12, 3000, True, False), (40, 64, 3000, False, True),
                                                 (40, 128, 2600, False, True), (64, 64, 1500, False, True),
                                                 (40, 20, 700, True, False)])
def test_emit_with_pruned_components(be, orc, D, K, N, student, mgemm):
    """pmc_importance_weights_emit_live (round 6): a proposal that holds pruned components (weight 0, left in the mixture:
    pmc.pyx:109-117).  The pass still evaluates them (log q's row maximum, _regularize.pyx:73-77); u = w rho [gamma] gets
    columns for the LIVE components only (calculate_rho_rb over live_components, pmc.pyx:23-43) -- against the oracle,
    through the exact kernel and (D >= 32, large K) the matrix-product form, and the update's statistics against the
    path that forms its responsibilities itself"""
    mu, cov, w = mk(K, D, 950 + D + K)
    rs = np.random.RandomState(K)
    dead = np.sort(rs.choice(K, max(K // 5, 1), replace=False))
    wl = w.copy()
    wl[dead] = 0.
    wl /= wl.sum()
    live = [k for k in range(K) if wl[k] != 0]
    # samples of the live mixture.  (On a pruned component far from every live one all live values lie below -700 and the
    # reference's own exp(a_k) / (exp(lse) + tiny) is a quotient of denormals, a few bits wide: nothing to compare with.)
    x, _ = draw(mu, 1.2 * cov, wl, N, 33)
    tmu, tcov, tw = mk(3, D, 78)
    target = gauss_set(0.5 * tmu, tcov, tw)[0]
    if student:
        dof = 3.5 + 0.5 * (np.arange(K) % 5)
        prop, inv, ln, pf, idf = student_set(mu, cov, wl, dof)
        logq, _ = orc.mixture_multi_evaluate(1, x, wl, mu, inv, ln, pf, idf)
        rho = orc.rho_rb(1, x, wl, mu, inv, ln, pf, idf, live)[:, live]
    else:
        prop, inv, ln = gauss_set(mu, cov, wl)
        logq, _ = orc.mixture_multi_evaluate(0, x, wl, mu, inv, ln)
        rho = orc.rho_rb(0, x, wl, mu, inv, ln, None, None, live)[:, live]
    assert be.can_emit(prop)
    if mgemm:
        be.configure("maha_gemm_min_n", 1000)
    try:
        em = be.importance_weights(x, prop, target, want_out=True, emit=True)
        if mgemm:
            rep = be.maha_gemm_report(N, K, D)
            assert rep is not None and rep["refused"] == 0, rep
    finally:
        be.reset_option("maha_gemm_min_n")
    resp = em["responsibilities"]
    assert resp is not None and resp.K == len(live) and resp.live == live
    assert_rel(be.tohost(em["out"]), logq, what="log q of the mixture with pruned components")
    wts = be.tohost(em["weights"])
    u = resp.host_matrix(be)
    assert u.shape == (N, len(live))
    ref = wts[:, None] * rho
    if student:
        maha = np.einsum('nki,kij,nkj->nk', x[:, None, :] - mu[None, live], inv[live], x[:, None, :] - mu[None, live])
        ref = ref * (dof[live] + D) / (dof[live] + maha)                      # gamma, pmc.pyx:610
    normal = ref > 1e-290
    assert_rel(u[normal], ref[normal], rtol=1e-10, what="u = w rho [gamma], live columns")
    # the statistics of the update, against the path that evaluates the live components again (max_init_zero rule)
    from pypmc_amd.backend import ComponentSet
    sel = np.array(live)
    cs = ComponentSet(prop.kind, prop.mu[sel], prop.precision[sel], prop.c0[sel], prop.c1[sel], prop.c2[sel], prop.c3[sel],
                      weight=prop.weight[sel], column=sel, ld=K)
    a = be.tohost(be.estep_from_u(x, cs, resp)["stats"])
    b = be.tohost(be.estep(x, cs, 1, max_init_zero=True, sample_w=em["weights"])["stats"])
    Kl, ps = len(live), 1 + D + D * (D + 1) // 2
    sa, sb = a[8:8 + Kl * ps].reshape(Kl, ps), b[8:8 + Kl * ps].reshape(Kl, ps)
    scale = np.abs(sb).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(sa - sb) / scale).max() < 1e-10, "statistics from the emitted responsibilities"
    if student:
        np.testing.assert_allclose(a[8 + Kl * ps:], b[8 + Kl * ps:], rtol=1e-10, atol=1e-300)


def test_front_end_iteration_after_a_prune(be):
    """ImportanceSampler.run_device(prepare_update=True) + gaussian_pmc on a proposal with pruned components: the pass
    emits (it used to fall back to keeping the Mahalanobis forms), and the update equals the one that forms its
    responsibilities itself"""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import ImportanceSampler
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    D, K, N = 12, 20, 30000
    tmu, tcov, tw = mk(4, D, 11)
    tmu /= 3.0
    target = create_gaussian_mixture(tmu, tcov, tw)
    which = np.arange(K) % 4
    w = np.ones(K)
    w[[3, 7, 8, 19]] = 0.
    proposal = create_gaussian_mixture(tmu[which] + np.random.RandomState(5).normal(0, 0.15, (K, D)), 1.5 * tcov[which], w / w.sum())
    np.random.seed(100)
    sampler = ImportanceSampler(target.evaluate, proposal)
    r = sampler.run_device(N, trace_sort=True, prepare_update=True)
    assert r["responsibilities"] is not None and r["mahalanobis"] is None
    assert r["responsibilities"].K == K - 4
    new = gaussian_pmc(r["samples"], sampler.proposal, r["weights"], r["origin"], mincount=0, rb=True, copy=True,
                       responsibilities=r["responsibilities"])
    ref = gaussian_pmc(r["samples"], sampler.proposal, r["weights"], r["origin"], mincount=0, rb=True, copy=True)
    np.testing.assert_allclose(new.weights, ref.weights, rtol=1e-10, atol=1e-300)
    assert (np.array(new.weights)[[3, 7, 8, 19]] == 0).all()
    for a, b in zip(new.components, ref.components):
        np.testing.assert_allclose(a.mu, b.mu, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(a.sigma, b.sigma, rtol=1e-9, atol=1e-11)
    # responsibilities of another live set are refused
    sampler.proposal.weights[0] = 0.
    sampler.proposal.normalize()
    with pytest.raises(ValueError):
        gaussian_pmc(r["samples"], sampler.proposal, r["weights"], r["origin"], mincount=0, rb=True, copy=True,
                     responsibilities=r["responsibilities"])


def test_student_t_pmc_solves_its_degrees_of_freedom_at_once(be):
    """student_t_pmc with 16 or more live components takes all roots of the degree-of-freedom condition at once
    (mix_adapt.pmc._solve_dofs) instead of one brentq per component (pmc.pyx:693-710): the same update, the dofs to brentq's
    tolerance"""
    import pypmc_amd.mix_adapt.pmc as pmc_mod
    from pypmc_amd.density.mixture import create_t_mixture
    from pypmc_amd.mix_adapt.pmc import student_t_pmc
    rs = np.random.RandomState(4)
    K, D, N = 24, 6, 40000
    mu = rs.normal(size=(K, D)) * 4
    cov = np.array([np.eye(D) * rs.uniform(0.5, 2.0) for _ in range(K)])
    dofs = rs.uniform(2., 30., size=K)
    prop = create_t_mixture(mu, cov, dofs, rs.uniform(0.5, 1.5, size=K))
    np.random.seed(5)
    x = prop.propose(N)
    w = rs.uniform(0.2, 2.0, size=N)
    batched = student_t_pmc(x, prop, weights=w, copy=True, backend=be)
    saved, pmc_mod.DOF_BATCH_FROM = pmc_mod.DOF_BATCH_FROM, 1 << 30
    try:
        loop = student_t_pmc(x, prop, weights=w, copy=True, backend=be)
    finally:
        pmc_mod.DOF_BATCH_FROM = saved
    np.testing.assert_array_equal(batched.weights, loop.weights)
    for a, b in zip(batched.components, loop.components):
        np.testing.assert_array_equal(a.mu, b.mu)
        np.testing.assert_array_equal(a.sigma, b.sigma)
        assert abs(a.dof - b.dof) <= 4e-12 + 1e-11 * b.dof, (a.dof, b.dof)
    assert len({round(c.dof, 6) for c in batched.components}) > K // 2      # (the dofs did move, each its own way)
