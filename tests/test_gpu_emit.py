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
    for key in ("weights", "out", "scalars"):                         # the weighting pass itself is unchanged
        np.testing.assert_array_equal(be.tohost(em[key]), be.tohost(plain[key]))
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
    for key in ("weights", "out", "scalars"):
        np.testing.assert_array_equal(be.tohost(em[key]), be.tohost(plain[key]))
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
    assert be.importance_weights(x, dead, target, emit=True).get("responsibilities") is None    # a dead component


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
