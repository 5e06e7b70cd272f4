"""The component x monomial form of the sufficient statistics (k_stats_gemm: moments about one common shift as a
matrix product on v_mfma_f64_16x16x4, re-centred on the device) against the per-component-shift kernel it replaces
in pmc_estep, and against the oracle: same statistics to rounding where the components are near the common shift,
the old kernel's numbers bit for bit where the a-posteriori test sends the call back to it.
Reference loops: variational.pyx:699-932, pmc.pyx:188-222."""
import numpy as np
import pytest
from scipy.special import digamma

from test_gpu_kernels import mk, draw, gauss_set, assert_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    b = HipBackend()
    b.configure("stats_common_shift_min_n", 0)          # the form from 16384 samples on (default: where it pays)
    b.configure("stats_common_shift_min_fill", 0)       # ... and for any K >= 17 (default: groups of 32 well filled)
    b.configure("estep_grouped_responsibilities", 2)    # ... with k_resp_groups in front of it at every dimension
    yield b
    b.configure("estep_grouped_responsibilities", 1)
    b.configure("stats_common_shift_min_fill", 0.63)
    b.configure("stats_common_shift_limit", 1000.0)
    b.configure("stats_common_shift_min_k", 17)
    b.configure("stats_common_shift_min_n", 524288)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def vb_set(mu, cov, D, K, seed):
    from pypmc_amd.backend import ComponentSet
    rs = np.random.RandomState(seed)
    nu = D + 2. + rs.uniform(0, 5, K)
    beta = 1. + rs.uniform(0, 5, K)
    alpha = 1. + rs.uniform(0, 5, K)
    W = np.linalg.inv(cov) / nu[:, None, None]
    W = 0.5 * (W + W.transpose(0, 2, 1))
    m = mu + 0.1 * rs.normal(size=mu.shape)
    ln_lambda = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + np.linalg.slogdet(W)[1]
    ln_pi = digamma(alpha) - digamma(alpha.sum())
    cs = ComponentSet(2, m, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    return cs, (m, W, beta, nu, ln_pi, ln_lambda)


def both_forms(be, x, cs, mode, **kw):
    """statistics of pmc_estep with the common-shift form allowed and with it switched off"""
    be.configure("stats_common_shift_limit", 1000.0)
    fast = be.tohost(be.estep(x, cs, mode, **kw)["stats"]).copy()
    be.configure("stats_common_shift_limit", 0.0)
    try:
        slow = be.tohost(be.estep(x, cs, mode, **kw)["stats"]).copy()
    finally:
        be.configure("stats_common_shift_limit", 1000.0)
    return fast, slow


def scaled_close(fast, slow, K, D, tol):
    """the sums of each component against the magnitude of that component's sums of the same order"""
    from pypmc_amd.mix_adapt._stats import split_stats
    a, b = split_stats(fast, K, D), split_stats(slow, K, D)
    # scalars: E[log q(Z)] / sum w log q from k_resp_groups (combined over groups of 16) against k_resp's two passes
    np.testing.assert_allclose(a[0], b[0], rtol=1e-10, atol=1e-11)       # (the contract's tolerance on E[log q(Z)])
    assert_rel(a[1], b[1], rtol=1e-12, what="sum u")
    for k in range(K):
        s1 = np.abs(b[2][k]).max() + np.sqrt(np.abs(np.diag(b[3][k])).max() * max(b[1][k], 0.0))
        assert np.abs(a[2][k] - b[2][k]).max() <= tol * max(s1, 1e-300), ("first moments", k)
        s2 = np.abs(np.diag(b[3][k])).max()
        assert np.abs(a[3][k] - b[3][k]).max() <= tol * max(s2, 1e-300), ("second moments", k)


@pytest.mark.parametrize("D,K,N", [(8, 17, 20001), (9, 20, 16384), (10, 32, 30000), (12, 33, 25000), (16, 40, 20011),
                                   (18, 32, 33333), (20, 32, 100003), (20, 64, 50000), (20, 100, 17000),
                                   (24, 17, 20000), (27, 24, 19999), (30, 32, 20000), (32, 48, 18000),
                                   (40, 128, 20000), (48, 20, 17001), (57, 18, 16500), (64, 33, 16999)])
def test_common_shift_form_matches_the_per_component_form(be, D, K, N):
    mu, cov, w = mk(K, D, 700 + D + K)
    x, _ = draw(mu, cov, w, N, 3)
    cs, _ = vb_set(mu, cov, D, K, D * K)
    rs = np.random.RandomState(N)
    sw = rs.uniform(0.5, 1.5, N)
    fast, slow = both_forms(be, x, cs, 0, sample_w=sw)
    assert not np.array_equal(fast, slow), "the common-shift form did not run"
    scaled_close(fast, slow, K, D, 1e-11)
    # bit-reproducible
    be.configure("stats_common_shift_limit", 1000.0)
    np.testing.assert_array_equal(be.tohost(be.estep(x, cs, 0, sample_w=sw)["stats"]), fast)


def test_common_shift_form_vs_oracle(be, orc):
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    D, K, N = 20, 32, 40000
    mu, cov, w = mk(K, D, 5)
    x, _ = draw(mu, cov, w, N, 6)
    cs, (m, W, beta, nu, ln_pi, ln_lambda) = vb_set(mu, cov, D, K, 7)
    ref = orc.vb_estep(x, None, m, W, beta, nu, ln_pi, ln_lambda)
    be.configure("stats_common_shift_limit", 1000.0)
    flat = be.tohost(be.estep(x, cs, 0)["stats"])
    sc, S0, M1, M2, _, _ = split_stats(flat, K, D)
    x_mean, S = centred_moments(S0, M1, M2, m)
    assert_rel(S0, ref["N_comp"], rtol=1e-11, what="N_comp")
    live = ref["N_comp"] > 1e-6
    np.testing.assert_allclose(x_mean[live], ref["x_mean_comp"][live], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(S[live], ref["S"][live], rtol=1e-9, atol=1e-11)


def test_far_components_fall_back_to_the_per_component_kernel(be):
    """modes 300 standard deviations apart: the a-priori test (Gaussian kinds) or the a-posteriori test (VB, whose
    pack holds W = precision / nu) refuses the common shift, and the result is the old kernel's, bit for bit"""
    D, K, N = 20, 32, 30000
    mu, cov, w = mk(K, D, 11)
    mu = mu * 100.0
    x, _ = draw(mu, cov, w, N, 12)
    cs, _ = vb_set(mu, cov, D, K, 13)
    gs, _, _ = gauss_set(mu, cov, w)
    be.configure("estep_grouped_responsibilities", 0)       # k_resp as ever: the fallback is the old path bit for bit
    try:
        fast, slow = both_forms(be, x, cs, 0)
        np.testing.assert_array_equal(fast, slow)
        fast, slow = both_forms(be, x, gs, 1)
        np.testing.assert_array_equal(fast, slow)
    finally:
        be.configure("estep_grouped_responsibilities", 2)
    # with the grouped responsibilities the refused form first completes u (u' x factor: one rounding more than
    # k_resp's e / s) and then runs the same per-component kernel
    for cset, mode in ((cs, 0), (gs, 1)):
        fast, slow = both_forms(be, x, cset, mode)
        scaled_close(fast, slow, K, D, 1e-13)


def test_grouped_responsibilities_vs_oracle(be, orc):
    """k_resp_groups + k_stats_gemm (the pmc_estep of the headline) against the oracle: VB statistics and
    E[log q(Z)], Gaussian PMC statistics and sum w log q, K = 32 / 40 / 64 incl. a ragged last group of 16"""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    for D, K, N in ((20, 32, 30000), (12, 56, 20000), (24, 64, 17000)):
        mu, cov, w = mk(K, D, 60 + K)
        x, _ = draw(mu, cov, w, N, 61)
        rs = np.random.RandomState(K)
        sw = rs.uniform(0.5, 1.5, N)
        cs, (m, W, beta, nu, ln_pi, ln_lambda) = vb_set(mu, cov, D, K, 62)
        ref = orc.vb_estep(x, sw, m, W, beta, nu, ln_pi, ln_lambda)
        flat = be.tohost(be.estep(x, cs, 0, sample_w=sw)["stats"])
        sc, S0, M1, M2, _, _ = split_stats(flat, K, D)
        x_mean, S = centred_moments(S0, M1, M2, m)
        assert_rel(S0, ref["N_comp"], rtol=1e-11, what="N_comp")
        live = ref["N_comp"] > 1e-6
        np.testing.assert_allclose(x_mean[live], ref["x_mean_comp"][live], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(S[live], ref["S"][live], rtol=1e-9, atol=1e-11)
        assert abs(sc[0] - ref["expectation_log_q_Z"]) <= 1e-10 * abs(ref["expectation_log_q_Z"]) + 1e-11
        gs, inv, ln = gauss_set(mu, cov, w)
        rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
        flat = be.tohost(be.estep(x, gs, 1, sample_w=sw)["stats"])
        sc, S0, M1, M2, _, _ = split_stats(flat, K, D)
        np.testing.assert_allclose(S0, (sw[:, None] * rho).sum(axis=0), rtol=1e-10, atol=1e-300)
        d = x[:, None, :] - mu[None]
        np.testing.assert_allclose(M1, np.einsum('n,nk,nki->ki', sw, rho, d), rtol=1e-9, atol=1e-9)
        lq = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)[0]
        assert abs(sc[3] - (sw * lq).sum()) <= 1e-11 * abs((sw * lq).sum())


def test_moderately_separated_components_keep_their_digits(be):
    """components ~25 standard deviations from the common shift: inside the limit, error of the covariances
    within 1e-10 of the per-component form"""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    D, K, N = 16, 32, 60000
    mu, cov, w = mk(K, D, 21)
    mu = mu * 4.0
    x, _ = draw(mu, cov, w, N, 22)
    gs, _, _ = gauss_set(mu, cov, w)
    fast, slow = both_forms(be, x, gs, 1)
    a, b = split_stats(fast, K, D), split_stats(slow, K, D)
    ma, ca = centred_moments(a[1], a[2], a[3], mu)
    mb, cb = centred_moments(b[1], b[2], b[3], mu)
    np.testing.assert_allclose(ma, mb, rtol=1e-12, atol=1e-12)
    scale = np.abs(np.einsum('kii->ki', cb)).max(axis=1)[:, None, None]
    assert np.abs(ca - cb).max() / scale.max() < 1e-10 and (np.abs(ca - cb) / scale).max() < 1e-10


def test_nan_sample_poisons_the_statistics_in_both_forms(be):
    D, K, N = 20, 32, 20000
    mu, cov, w = mk(K, D, 31)
    x, _ = draw(mu, cov, w, N, 32)
    x[12345, 7] = np.nan
    cs, _ = vb_set(mu, cov, D, K, 33)
    fast, slow = both_forms(be, x, cs, 0)
    from pypmc_amd.mix_adapt._stats import split_stats
    for flat in (fast, slow):
        S0 = split_stats(flat, K, D)[1]
        assert np.isnan(S0).all()


def test_k_below_the_threshold_and_small_n_use_the_per_component_kernel(be):
    D = 20
    for K, N in ((16, 30000), (32, 5000)):
        mu, cov, w = mk(K, D, 41)
        x, _ = draw(mu, cov, w, N, 42)
        cs, _ = vb_set(mu, cov, D, K, 43)
        fast, slow = both_forms(be, x, cs, 0)
        np.testing.assert_array_equal(fast, slow)
    with pytest.raises(Exception):
        be.configure("no_such_key", 1.0)
    # the default thresholds: 524288 samples per 32 components, groups of 32 at least 63 % full
    be.configure("stats_common_shift_min_n", 524288)
    be.configure("stats_common_shift_min_fill", 0.63)
    try:
        for K, N, runs in ((32, 300000, False), (64, 300000, True), (32, 600000, True), (20, 600000, False),
                           (21, 600000, True), (40, 300000, False), (41, 300000, True)):
            mu, cov, w = mk(K, D, 44)
            x, _ = draw(mu, cov, w, N, 45)
            cs, _ = vb_set(mu, cov, D, K, 46)
            fast, slow = both_forms(be, x, cs, 0)
            assert np.array_equal(fast, slow) != runs, (K, N)
    finally:
        be.configure("stats_common_shift_min_n", 0)
        be.configure("stats_common_shift_min_fill", 0)


@pytest.mark.parametrize("seed", [0, 1])
def test_sweep_every_dimension_from_8_to_64(be, seed):
    """every sample dimension the form exists for -- exact and zero-padded kernel units -- with random K >= 17
    (complete and ragged last groups of 32), ragged N, VB and Gaussian Rao-Blackwell kinds, weighted or not"""
    rs = np.random.RandomState(100 + seed)
    for D in range(8, 65):
        K = int(rs.choice([17, 31, 32, 33, 48, 49, 64, int(rs.randint(17, 90))]))
        N = int(rs.randint(16384, 24000))
        mu, cov, w = mk(K, D, 3000 + 7 * D + seed)
        x, _ = draw(mu, cov, w, N, 11 + seed)
        sw = rs.uniform(0.5, 1.5, N) if rs.rand() < 0.5 else None
        if rs.rand() < 0.5:
            cs, _ = vb_set(mu, cov, D, K, D + K)
            mode = 0
        else:
            cs = gauss_set(mu, cov, w)[0]
            mode = 1
        fast, slow = both_forms(be, x, cs, mode, sample_w=sw)
        assert not np.array_equal(fast, slow), ("the common-shift form did not run", D, K, N)
        try:
            scaled_close(fast, slow, K, D, 1e-11)
        except AssertionError as exc:
            raise AssertionError("D=%d K=%d N=%d mode=%d: %s" % (D, K, N, mode, exc))
