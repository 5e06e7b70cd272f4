"""GPU parity tests proper: the HIP kernels, called through the C ABI, against the CPU oracle on
the same seeded inputs and against the golden vectors generated from the reference."""
import numpy as np
import pytest
from scipy.special import digamma, gammaln

from conftest import load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-10      # BASELINE.json: within 1e-10 relative on log-weights and responsibilities


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    return HipBackend()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


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
    z = rs.normal(size=(N, mu.shape[1]))
    return mu[k] + np.einsum('nij,nj->ni', L[k], z), k


def gauss_set(mu, cov, w, columns=None, ld=None):
    from pypmc_amd.backend import ComponentSet
    D = mu.shape[1]
    inv = np.linalg.inv(cov)
    inv = 0.5 * (inv + inv.transpose(0, 2, 1))
    logdet = np.linalg.slogdet(cov)[1]
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * logdet
    return ComponentSet(0, mu, inv, c0=ln, weight=w, column=columns, ld=ld), inv, ln


def student_set(mu, cov, w, dof):
    from pypmc_amd.backend import ComponentSet
    D = mu.shape[1]
    inv = np.linalg.inv(cov)
    inv = 0.5 * (inv + inv.transpose(0, 2, 1))
    logdet = np.linalg.slogdet(cov)[1]
    ln = gammaln(.5 * (dof + D)) - gammaln(.5 * dof) - 0.5 * D * np.log(dof * np.pi) - 0.5 * logdet
    pf, idf = -.5 * (dof + D), 1. / dof
    return ComponentSet(1, mu, inv, c0=ln, c1=pf, c2=idf, c3=dof, weight=w), inv, ln, pf, idf


def assert_rel(a, b, rtol=RTOL, what=""):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    same = (a == b)                                   # covers matching +-inf
    assert (np.isnan(a) == np.isnan(b)).all(), what + ": NaN pattern differs"
    a, b = np.where(same, 0.0, a), np.where(same, 0.0, b)
    scale = np.maximum(np.abs(b), 1e-300)
    err = np.nanmax(np.abs(a - b) / scale) if a.size else 0.0
    assert err <= rtol, "%s: max relative error %.3e > %.1e" % (what, err, rtol)


@pytest.mark.parametrize("D,K,N", [(1, 2, 70), (2, 3, 257), (3, 1, 64), (5, 4, 300), (7, 5, 1000),
                                   (8, 2, 129), (9, 3, 200), (11, 3, 333), (16, 4, 500),
                                   (20, 16, 4096), (23, 3, 150), (30, 8, 700), (40, 6, 300),
                                   (48, 3, 130), (57, 2, 100), (64, 2, 70)])
def test_gauss_logpdf_vs_oracle(be, orc, D, K, N):
    mu, cov, w = mk(K, D, 100 + D)
    x, _ = draw(mu, cov, w, N, 7)
    cs, inv, ln = gauss_set(mu, cov, w)
    ref_out, ref_ind = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    res = be.logpdf(x, cs, want_out=True, want_individual=True)
    assert_rel(be.tohost(res["individual"]), ref_ind, what="individual")
    assert_rel(be.tohost(res["out"]), ref_out, what="out")
    # out-only and individual-only launches give bitwise the same numbers
    res2 = be.logpdf(x, cs, want_out=True)
    np.testing.assert_array_equal(be.tohost(res2["out"]), be.tohost(res["out"]))


@pytest.mark.parametrize("D,K,N,dof", [(2, 3, 200, 1.0), (3, 2, 130, 4.5), (10, 4, 500, 3.0),
                                       (30, 8, 1000, 8.0), (13, 2, 99, 50.)])
def test_student_logpdf_vs_oracle(be, orc, D, K, N, dof):
    mu, cov, w = mk(K, D, 200 + D)
    x, _ = draw(mu, cov * 1.5, w, N, 8)
    dofs = np.full(K, dof) + 0.25 * np.arange(K)
    cs, inv, ln, pf, idf = student_set(mu, cov, w, dofs)
    ref_out, ref_ind = orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, pf, idf)
    res = be.logpdf(x, cs, want_out=True, want_individual=True)
    assert_rel(be.tohost(res["individual"]), ref_ind, what="individual")
    assert_rel(be.tohost(res["out"]), ref_out, what="out")


@pytest.mark.parametrize("tag", ["d2k3", "d5k4", "d20k16", "d1k2", "d7k1"])
def test_gauss_logpdf_golden(be, tag):
    from pypmc_amd.backend import ComponentSet
    g = load_golden("logpdf_gauss_" + tag)
    K = len(g["weights"])
    cs = ComponentSet(0, g["mu"], g["inv_sigma"], c0=g["log_norm"], weight=g["weights"])
    res = be.logpdf(g["x"], cs, want_out=True, want_individual=True)
    assert_rel(be.tohost(res["out"]), g["out"], what="out")
    assert_rel(be.tohost(res["individual"]), g["individual"], what="individual")
    # components= subset: only those columns are touched
    sub = list(g["subset"])
    css = ComponentSet(0, g["mu"][sub], g["inv_sigma"][sub], c0=g["log_norm"][sub],
                       weight=g["weights"][sub], column=sub, ld=K)
    ind = be.zeros((len(g["x"]), K))
    be.logpdf(g["x"], css, want_out=False, individual=ind)
    got = be.tohost(ind)
    assert_rel(got[:, sub], g["individual_subset"][:, sub], what="subset")
    rest = [k for k in range(K) if k not in sub]
    assert (got[:, rest] == 0).all()
    # zero-weight component still takes part in the row maximum
    cs0 = ComponentSet(0, g["mu"], g["inv_sigma"], c0=g["log_norm"], weight=g["weights_zero0"])
    assert_rel(be.tohost(be.logpdf(g["x"], cs0)["out"]), g["out_zero0"], what="zero weight")


@pytest.mark.parametrize("tag", ["d3k2", "d30k8", "d2k3"])
def test_student_logpdf_golden(be, tag):
    from pypmc_amd.backend import ComponentSet
    g = load_golden("logpdf_student_" + tag)
    D = g["x"].shape[1]
    cs = ComponentSet(1, g["mu"], g["inv_sigma"], c0=g["log_norm"], c1=-.5 * (g["dof"] + D),
                      c2=1. / g["dof"], c3=g["dof"], weight=g["weights"])
    res = be.logpdf(g["x"], cs, want_out=True, want_individual=True)
    assert_rel(be.tohost(res["out"]), g["out"], what="out")
    assert_rel(be.tohost(res["individual"]), g["individual"], what="individual")


@pytest.mark.parametrize("tag,student", [("gauss_d2", False), ("student_d5", True)])
def test_importance_weights_golden(be, orc, tag, student):
    from pypmc_amd.backend import ComponentSet
    g = load_golden("is_" + tag)
    D = g["samples"].shape[1]
    if student:
        cs = ComponentSet(1, g["prop_mu"], g["prop_inv_sigma"], c0=g["prop_log_norm"],
                          c1=-.5 * (g["prop_dof"] + D), c2=1. / g["prop_dof"], c3=g["prop_dof"],
                          weight=g["prop_weights"])
    else:
        cs = ComponentSet(0, g["prop_mu"], g["prop_inv_sigma"], c0=g["prop_log_norm"],
                          weight=g["prop_weights"])
    res = be.logpdf(g["samples"], cs, log_target=g["target_values"], want_scalars=True)
    w = be.tohost(res["weights"])
    assert_rel(w, g["weights"], what="weights")
    sc = be.tohost(res["scalars"])
    N = len(w)
    perp = np.exp(-(sc[1] / sc[0] - np.log(sc[0]))) / N
    ess = sc[0] ** 2 / (N * sc[2])
    assert abs(perp - float(g["perp"])) < 1e-12
    assert abs(ess - float(g["ess"])) < 1e-12
    assert sc[4] == 0
    sc2 = be.tohost(be.weight_sums(g["weights"]))
    assert_rel(sc2[:3], [g["weights"].sum(), (g["weights"] * np.log(g["weights"])).sum(),
                         (g["weights"] ** 2).sum()], rtol=1e-13, what="weight sums")


def _vb_set(g, stage, D):
    from pypmc_amd.backend import ComponentSet
    p = lambda k: g[stage + k]
    return ComponentSet(2, p("m"), p("W"), c0=D / p("beta"), c1=p("nu"),
                        c2=p("expectation_ln_pi"),
                        c3=p("expectation_det_ln_lambda") - D * np.log(2. * np.pi))


@pytest.mark.parametrize("tag", ["d2k3", "d5k4w", "d20k8", "d3k5first"])
@pytest.mark.parametrize("stage", ["e0_", "u1_"])
def test_vb_estep_golden(be, tag, stage):
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    g = load_golden("vb_" + tag)
    data = g["data"]
    N, D = data.shape
    K = g[stage + "m"].shape[0]
    sw = g["sample_weights"]
    sw = None if sw.size == 0 else N * (sw / sw.sum())
    cs = _vb_set(g, stage, D)
    res = be.estep(data, cs, 0, sample_w=sw, want_r=True, want_log_rho=True, want_exponent=True)
    assert_rel(be.tohost(res["exponent"]), g[stage + "expectation_gauss_exponent"], what="exponent")
    assert_rel(be.tohost(res["r"]), g[stage + "r"], what="r")
    lr, lr_ref = be.tohost(res["log_rho"]), g[stage + "log_rho"]
    # log_rho: 1e-10 relative, with an absolute floor for entries that are ~0
    assert np.max(np.abs(lr - lr_ref) / np.maximum(np.abs(lr_ref), 1e-3)) < RTOL
    sc, S0, M1, M2, _, _ = split_stats(be.tohost(res["stats"]), K, D)
    x_mean, S = centred_moments(S0, M1, M2, g[stage + "m"])
    assert_rel(S0, g[stage + "N_comp"], rtol=1e-11, what="N_comp")
    np.testing.assert_allclose(x_mean, g[stage + "x_mean_comp"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(S, g[stage + "S"], rtol=1e-9, atol=1e-11)
    assert abs(sc[0] - float(g[stage + "log_q_Z"])) <= 1e-10 * abs(float(g[stage + "log_q_Z"])) + 1e-12


@pytest.mark.parametrize("D,K,N,weighted", [(1, 3, 65, True), (1, 2, 1, False), (3, 4, 129, True), (5, 2, 65, False),
                                            (2, 2, 64, False), (4, 7, 1000, True), (20, 32, 5000, False),
                                            (6, 40, 777, True), (30, 5, 300, False), (12, 9, 64 * 9 + 1, True),
                                            (17, 3, 500, False), (40, 4, 400, True), (64, 2, 200, False),
                                            (65, 3, 200, True), (97, 2, 150, False), (130, 2, 100, True)])
def test_vb_estep_vs_oracle(be, orc, D, K, N, weighted):
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    mu, cov, w = mk(K, D, 300 + D)
    x, _ = draw(mu, cov, w, N, 9)
    rs = np.random.RandomState(D * K)
    sw = rs.uniform(0.5, 1.5, N) if weighted else None
    if sw is not None:
        sw = N * sw / sw.sum()
    nu = D + 2. + rs.uniform(0, 5, K)
    beta = 1. + rs.uniform(0, 5, K)
    alpha = 1. + rs.uniform(0, 5, K)
    W = np.linalg.inv(cov) / nu[:, None, None]
    W = 0.5 * (W + W.transpose(0, 2, 1))
    m = mu + 0.1 * rs.normal(size=mu.shape)
    ln_lambda = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + \
        np.linalg.slogdet(W)[1]
    ln_pi = digamma(alpha) - digamma(alpha.sum())
    ref = orc.vb_estep(x, sw, m, W, beta, nu, ln_pi, ln_lambda)
    cs = ComponentSet(2, m, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    res = be.estep(x, cs, 0, sample_w=sw, want_r=True, want_log_rho=True, want_exponent=True)
    assert_rel(be.tohost(res["exponent"]), ref["expectation_gauss_exponent"], what="exponent")
    assert_rel(be.tohost(res["r"]), ref["r"], what="r")
    sc, S0, M1, M2, _, _ = split_stats(be.tohost(res["stats"]), K, D)
    x_mean, S = centred_moments(S0, M1, M2, m)
    assert_rel(S0, ref["N_comp"], rtol=1e-11, what="N_comp")
    live = ref["N_comp"] > 1e-6
    np.testing.assert_allclose(x_mean[live], ref["x_mean_comp"][live], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(S[live], ref["S"][live], rtol=1e-8, atol=1e-10)
    assert abs(sc[0] - ref["expectation_log_q_Z"]) <= 1e-10 * abs(ref["expectation_log_q_Z"]) + 1e-11
    # the E-step proper (no N x K matrices requested) normalises with the product form of the
    # streaming log-sum-exp instead of a second exp per pair: same statistics to rounding, the same r
    res2 = be.estep(x, cs, 0, sample_w=sw, want_r=True)
    assert_rel(be.tohost(res2["r"]), ref["r"], what="r (product form)")
    np.testing.assert_allclose(be.tohost(res2["stats"]), be.tohost(res["stats"]), rtol=1e-12, atol=1e-13)
    sc2 = split_stats(be.tohost(res2["stats"]), K, D)[0]
    assert abs(sc2[0] - ref["expectation_log_q_Z"]) <= 1e-10 * abs(ref["expectation_log_q_Z"]) + 1e-11
    # without N x K outputs the E-step is one call (one fused kernel for small D): same statistics
    # to rounding, and a second launch gives them bitwise again
    res3 = be.tohost(be.estep(x, cs, 0, sample_w=sw)["stats"]).copy()
    np.testing.assert_allclose(res3, be.tohost(res2["stats"]), rtol=1e-11, atol=1e-12)
    np.testing.assert_array_equal(be.tohost(be.estep(x, cs, 0, sample_w=sw)["stats"]), res3)


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 16])
@pytest.mark.parametrize("K,N,weighted", [(1, 70, False), (2, 1, True), (3, 257, True), (5, 64, False), (8, 1000, True),
                                          (9, 65, False), (16, 3000, True), (17, 511, False), (24, 2000, True),
                                          (31, 129, False), (32, 4097, True)])
def test_fused_estep_vs_oracle(be, orc, D, K, N, weighted):
    """pmc_estep on its one-kernel path (compiled dimension <= 16, K <= 32; every split of the components
    among the wavefronts): VB statistics and E[log q(Z)], Gaussian PMC statistics and log-likelihood
    against the restated reference loops, and against the two-kernel path."""
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    from pypmc_amd._lib import PMC_KIND_VB, PMC_KIND_GAUSS
    dp = be.lib.pmc_padded_dim(D)
    fusable = int(dp <= 7 and not (dp >= 5 and K < 9))    # otherwise pmc_estep is the two kernels (still tested here)
    assert be.lib.pmc_estep_is_fused(K, D, PMC_KIND_VB, 0) == fusable
    assert be.lib.pmc_estep_is_fused(K, D, PMC_KIND_GAUSS, 1) == fusable
    assert be.lib.pmc_estep_is_fused(K, D, 1, 1) == int(fusable and dp >= 3)       # Student-t: the LDS form only
    assert be.lib.pmc_estep_is_fused(K, D, PMC_KIND_GAUSS, 2) == 0
    mu, cov, w = mk(K, D, 900 + D + K)
    x, _ = draw(mu, cov, w, N, 19)
    rs = np.random.RandomState(D * K + N)
    sw = rs.uniform(0.5, 1.5, N) if weighted else None
    nu = D + 2. + rs.uniform(0, 5, K)
    beta = 1. + rs.uniform(0, 5, K)
    alpha = 1. + rs.uniform(0, 5, K)
    W = np.linalg.inv(cov) / nu[:, None, None]
    W = 0.5 * (W + W.transpose(0, 2, 1))
    m = mu + 0.1 * rs.normal(size=mu.shape)
    ln_lambda = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + np.linalg.slogdet(W)[1]
    ln_pi = digamma(alpha) - digamma(alpha.sum())
    ref = orc.vb_estep(x, sw, m, W, beta, nu, ln_pi, ln_lambda)
    cs = ComponentSet(2, m, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    fused = be.tohost(be.estep(x, cs, 0, sample_w=sw)["stats"]).copy()
    two = be.tohost(be.estep(x, cs, 0, sample_w=sw, want_r=True)["stats"])
    np.testing.assert_allclose(fused, two, rtol=1e-11, atol=1e-12)
    sc, S0, M1, M2, _, _ = split_stats(fused, K, D)
    assert_rel(S0, ref["N_comp"], rtol=1e-11, what="N_comp")
    x_mean, S = centred_moments(S0, M1, M2, m)
    live = ref["N_comp"] > 1e-6
    np.testing.assert_allclose(x_mean[live], ref["x_mean_comp"][live], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(S[live], ref["S"][live], rtol=1e-8, atol=1e-10)
    assert abs(sc[0] - ref["expectation_log_q_Z"]) <= 1e-10 * abs(ref["expectation_log_q_Z"]) + 1e-11
    np.testing.assert_array_equal(be.tohost(be.estep(x, cs, 0, sample_w=sw)["stats"]), fused)     # deterministic
    # Gaussian PMC, Rao-Blackwellised, with one dead component when there is more than one
    _, inv, ln = gauss_set(mu, cov, w)
    live_k = list(range(K)) if K == 1 else [k for k in range(K) if k != K // 2]
    wl = w.copy()
    if K > 1:
        wl[K // 2] = 0.
    gs = gauss_set(mu[live_k], cov[live_k], wl[live_k], columns=live_k, ld=K)[0]
    fused = be.tohost(be.estep(x, gs, 1, max_init_zero=K > 1, sample_w=sw)["stats"]).copy()
    two = be.tohost(be.estep(x, gs, 1, max_init_zero=K > 1, sample_w=sw, want_r=True)["stats"])
    np.testing.assert_allclose(fused, two, rtol=1e-11, atol=1e-12)
    rho = orc.rho_rb(0, x, wl, mu, inv, ln, None, None, live_k)[:, live_k]
    swv = np.ones(N) if sw is None else sw
    sc, S0, M1, M2, _, _ = split_stats(fused, len(live_k), D)
    np.testing.assert_allclose(S0, (swv[:, None] * rho).sum(axis=0), rtol=1e-10, atol=1e-300)
    d = x[:, None, :] - mu[None, live_k, :]
    np.testing.assert_allclose(M1, np.einsum('n,nk,nki->ki', swv, rho, d), rtol=1e-9, atol=1e-10)
    M2ref = np.einsum('n,nk,nki,nkj->kij', swv, rho, d, d)
    np.testing.assert_allclose(M2, M2ref, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(M2ref).max()))
    # Student-t PMC (pmc.pyx:602-691): u = w rho gamma and the two sums of the degree-of-freedom condition -- in the
    # one-kernel form gamma and log((maha + nu) / 2) are recovered from the parked a_nk -- against the two kernels
    dof = 2.5 + (np.arange(K) % 6) * 1.5
    ts = student_set(mu, cov, w, dof)[0]
    fused = be.tohost(be.estep(x, ts, 1, sample_w=sw)["stats"]).copy()
    two = be.tohost(be.estep(x, ts, 1, sample_w=sw, want_r=True)["stats"])
    ps = 1 + D + D * (D + 1) // 2
    a, b = fused[8:8 + K * ps].reshape(K, ps), two[8:8 + K * ps].reshape(K, ps)
    scale = np.abs(b).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(a - b) / scale).max() < 1e-11, "Student-t statistics, one kernel against two"
    np.testing.assert_allclose(fused[8 + K * ps:], two[8 + K * ps:], rtol=1e-11, atol=1e-300)
    assert abs(fused[3] - two[3]) <= 1e-12 * abs(two[3])                          # sum w log q
    np.testing.assert_array_equal(be.tohost(be.estep(x, ts, 1, sample_w=sw)["stats"]), fused)


@pytest.mark.parametrize("D,K", [(1, 3), (2, 5), (2, 32), (3, 4), (5, 9), (7, 32), (8, 5)])
def test_estep_nan_sample_poisons_the_statistics(be, orc, D, K):
    """A sample with a NaN coordinate makes every a_nk of its row NaN in the reference (variational.pyx:675-755,
    pmc.pyx:23-43) and with it every sum it enters.  The register form of the one-kernel path uses an exp
    without NaN handling and carries the NaN in the sample's weight instead: same outcome, on every path.
    (Infinite coordinates are not pinned: the reference's full quadratic form turns them into NaN, the
    triangular form here into maha = inf, i.e. rho = 0 for the PMC update.)"""
    from pypmc_amd.backend import ComponentSet
    mu, cov, w = mk(K, D, 40 + D)
    x, _ = draw(mu, cov, w, 300, 3)
    nu = D + 3. + np.arange(K)
    W = np.linalg.inv(cov) / nu[:, None, None]
    W = 0.5 * (W + W.transpose(0, 2, 1))
    cs = ComponentSet(2, mu, W, c0=np.full(K, D / 2.), c1=nu, c2=np.log(w), c3=np.linalg.slogdet(W)[1])
    gs = gauss_set(mu, cov, w)[0]
    clean_vb = be.tohost(be.estep(x, cs, 0)["stats"]).copy()
    assert np.isfinite(clean_vb).all()
    ref = orc.vb_estep(x, None, mu, W, np.full(K, 2.), nu, np.log(w), np.linalg.slogdet(W)[1] + D * np.log(2. * np.pi))
    assert np.isfinite(ref["N_comp"]).all()
    for bad in (np.nan,):
        xb = x.copy()
        xb[137, D - 1] = bad
        refb = orc.vb_estep(xb, None, mu, W, np.full(K, 2.), nu, np.log(w), np.linalg.slogdet(W)[1] + D * np.log(2. * np.pi))
        assert np.isnan(refb["N_comp"]).all(), "the reference loops give NaN here"
        for comps, mode in ((cs, 0), (gs, 1)):
            one = be.tohost(be.estep(xb, comps, mode)["stats"])
            two = be.tohost(be.estep(xb, comps, mode, want_r=True)["stats"])
            k0 = 8                                                  # [scalars | K x (sum u, sum u d, sum u d d^T) | ...]
            ps = 1 + D + D * (D + 1) // 2
            for got, path in ((one, "pmc_estep"), (two, "two kernels")):
                assert np.isnan(got[k0:k0 + K * ps:ps]).all(), (path, mode, bad)


@pytest.mark.parametrize("D,K,N", [(2, 3, 1000), (5, 9, 257), (8, 17, 4097), (20, 32, 3000), (32, 5, 640), (40, 40, 1500),
                                   (13, 7, 1), (3, 33, 65), (72, 5, 300), (129, 3, 65)])
def test_estep_from_kept_logpdf(be, orc, D, K, N):
    """pmc_mixture_logpdf_keep / pmc_importance_weights_keep leave the proposal's Mahalanobis forms on the
    device; pmc_estep_from_tiles turns them into the Rao-Blackwellised PMC statistics -- bit for bit what the
    two kernels compute from the samples (same arithmetic in the same order), Gauss and Student-t (gamma and
    the dof sums included), for the whole mixture and for a live subset with dead components in the row
    maximum; the weighting pass itself is unchanged."""
    mu, cov, w = mk(K, D, 500 + D + K)
    x, _ = draw(mu, cov, w, N, 23)
    rs = np.random.RandomState(N)
    sw = rs.uniform(0.5, 1.5, N)
    tmu, tcov, tw = mk(2, D, 77)
    target = gauss_set(tmu, tcov, tw)[0]
    dof = 3. + np.arange(K) % 5
    for family in ("gauss", "student"):
        def make(m, c, ww, **kw):
            if family == "gauss":
                return gauss_set(m, c, ww, **kw)[0]
            from pypmc_amd.backend import ComponentSet
            base = student_set(m, c, ww, kw.pop("dof"))[0]
            return ComponentSet(1, base.mu, base.precision, base.c0, base.c1, base.c2, base.c3, weight=ww,
                                column=kw.get("columns"), ld=kw.get("ld"))
        full = make(mu, cov, w) if family == "gauss" else make(mu, cov, w, dof=dof)
        plain = be.importance_weights(x, full, target, want_out=True)
        kept = be.importance_weights(x, full, target, want_out=True, keep=True)
        for key in ("weights", "out", "scalars"):
            np.testing.assert_array_equal(be.tohost(kept[key]), be.tohost(plain[key]))
        tiles = kept["tiles"]
        assert tiles.N == N and tiles.K == K and tiles.matches(full)
        # the kept values are the Mahalanobis forms: -2 (log q_k - log_norm_k) for the Gaussian family
        t = be.tohost(tiles.data).reshape(-1, K, 64)
        maha = np.concatenate([t[i].T for i in range(t.shape[0])])[:N]
        d = x[:, None, :] - mu[None]
        ref = np.einsum('nki,kij,nkj->nk', d, full.precision, d)
        np.testing.assert_allclose(maha, ref, rtol=1e-10, atol=1e-12)
        also = be.logpdf(x, full, want_out=True, keep=True)["tiles"]
        np.testing.assert_array_equal(be.tohost(also.data)[:tiles.data.numel()], be.tohost(tiles.data))
        two = be.tohost(be.estep(x, full, 1, sample_w=sw, want_r=True)["stats"])
        pre = be.tohost(be.estep_from_tiles(x, full, tiles, sample_w=sw)["stats"])
        np.testing.assert_array_equal(pre, two)
        if family == "student":
            assert np.all(pre[-2 * K:] != 0.)                       # the dof sums are there
        if K > 2:                                    # a live subset: dead components' zeros take part in the maximum
            live = [k for k in range(K) if k % 3 != 1]
            wl = w.copy()
            wl[[k for k in range(K) if k % 3 == 1]] = 0.
            sub = make(mu[live], cov[live], wl[live], columns=live, ld=K) if family == "gauss" else \
                make(mu[live], cov[live], wl[live], columns=live, ld=K, dof=dof[live])
            two = be.tohost(be.estep(x, sub, 1, max_init_zero=True, sample_w=sw, want_r=True)["stats"])
            pre = be.tohost(be.estep_from_tiles(x, sub, tiles, max_init_zero=True, sample_w=sw)["stats"])
            np.testing.assert_array_equal(pre, two)
        other = make(mu + 1e-3, cov, w) if family == "gauss" else make(mu + 1e-3, cov, w, dof=dof)
        assert not tiles.matches(other)


@pytest.mark.parametrize("tag", ["d2k3", "d5k4"])
def test_pmc_gauss_rho_and_stats(be, orc, tag):
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    g = load_golden("pmc_gauss_" + tag)
    x = g["samples"]
    N, D = x.shape
    K = len(g["in_weights"])
    cs = ComponentSet(0, g["in_mu"], g["in_inv_sigma"], c0=g["in_log_norm"], weight=g["in_weights"])
    for case, w, latent, mode in (("rb_w", g["weights"], None, 1), ("rb_u", None, None, 1),
                                  ("nrb_w", g["weights"], g["latent"], 2),
                                  ("nrb_u", None, g["latent"], 2)):
        res = be.estep(x, cs, mode, sample_w=w, latent=latent, want_r=True)
        if mode == 1:
            rho_ref = orc.rho_rb(0, x, g["in_weights"], g["in_mu"], g["in_inv_sigma"],
                                 g["in_log_norm"], None, None, list(range(K)))
        else:
            rho_ref = orc.rho_non_rb(N, K, latent, list(range(K)))
        assert_rel(be.tohost(res["r"]), rho_ref, what="rho " + case)
        sc, S0, M1, M2, _, _ = split_stats(be.tohost(res["stats"]), K, D)
        mean, cov = centred_moments(S0, M1, M2, g["in_mu"])
        norm = w.sum() if w is not None else float(N)
        assert_rel(S0 / norm, g[case + "_weights"], rtol=1e-11, what="alpha " + case)
        np.testing.assert_allclose(mean, g[case + "_mu"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(cov, g[case + "_sigma"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("tag", ["d2k3", "d4k3"])
def test_pmc_student_stats(be, orc, tag):
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    g = load_golden("pmc_student_" + tag)
    x = g["samples"]
    N, D = x.shape
    K = len(g["in_weights"])
    dof = g["in_dof"]
    cs = ComponentSet(1, g["in_mu"], g["in_inv_sigma"], c0=g["in_log_norm"], c1=-.5 * (dof + D),
                      c2=1. / dof, c3=dof, weight=g["in_weights"])
    for case, w, latent, mode in (("rb_w_nodof", g["weights"], None, 1),
                                  ("nrb_u_nodof", None, g["latent"], 2)):
        res = be.estep(x, cs, mode, sample_w=w, latent=latent)
        sc, S0g, M1, M2, V1, V2 = split_stats(be.tohost(res["stats"]), K, D)
        mean, cov = centred_moments(S0g, M1, M2, g["in_mu"], S0_cov=V1)
        norm = w.sum() if w is not None else float(N)
        assert_rel(V1 / norm, g[case + "_weights"], rtol=1e-11, what="alpha " + case)
        np.testing.assert_allclose(mean, g[case + "_mu"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(cov, g[case + "_sigma"], rtol=1e-9, atol=1e-11)
    # degree-of-freedom condition constant (pmc.pyx:654-691) from the device sums
    w = g["weights"]
    live = list(range(K))
    res = be.estep(x, cs, 1, sample_w=w)
    sc, S0g, M1, M2, V1, V2 = split_stats(be.tohost(res["stats"]), K, D)
    pf, idf = -.5 * (dof + D), 1. / dof
    rho = orc.rho_rb(1, x, g["in_weights"], g["in_mu"], g["in_inv_sigma"], g["in_log_norm"], pf, idf, live)
    c_ref = orc.student_t_dof_const(x, rho, w, w.sum(), g["in_mu"], g["in_inv_sigma"], dof,
                                    digamma(.5 * (D + dof)), digamma(.5 * dof), live)
    W = w.sum()
    total = V2 - digamma(.5 * (D + dof)) * V1 + (W - V1) * (np.log(.5 * dof) - digamma(.5 * dof)) + S0g + (W - V1)
    c = 1. - total / W
    np.testing.assert_allclose(c, c_ref, rtol=1e-9, atol=1e-11)


def test_edge_cases(be, orc):
    from pypmc_amd.backend import ComponentSet
    mu, cov, w = mk(3, 4, 1)
    cs, inv, ln = gauss_set(mu, cov, w)
    # empty input
    res = be.logpdf(np.zeros((0, 4)), cs)
    assert res["out"].shape[0] == 0
    for kind_set, mode in ((cs, 1), (ComponentSet(2, mu, inv, c0=np.ones(3), c1=np.ones(3) * 6), 0)):
        empty = be.tohost(be.estep(np.zeros((0, 4)), kind_set, mode)["stats"])      # fused path (D = 4)
        assert empty.shape[0] == be.stats_len(3, 4) and not empty.any()
    mu9, cov9, w9 = mk(3, 9, 1)                                                     # two-kernel path
    empty = be.tohost(be.estep(np.zeros((0, 9)), gauss_set(mu9, cov9, w9)[0], 1)["stats"])
    assert not empty.any()
    # the LinAlgError a caller written against the reference catches (gauss.pyx:40-48, pmc.pyx:227-244)
    assert issubclass(__import__("pypmc_amd").backend.NotPositiveDefinite, np.linalg.LinAlgError)
    # ragged tail: N = 1 and N = 65
    for N in (1, 63, 65):
        x, _ = draw(mu, cov, w, N, N)
        ref, _ = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
        assert_rel(be.tohost(be.logpdf(x, cs)["out"]), ref)
    # far-away sample: every component underflows relative to nothing -> finite log density
    x = np.full((1, 4), 1e3)
    ref, _ = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    assert_rel(be.tohost(be.logpdf(x, cs)["out"]), ref)
    # a non positive definite precision is refused with a status, not a crash
    from pypmc_amd.backend import NotPositiveDefinite
    bad = inv.copy()
    bad[1] = -np.eye(4)
    with pytest.raises(NotPositiveDefinite):
        be.pack(ComponentSet(0, mu, bad, c0=ln, weight=w))
    # unsupported dimension
    from pypmc_amd.backend import HipLibraryError
    with pytest.raises(HipLibraryError):
        be.logpdf(np.zeros((2, 1025)), ComponentSet(0, np.zeros((1, 1025)), np.eye(1025)[None]))


@pytest.mark.parametrize("cond", [1e4, 1e8, 1e10])
def test_ill_conditioned_covariances(be, orc, cond):
    """Whitened Mahalanobis form vs the reference's symmetric form when cond(Sigma) is large and the
    components are far apart relative to their widths (the case a raw-moment / global-shift
    formulation would lose digits on)."""
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    from pypmc_amd.tools._linalg import chol_inv_det
    rs = np.random.RandomState(int(np.log10(cond)))
    K, D, N = 3, 6, 3000
    mu = rs.normal(0, 1e3, size=(K, D))                       # separation >> width
    cov, inv, ln = np.empty((K, D, D)), np.empty((K, D, D)), np.empty(K)
    for k in range(K):
        Q, _ = np.linalg.qr(rs.normal(size=(D, D)))
        cov[k] = (Q * np.logspace(0, -np.log10(cond), D)).dot(Q.T)
        cov[k] = 0.5 * (cov[k] + cov[k].T)
        _, inv[k], logdet = chol_inv_det(cov[k])                # the reference's own inverse
        ln[k] = -0.5 * D * np.log(2 * np.pi) - 0.5 * logdet
    w = np.array([0.2, 0.5, 0.3])
    x, comp = draw(mu, cov, w, N, 3)
    cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
    ref_out, ref_ind = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    res = be.logpdf(x, cs, want_out=True, want_individual=True)
    own = ref_ind[np.arange(N), comp]                           # each sample under its own component
    got = be.tohost(res["individual"])[np.arange(N), comp]
    # both forms carry the conditioning of inv_sigma itself: agree to ~cond * eps, relative
    tol = max(1e-10, 50 * cond * 2.2e-16)
    # (log densities cross zero here: relative to max(|value|, 1))
    assert np.max(np.abs(got - own) / np.maximum(np.abs(own), 1.0)) < tol
    assert np.max(np.abs(be.tohost(res["out"]) - ref_out) / np.maximum(np.abs(ref_out), 1.0)) < tol
    # Rao-Blackwell statistics with per-component shift: means to ~1e-12 of the separation scale,
    # covariances relative to their own (tiny) scale
    e = be.estep(x, cs, 1)
    _, S0, M1, M2, _, _ = split_stats(be.tohost(e["stats"]), K, D)
    mean, sigma = centred_moments(S0, M1, M2, mu)
    rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, [0, 1, 2])
    _, mu_ref, cov_ref = orc.pmc_reductions(x, rho, None, None, [0, 1, 2])
    np.testing.assert_allclose(mean, mu_ref, rtol=1e-13, atol=1e-10)
    scale = np.abs(cov_ref).max(axis=(1, 2), keepdims=True)
    # the oracle (like the reference) subtracts a mean of magnitude 1e3 from data of width <= 1:
    # its own rounding is ~1e3 * eps / width; ours is bounded by the same quantity
    assert np.max(np.abs(sigma - cov_ref) / scale) < 1e-9


@pytest.mark.parametrize("D,K,KT,N,kinds", [(2, 3, 2, 257, "gg"), (5, 4, 1, 64, "gg"), (20, 32, 4, 5000, "gg"),
                                            (9, 5, 3, 1, "gg"), (30, 8, 4, 700, "tt"), (7, 3, 2, 333, "tg"),
                                            (33, 2, 5, 129, "gt"), (30, 32, 4, 4000, "tg"), (40, 6, 3, 300, "gt"),
                                            (70, 4, 3, 300, "tg"), (96, 3, 2, 129, "gt"), (150, 2, 2, 70, "gg")])
def test_importance_weights_against_a_mixture_target(be, orc, D, K, KT, N, kinds):
    """pmc_importance_weights (proposal and target in one pass over the samples): bitwise the numbers
    of the two-launch path, and the oracle's weights to 1e-10."""
    def build(kind, K_, seed):
        mu, cov, w = mk(K_, D, seed)
        if kind == "g":
            cs, inv, ln = gauss_set(mu, cov, w)
            return cs, lambda x: orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)[0]
        dof = np.full(K_, 5.0)
        cs, inv, ln, pf, idf = student_set(mu, cov, w, dof)
        return cs, lambda x: orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, pf, idf)[0]
    prop, ref_q = build(kinds[0], K, 70 + D)
    tgt, ref_t = build(kinds[1], KT, 80 + D)
    x, _ = draw(*mk(K, D, 70 + D), N, 3)
    sw = np.random.RandomState(1).uniform(0.5, 1.5, N)
    fused = be.importance_weights(x, prop, tgt, sample_w=sw, want_out=True, want_log_target=True)
    lt = be.logpdf(x, tgt)["out"]
    two = be.logpdf(x, prop, log_target=lt, sample_w=sw, want_scalars=True)
    for key_f, ref in (("weights", two["weights"]), ("out", two["out"]), ("log_target", lt)):
        np.testing.assert_array_equal(be.tohost(fused[key_f]), be.tohost(ref))
    np.testing.assert_array_equal(be.tohost(fused["scalars"]), be.tohost(two["scalars"]))
    assert_rel(be.tohost(fused["weights"]), orc.is_weights(ref_t(x), ref_q(x)), what="weights")
    # nothing but the weights and the sums requested
    lean = be.importance_weights(x, prop, tgt)
    np.testing.assert_array_equal(be.tohost(lean["weights"]), be.tohost(two["weights"]))
    assert lean["out"] is None and lean["log_target"] is None       # also for mixed families: one pass


def test_importance_weights_errors(be):
    import ctypes as C
    lib = be.lib
    rc = lib.pmc_importance_weights(None, 10, 2, None, 1, 0, None, 1, 0, None, None, None, None, None, None, None)
    assert rc == -1 and b"pmc_importance_weights" in lib.pmc_last_error()
    mu, cov, w = mk(2, 2, 1)
    g, _, _ = gauss_set(mu, cov, w)
    t = student_set(mu, cov, w, np.full(2, 3.))[0]
    pg, pt = be.pack(g), be.pack(t)
    x = be.asdevice(np.zeros((4, 2)))
    wts = be.empty(4)
    # kinds differ and no buffer for the target values: fine, the two families share one pass
    rc = lib.pmc_importance_weights(C.c_void_p(x.data_ptr()), 4, 2, C.c_void_p(pg.data_ptr()), 2, 0,
                                    C.c_void_p(pt.data_ptr()), 2, 1, None, None, C.c_void_p(wts.data_ptr()),
                                    None, None, None, None)
    assert rc == 0
    # a VB pack is not a density
    rc = lib.pmc_importance_weights(C.c_void_p(x.data_ptr()), 4, 2, C.c_void_p(pg.data_ptr()), 2, 0,
                                    C.c_void_p(pt.data_ptr()), 2, 2, None, None, C.c_void_p(wts.data_ptr()),
                                    None, None, None, None)
    assert rc == -1 and b"kinds" in lib.pmc_last_error()
    # empty input is fine
    rc = lib.pmc_importance_weights(None, 0, 2, C.c_void_p(pg.data_ptr()), 2, 0, C.c_void_p(pg.data_ptr()), 2, 0,
                                    None, None, None, None, None, None, None)
    assert rc == 0


def test_evaluate_once_with_no_samples(be):
    """N = 0 through the keeping weighting pass and pmc_estep_from_tiles: zero statistics, zero sums, no launch."""
    mu, cov, w = mk(3, 4, 5)
    g = gauss_set(mu, cov, w)[0]
    t = student_set(mu, cov, w, np.full(3, 4.))[0]
    for cs in (g, t):
        kept = be.importance_weights(np.zeros((0, 4)), cs, g, keep=True)
        assert kept["tiles"].N == 0 and not be.tohost(kept["scalars"]).any()
        out = be.tohost(be.estep_from_tiles(np.zeros((0, 4)), cs, kept["tiles"])["stats"])
        assert out.shape[0] == be.stats_len(3, 4) and not out.any()


def test_rho_where_the_reference_underflows(be, orc):
    """pmc.pyx:36-41 computes rho = exp(log q_k) w_k / (exp(lse) + tiny): for samples so far out that
    log q_k < -708 the numerator is denormal or zero although the ratio is representable.  The
    product-form normalisation of k_resp must follow the reference there, not the exact ratio."""
    D, K = 5, 4
    mu, cov, w = mk(K, D, 77)
    cs, inv, ln = gauss_set(mu, cov, w)
    rs = np.random.RandomState(3)
    v = rs.normal(size=D)
    v /= np.sqrt(v.dot(inv[0]).dot(v))                      # unit Mahalanobis length w.r.t. component 0
    # walk away from the mixture and keep the stretch where log q runs from -600 down to -800
    # (numerator normal -> denormal -> zero)
    x = mu[0] + np.linspace(20, 80, 6000)[:, None] * v[None, :]
    lq = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)[0]
    keep = (lq < -600) & (lq > -800)
    x, lq = np.ascontiguousarray(x[keep]), lq[keep]
    assert len(x) > 200 and lq.min() < -760 and lq.max() > -640
    got = be.tohost(be.estep(x, cs, 1, want_r=True)["r"])
    ref = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
    assert np.isfinite(got).all() and (got >= 0).all()
    normal = lq > -700                                       # numerator and denominator normal numbers
    np.testing.assert_allclose(got[normal], ref[normal], rtol=1e-10, atol=1e-300)
    # below: the reference's own numerator has only a few bits left; agree to those and go to 0 with it
    np.testing.assert_allclose(got[~normal], ref[~normal], rtol=1e-2, atol=1e-12)
    gone = lq < -750
    assert (ref[gone] == 0).all() and (got[gone] == 0).all()
    # the one-kernel E-step (D = 5) forms rho the same way: its N_k against the matrix just checked
    from pypmc_amd.mix_adapt._stats import split_stats
    S0 = split_stats(be.tohost(be.estep(x, cs, 1)["stats"]), K, D)[1]
    np.testing.assert_allclose(S0, got.sum(axis=0), rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("D", [1, 2, 3, 5, 7, 9, 13])
def test_statistics_with_a_lonely_last_sample(be, D):
    """Sufficient statistics against plain numpy sums when the array's last sample starts a tile of
    its own (N = 1 mod 64) -- the 16-byte pieces of the LDS-DMA then begin at the very last element
    (regression: D = 1 read 8 bytes past the end and picked up the wrong half of the piece)."""
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats
    for K in (1, 17):
        for N in (1, 2, 65, 257, 64 * 9 + 1):
            rs = np.random.RandomState(N + K + D)
            x = rs.normal(size=(N, D)) + 5
            mu = rs.normal(size=(K, D))
            w = rs.uniform(0.5, 1.5, N)
            lat = rs.randint(0, K, N)                       # one-hot responsibilities: exact sums
            cs = ComponentSet(0, mu, np.tile(np.eye(D), (K, 1, 1)), c0=np.zeros(K), weight=np.full(K, 1. / K))
            out = be.estep(x, cs, 2, sample_w=w, latent=lat)
            sc, S0, M1, M2, _, _ = split_stats(be.tohost(out["stats"]), K, D)
            u = np.zeros((N, K))
            u[np.arange(N), lat] = w
            d = x[:, None, :] - mu[None]
            np.testing.assert_allclose(S0, u.sum(0), rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(M1, np.einsum('nk,nki->ki', u, d), rtol=1e-12, atol=1e-11)
            np.testing.assert_allclose(M2, np.einsum('nk,nki,nkj->kij', u, d, d), rtol=1e-12, atol=1e-10)


def test_expected_log_q_Z_when_responsibilities_are_nearly_one_hot(be, orc):
    """E[log q(Z)] = sum r log r is tiny when every sample belongs to one component; the streaming
    form must not lose it in the cancellation of sum r a - lse (variational.pyx:1003-1013)."""
    from pypmc_amd.backend import ComponentSet
    D, K, N = 13, 24, 65
    rs = np.random.RandomState(5)
    mu = rs.normal(0, 30, size=(K, D))                       # far apart: r is one-hot to ~1e-10 or better
    cov = np.tile(np.eye(D), (K, 1, 1))
    x = mu[rs.randint(0, K, N)] + rs.normal(size=(N, D))
    nu = D + 2. + rs.uniform(0, 5, K)
    beta = 1. + rs.uniform(0, 5, K)
    W = cov / nu[:, None, None]
    ln_lambda = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + np.linalg.slogdet(W)[1]
    ln_pi = np.log(np.full(K, 1. / K))
    ref = orc.vb_estep(x, None, mu, W, beta, nu, ln_pi, ln_lambda)
    cs = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    sc = be.tohost(be.estep(x, cs, 0)["stats"])[:8]
    assert -1e-3 < ref["expectation_log_q_Z"] <= 0
    assert abs(sc[0] - ref["expectation_log_q_Z"]) <= 1e-10 * abs(ref["expectation_log_q_Z"]) + 1e-300


@pytest.mark.parametrize("D,K,N", [(3, 2, 130), (12, 5, 500), (20, 7, 700), (32, 3, 333), (40, 6, 450), (64, 2, 129),
                                   (65, 3, 200), (100, 2, 131)])
def test_student_t_pmc_vs_oracle(be, orc, D, K, N):
    """student_t_pmc's N-sized part (rho, gamma, the reductions and the dof-condition constant,
    pmc.pyx:499-691) against the restated reference loops on random inputs -- including the sample
    dimensions served by the matrix-pipe Mahalanobis engine (D >= 32)."""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    mu, cov, w = mk(K, D, 500 + D)
    x, _ = draw(mu, cov * 1.5, w, N, 13)
    rs = np.random.RandomState(D + K)
    dof = rs.uniform(2.0, 9.0, K)
    iw = rs.uniform(0.2, 2.0, N)
    cs, inv, ln, pf, idf = student_set(mu, cov, w, dof)
    live = list(range(K))
    res = be.estep(x, cs, 1, sample_w=iw, want_r=True)
    rho = orc.rho_rb(1, x, w, mu, inv, ln, pf, idf, live)
    assert_rel(be.tohost(res["r"]), rho, rtol=1e-10, what="rho")
    gamma = orc.student_t_gamma(x, mu, inv, dof, live)
    alpha_ref, mu_ref, cov_ref = orc.pmc_reductions(x, rho, gamma, iw, live)
    sc, S0g, M1, M2, V1, V2 = split_stats(be.tohost(res["stats"]), K, D)
    mean, sigma = centred_moments(S0g, M1, M2, mu, S0_cov=V1)
    assert_rel(V1, alpha_ref, rtol=1e-10, what="alpha (unnormalised)")
    np.testing.assert_allclose(mean, mu_ref, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(sigma, cov_ref, rtol=1e-8, atol=1e-10)
    c_ref = orc.student_t_dof_const(x, rho, iw, iw.sum(), mu, inv, dof, digamma(.5 * (D + dof)),
                                    digamma(.5 * dof), live)
    W = iw.sum()
    total = V2 - digamma(.5 * (D + dof)) * V1 + (W - V1) * (np.log(.5 * dof) - digamma(.5 * dof)) + S0g + (W - V1)
    np.testing.assert_allclose(1. - total / W, c_ref, rtol=1e-9, atol=1e-11)



def test_kernel_timings_through_the_abi(be):
    """pmc_timing_enable / pmc_get_timings: launches, milliseconds and the algorithmic work per kernel"""
    K, D, N = 5, 12, 20000
    mu, cov, w = mk(K, D, 4)
    x, _ = draw(mu, cov, w, N, 5)
    cs = gauss_set(mu, cov, w)[0]
    be.kernel_timings()                                   # clear
    be.kernel_timing(True)
    try:
        be.logpdf(x, cs)
        be.logpdf(x, cs, log_target=np.zeros(N), want_scalars=True)
        be.estep(x, cs, 1)
    finally:
        be.kernel_timing(False)
    t = be.kernel_timings()
    assert t["k_logpdf"]["calls"] == 2 and t["k_resp"]["calls"] == 1 and t["k_stats"]["calls"] == 1
    assert t["finishing reductions"]["calls"] == 3        # scalars of the weights call; scalars + statistics of the E-step
    pair = D * D + 4 * D + 40
    assert t["k_logpdf"]["flops"] == 2.0 * N * K * pair and t["k_resp"]["flops"] == 1.0 * N * K * pair
    assert t["k_stats"]["flops"] == 1.0 * N * K * (1 + 2 * D + D * (D + 1))
    assert t["k_logpdf"]["bytes"] == 2 * 8.0 * N * (D + 1) and t["k_stats"]["bytes"] == 8.0 * N * (D + K)
    assert all(0 < v["ms"] < 50 for v in t.values())
    assert be.kernel_timings() == {}                      # cleared by the read; nothing recorded while off
    be.logpdf(x, cs)
    assert be.kernel_timings() == {}
    # the fused small-dimension E-step reports itself
    mu, cov, w = mk(3, 2, 4)
    x, _ = draw(mu, cov, w, 5000, 5)
    be.kernel_timing(True)
    be.estep(x, gauss_set(mu, cov, w)[0], 1)
    be.kernel_timing(False)
    t = be.kernel_timings()
    assert t["k_estep_fused"]["calls"] == 1 and t["k_estep_fused"]["bytes"] == 8.0 * 5000 * 2 and "k_resp" not in t


@pytest.mark.parametrize("D,K,N", [(2, 300, 500), (3, 1000, 130), (5, 33, 1000), (8, 129, 300), (20, 200, 257),
                                   (40, 130, 100), (64, 70, 65), (1, 500, 64), (1, 33, 300), (1, 50, 1000), (1, 64, 129),
                                   (1, 65, 200), (2, 33, 200), (80, 70, 100)])
def test_many_components(be, orc, D, K, N):
    """K far beyond the component counts the kernels are tuned for (and beyond the one-kernel E-step's 32;
    at D = 1 its register form takes up to 64 components, 8 per wavefront)"""
    assert be.lib.pmc_estep_is_fused(K, D, 0, 1) == int(D == 1 and K <= 64)
    from pypmc_amd.mix_adapt._stats import split_stats
    mu, cov, w = mk(K, D, 1200 + D)
    x, _ = draw(mu, cov, w, N, 21)
    cs, inv, ln = gauss_set(mu, cov, w)
    ref_q, ref_ind = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    assert_rel(be.tohost(be.logpdf(x, cs)["out"]), ref_q, what="log q")
    iw = np.random.RandomState(K).uniform(0.1, 2, N)
    rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
    got = be.tohost(be.estep(x, cs, 1, sample_w=iw, want_r=True)["r"])
    # where the reference's numerator exp(log q_k) is a normal number (below it keeps only a few bits; at D = 80 the
    # row's normalisation is ~1e-50, so such a rho is far from tiny itself)
    normal = (rho > 1e-280) & (ref_ind > -690)
    assert_rel(got[normal], rho[normal], what="rho")
    S0 = split_stats(be.tohost(be.estep(x, cs, 1, sample_w=iw)["stats"]), K, D)[1]
    np.testing.assert_allclose(S0, (iw[:, None] * rho).sum(axis=0), rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("D,K", [(1, 2), (4, 3), (20, 4), (30, 8), (40, 5), (70, 2)])
def test_student_t_component_whose_mahalanobis_form_overflows(be, orc, D, K):
    """verdict r4: a point 1e160 from a Student-t component -- maha = 1e320 = inf, log(1 + maha / nu) = +inf, the
    component's value -inf (student_t.pyx:159-164), the mixture's value -inf when every component's form overflows
    (_regularize.pyx:72-81).  ONE far coordinate per such row: the reference's bilinear form (_linalg.pyx:32-37) then
    adds finite cross terms to an infinite square and overflows cleanly (with two far coordinates it multiplies
    inf by M_ij and subtracts infinities -- NaN, an artefact of its summation that |R d|^2 does not share, and not
    compared here).  Against the oracle, -inf for -inf."""
    rs = np.random.RandomState(50 + D)
    mu, cov, w = mk(K, D, 60 + D)
    dof = rs.uniform(2., 9., K)
    cs, inv, ln, pf, idf = student_set(mu, cov, w, dof)
    x, _ = draw(mu, cov, w, 300, 3)
    far = np.arange(0, 300, 7)
    x[far, rs.randint(0, D, len(far))] = 1e160 * rs.choice([-1., 1.], len(far))
    with np.errstate(all="ignore"):
        ref, ref_ind = orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, pf, idf)
    assert np.isneginf(ref_ind[far]).all() and np.isneginf(ref[far]).all() and not np.isnan(ref_ind).any()
    res = be.logpdf(x, cs, want_individual=True)
    ind, out = be.tohost(res["individual"]), be.tohost(res["out"])
    assert_rel(ind, ref_ind, what="a_nk with overflowing forms")
    assert_rel(out, ref, what="log q with overflowing forms")
    assert np.isneginf(out[far]).all()
    # importance weights of the all-overflow rows: exp(log P - (-inf)) = inf; the finite rows are untouched
    lt = rs.normal(size=300)
    wts = be.tohost(be.logpdf(x, cs, log_target=lt, want_scalars=True)["weights"])
    ok = np.isfinite(ref)
    assert_rel(wts[ok], np.exp(lt[ok] - ref[ok]), what="weights of the finite rows")
    assert np.isposinf(wts[~ok]).all()
