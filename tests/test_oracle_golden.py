"""Pin the CPU oracle (oracle/pmc_oracle.c) against golden vectors generated from the
reference itself (tests/golden/make_golden.py) and the reference's known-answer values."""
import numpy as np
import pytest
from scipy.special import digamma, gammaln

from conftest import load_golden
from oracle import oracle as orc


def student_consts(dof, D):
    return -.5 * (dof + D), 1. / dof


def test_kat_values():
    g = load_golden("kat")
    # bit-exact where the loop order is the reference's own
    assert orc.bilinear_sym(g["bil_matrix"], g["bil_vector"]) == float(g["bil_ref"])
    assert abs(float(g["bil_ref"]) - float(g["bil_pinned"])) < 1e-7
    assert orc.logsumexp(g["lse_values"], g["lse_weights"]) == float(g["lse_ref"])
    assert abs(float(g["lse_ref"]) - float(g["lse_pinned"])) < 1e-7
    np.testing.assert_array_equal(orc.logsumexp2D(g["lse2_values"], g["lse2_weights"]), g["lse2_ref"])
    np.testing.assert_allclose(g["lse2_ref"], g["lse2_pinned"])
    out = orc.gauss_multi_evaluate(g["gauss_point"][None, :], g["gauss_mean"], g["gauss_inv_sigma"],
                                   float(g["gauss_log_norm"]))
    assert out[0] == float(g["gauss_ref"])
    assert abs(out[0] - float(g["gauss_pinned"])) < 1e-7
    pf, idf = student_consts(float(g["t_dof"]), 2)
    out = orc.student_t_multi_evaluate(g["t_points"], g["t_mean"], g["t_inv_sigma"],
                                       float(g["t_log_norm"]), pf, idf)
    np.testing.assert_array_equal(out, g["t_ref"])
    np.testing.assert_allclose(out, g["t_pinned"], atol=1e-9)
    # Cauchy 1-D: lognorm = gammaln(1) - gammaln(.5) - .5 log(pi)
    ln = gammaln(1.) - gammaln(.5) - .5 * np.log(np.pi)
    out = orc.student_t_multi_evaluate(g["cauchy_x"][None, :], np.zeros(1), np.ones((1, 1)), ln, -1., 1.)
    assert abs(out[0] - float(g["cauchy_pinned"])) < 1e-14
    assert abs(orc.perp(g["perp_weights"]) - float(g["perp_ref"])) < 1e-15
    assert abs(float(g["perp_ref"]) - float(g["perp_pinned"])) < 1e-15
    assert abs(orc.ess(g["perp_weights"]) - float(g["ess_ref"])) < 1e-15


@pytest.mark.parametrize("tag", ["d2k3", "d5k4", "d20k16", "d1k2", "d7k1"])
def test_gauss_mixture_logpdf_bitexact(tag):
    g = load_golden("logpdf_gauss_" + tag)
    out, ind = orc.mixture_multi_evaluate(0, g["x"], g["weights"], g["mu"], g["inv_sigma"], g["log_norm"])
    np.testing.assert_array_equal(ind, g["individual"])
    np.testing.assert_array_equal(out, g["out"])
    # components= subset fills only those columns
    ind2 = np.zeros_like(ind)
    orc.mixture_multi_evaluate(0, g["x"], g["weights"], g["mu"], g["inv_sigma"], g["log_norm"],
                               components=list(g["subset"]), individual=ind2)
    np.testing.assert_array_equal(ind2, g["individual_subset"])
    # zero weight component participates in the row maximum
    out0, _ = orc.mixture_multi_evaluate(0, g["x"], g["weights_zero0"], g["mu"], g["inv_sigma"], g["log_norm"])
    np.testing.assert_array_equal(out0, g["out_zero0"])
    np.testing.assert_allclose(out[:5], g["evaluate_first5"], rtol=1e-14)
    # threaded variant: same per-sample arithmetic
    outm, indm = orc.mixture_multi_evaluate(0, g["x"], g["weights"], g["mu"], g["inv_sigma"],
                                            g["log_norm"], mt=True)
    np.testing.assert_array_equal(outm, out)
    np.testing.assert_array_equal(indm, ind)


@pytest.mark.parametrize("tag", ["d40k32", "d24k64"])
def test_gauss_mixture_with_dead_components_far_away_bitexact(tag):
    """pruned (zero-weight) components take part in the row maximum (logsumexp2D, _regularize.pyx:73-77): samples that sit on a
    dead component 70 sigma from every live one come out of the reference as log 0 = -inf -- and of the oracle, bit for bit"""
    g = load_golden("logpdf_dead_" + tag)
    K = len(g["weights"])
    inv = np.repeat(g["inv_sigma0"][None], K, axis=0)
    out, ind = orc.mixture_multi_evaluate(0, g["x"], g["weights"], g["mu"], inv, np.full(K, float(g["log_norm0"])))
    np.testing.assert_array_equal(out, g["out"])
    live = g["weights"] > 0
    np.testing.assert_array_equal(ind[:, live].max(axis=1), g["individual_live_max"])
    np.testing.assert_array_equal(ind[:, ~live].max(axis=1), g["individual_dead_max"])
    assert np.isneginf(out).sum() == 81 and np.isfinite(out[:384]).all()


@pytest.mark.parametrize("tag", ["d40k32", "d64k64"])
def test_student_mixture_logpdf_large_dimensions_bitexact(tag):
    g = load_golden("logpdf_student_shared_" + tag)
    K, D = g["mu"].shape
    pf, idf = student_consts(g["dof"], D)
    inv = np.repeat(g["inv_sigma0"][None], K, axis=0)
    out, _ = orc.mixture_multi_evaluate(1, g["x"], g["weights"], g["mu"], inv, g["log_norm"], pf, idf)
    np.testing.assert_array_equal(out, g["out"])


@pytest.mark.parametrize("tag", ["d3k2", "d30k8", "d2k3"])
def test_student_mixture_logpdf_bitexact(tag):
    g = load_golden("logpdf_student_" + tag)
    D = g["x"].shape[1]
    pf, idf = student_consts(g["dof"], D)
    out, ind = orc.mixture_multi_evaluate(1, g["x"], g["weights"], g["mu"], g["inv_sigma"],
                                          g["log_norm"], pf, idf)
    np.testing.assert_array_equal(ind, g["individual"])
    np.testing.assert_array_equal(out, g["out"])


@pytest.mark.parametrize("tag,student", [("gauss_d2", False), ("student_d5", True)])
def test_importance_weights(tag, student):
    g = load_golden("is_" + tag)
    D = g["samples"].shape[1]
    if student:
        pf, idf = student_consts(g["prop_dof"], D)
        logq, _ = orc.mixture_multi_evaluate(1, g["samples"], g["prop_weights"], g["prop_mu"],
                                             g["prop_inv_sigma"], g["prop_log_norm"], pf, idf)
    else:
        logq, _ = orc.mixture_multi_evaluate(0, g["samples"], g["prop_weights"], g["prop_mu"],
                                             g["prop_inv_sigma"], g["prop_log_norm"])
    w = orc.is_weights(g["target_values"], logq)
    # the reference evaluates sample by sample (evaluate + 1-D logsumexp) -> same arithmetic
    np.testing.assert_allclose(w, g["weights"], rtol=1e-13)
    assert abs(orc.perp(w) - float(g["perp"])) < 1e-13
    assert abs(orc.ess(w) - float(g["ess"])) < 1e-13
    # origin array is a deterministic function of the multinomial counts (mixture.pyx:203-208)
    origin = np.repeat(np.arange(len(g["counts"])), g["counts"])
    np.testing.assert_array_equal(origin, g["origin"])


def _vb_host_expectations(alpha, nu, log_det_W, D):
    ln_lambda = np.zeros_like(nu)
    for i in range(1, D + 1):
        ln_lambda += digamma(0.5 * (nu + 1. - i))
    ln_lambda += D * np.log(2.)
    ln_lambda += log_det_W
    ln_pi = digamma(alpha) - digamma(alpha.sum())
    return ln_lambda, ln_pi


@pytest.mark.parametrize("tag", ["d2k3", "d5k4w", "d20k8", "d3k5first"])
@pytest.mark.parametrize("stage", ["e0_", "u1_"])
def test_vb_estep(tag, stage):
    g = load_golden("vb_" + tag)
    data = g["data"]
    N, D = data.shape
    sw = g["sample_weights"]
    sw = None if sw.size == 0 else N * (sw / sw.sum())     # variational.pyx:94
    p = lambda k: g[stage + k]
    ln_lambda, ln_pi = _vb_host_expectations(p("alpha"), p("nu"), p("log_det_W"), D)
    np.testing.assert_allclose(ln_lambda, p("expectation_det_ln_lambda"), rtol=1e-14)
    np.testing.assert_allclose(ln_pi, p("expectation_ln_pi"), rtol=1e-14)
    for mt in (False, True):
        res = orc.vb_estep(data, sw, p("m"), p("W"), p("beta"), p("nu"), p("expectation_ln_pi"),
                           p("expectation_det_ln_lambda"), mt=mt)
        np.testing.assert_array_equal(res["expectation_gauss_exponent"], p("expectation_gauss_exponent"))
        np.testing.assert_array_equal(res["log_rho"], p("log_rho"))
        np.testing.assert_array_equal(res["r"], p("r"))
        np.testing.assert_allclose(res["N_comp"], p("N_comp"), rtol=1e-13)
        np.testing.assert_allclose(res["x_mean_comp"], p("x_mean_comp"), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(res["S"], p("S"), rtol=1e-11, atol=1e-13)
        assert abs(res["expectation_log_q_Z"] - float(g[stage + "log_q_Z"])) <= 1e-12 * abs(float(g[stage + "log_q_Z"])) + 1e-12


def test_vb_estep_headline_shape():
    """K = 32, D = 20 (BASELINE's metric shape): the first E-step of the reference's GaussianInference, r bit for bit"""
    g = load_golden("vb_d20k32")
    p = lambda k: g["e0_" + k]
    res = orc.vb_estep(g["data"], None, p("m"), p("W"), p("beta"), p("nu"), p("expectation_ln_pi"),
                       p("expectation_det_ln_lambda"))
    np.testing.assert_array_equal(res["r"], p("r"))
    np.testing.assert_allclose(res["N_comp"], p("N_comp"), rtol=1e-13)
    np.testing.assert_allclose(res["x_mean_comp"], p("x_mean_comp"), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(res["S"], p("S"), rtol=1e-11, atol=1e-13)
    assert abs(res["expectation_log_q_Z"] - float(g["e0_log_q_Z"])) <= 1e-12 * abs(float(g["e0_log_q_Z"]))


def _gauss_lognorm(log_det, D):
    return -0.5 * D * np.log(2 * np.pi) - 0.5 * log_det


@pytest.mark.parametrize("tag", ["d2k3", "d5k4"])
def test_gaussian_pmc_reductions(tag):
    g = load_golden("pmc_gauss_" + tag)
    x = g["samples"]
    N, D = x.shape
    K = len(g["in_weights"])
    live = list(range(K))
    for case, w, latent, rb in (("rb_w", g["weights"], None, True), ("rb_u", None, None, True),
                                ("nrb_w", g["weights"], g["latent"], False),
                                ("nrb_u", None, g["latent"], False)):
        if rb:
            rho = orc.rho_rb(0, x, g["in_weights"], g["in_mu"], g["in_inv_sigma"], g["in_log_norm"],
                             None, None, live)
        else:
            rho = orc.rho_non_rb(N, K, latent, live)
        alpha, mu, cov = orc.pmc_reductions(x, rho, None, w, live)
        norm = w.sum() if w is not None else float(N)
        np.testing.assert_allclose(alpha / norm, g[case + "_weights"], rtol=1e-12)
        np.testing.assert_allclose(mu, g[case + "_mu"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(cov, g[case + "_sigma"], rtol=1e-10, atol=1e-12)
    # dead component: zero column participates in the row maximum
    wd = g["dead_in_weights"]
    live = [k for k in range(K) if wd[k] != 0]
    rho = orc.rho_rb(0, x, wd, g["in_mu"], g["in_inv_sigma"], g["in_log_norm"], None, None, live)
    alpha, mu, cov = orc.pmc_reductions(x, rho, None, g["weights"], live)
    np.testing.assert_allclose((alpha / g["weights"].sum())[live], g["dead_rb_w_weights"][live], rtol=1e-12)
    np.testing.assert_allclose(mu[live], g["dead_rb_w_mu"][live], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(cov[live], g["dead_rb_w_sigma"][live], rtol=1e-10, atol=1e-12)
    assert g["dead_rb_w_weights"][1] == 0.0
    # log likelihood of the PMC driver (pmc.pyx:388-391)
    logq, _ = orc.mixture_multi_evaluate(0, x, g["in_weights"], g["in_mu"], g["in_inv_sigma"], g["in_log_norm"])
    ll = orc.pmc_log_likelihood(logq, g["weights"] / g["weights"].sum())
    assert abs(ll - float(g["pmcrun_ll0"])) < 1e-12 * abs(ll)


@pytest.mark.parametrize("tag", ["d2k3", "d4k3"])
def test_student_t_pmc_reductions(tag):
    from scipy.optimize import brentq
    g = load_golden("pmc_student_" + tag)
    x = g["samples"]
    N, D = x.shape
    K = len(g["in_weights"])
    live = list(range(K))
    dof = g["in_dof"]
    pf, idf = student_consts(dof, D)
    for case, w, latent, rb, solve in (("rb_w_dof", g["weights"], None, True, True),
                                       ("rb_w_nodof", g["weights"], None, True, False),
                                       ("rb_u_dof", None, None, True, True),
                                       ("nrb_w_dof", g["weights"], g["latent"], False, True),
                                       ("nrb_u_nodof", None, g["latent"], False, False)):
        if rb:
            rho = orc.rho_rb(1, x, g["in_weights"], g["in_mu"], g["in_inv_sigma"], g["in_log_norm"],
                             pf, idf, live)
        else:
            rho = orc.rho_non_rb(N, K, latent, live)
        gamma = orc.student_t_gamma(x, g["in_mu"], g["in_inv_sigma"], dof, live)
        alpha, mu, cov = orc.pmc_reductions(x, rho, gamma, w, live)
        norm = w.sum() if w is not None else float(N)
        np.testing.assert_allclose(alpha / norm, g[case + "_weights"], rtol=1e-12)
        np.testing.assert_allclose(mu, g[case + "_mu"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(cov, g[case + "_sigma"], rtol=1e-10, atol=1e-12)
        if solve:
            c = orc.student_t_dof_const(x, rho, w, norm, g["in_mu"], g["in_inv_sigma"], dof,
                                        digamma(.5 * (D + dof)), digamma(.5 * dof), live)
            new = [brentq(lambda nu, ck=ck: ck + np.log(.5 * nu) - digamma(.5 * nu), 1e-5, 1e3, maxiter=100)
                   for ck in c]
            np.testing.assert_allclose(new, g[case + "_dof"], rtol=1e-9)
        else:
            np.testing.assert_array_equal(g[case + "_dof"], dof)


def test_combine_weights_both_branches():
    """importance_sampling.py:238-371 — deterministic-mixture weights, log and linear branch,
    against the reference's own output.  Same statement order; the reference exponentiates with
    numpy's SIMD exp/log, which differ from libm's by at most 1 ulp -> 4 ulp tolerance."""
    g = load_golden("combine_weights")
    samples = [g["s1"], g["s2"]]
    counts = np.array([len(s) for s in samples], dtype=np.float64)
    n_total = counts.sum()

    def q_matrix(x):
        q = np.empty((len(x), 2))
        for l, p in enumerate(("p1_", "p2_")):
            q[:, l], _ = orc.mixture_multi_evaluate(0, x, g[p + "weights"], g[p + "mu"], g[p + "inv_sigma"],
                                                    g[p + "log_norm"])
        return q

    for omegas, key, log_scale in (([g["w1"], g["w2"]], "combined_log", True),
                                   ([g["w1_zeros"], g["w2"]], "combined_linear", False)):
        got = np.concatenate([orc.combine_weights_run(q_matrix(samples[t]), counts, t, omegas[t], n_total, log_scale)
                              for t in range(2)])
        np.testing.assert_allclose(got, g[key], rtol=9e-16, atol=0)


def test_multithreaded_checkers_are_bit_identical_to_the_loops():
    """orc_pmc_reductions_mt / rho_rb(mt=True) (full-size GPU tests run them on all host cores) keep every
    component's summation order: same bits as the single-threaded restatement of pmc.pyx:23-43, :188-222"""
    from oracle import oracle as orc
    rs = np.random.RandomState(3)
    K, D, N = 7, 5, 9000
    mu = rs.normal(0, 2, (K, D))
    A = rs.normal(size=(K, D, D))
    cov = np.einsum('kij,klj->kil', A, A) / D + 0.5 * np.eye(D)
    inv = np.linalg.inv(cov)
    inv = 0.5 * (inv + inv.transpose(0, 2, 1))
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
    w = rs.uniform(0.5, 1.5, K)
    w /= w.sum()
    x = mu[rs.choice(K, N)] + rs.normal(size=(N, D))
    live = list(range(K))
    r1 = orc.rho_rb(0, x, w, mu, inv, ln, None, None, live)
    r2 = orc.rho_rb(0, x, w, mu, inv, ln, None, None, live, mt=True)
    np.testing.assert_array_equal(r1, r2)
    sw = rs.uniform(0.5, 1.5, N)
    for lv in (live, [0, 2, 3, 6]):
        for a, b in zip(orc.pmc_reductions(x, r1, None, sw, lv), orc.pmc_reductions(x, r1, None, sw, lv, mt=True)):
            np.testing.assert_array_equal(a, b)
