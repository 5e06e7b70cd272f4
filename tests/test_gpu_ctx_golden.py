"""The handle layer (include/pmc_ctx.h) against the golden vectors generated from the reference itself
(tests/golden/make_golden.py): the arrays a .pyx binding would pass, the numbers the reference's loops produced."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import brentq
from scipy.special import digamma

from conftest import load_golden

pytestmark = pytest.mark.gpu

dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
ip = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_int64))
c64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)


@pytest.fixture(scope="module")
def lib():
    from pypmc_amd import _lib
    return _lib.load()


# every golden vector twice: through a one-device context, and through ONE context of three parts (pmc_init_devices with
# the box's GPU named three times: virtual shards with their own streams, scratch and host threads, the K-sized vectors
# added in part order -- SURVEY 8(b) row 1, verdict r4 #1)
@pytest.fixture(params=[(0,), (0, 0, 0)], ids=["one_device", "three_parts"])
def ctx(lib, request):
    h = C.c_void_p()
    ids = (C.c_int * len(request.param))(*request.param)
    assert lib.pmc_init_devices(len(request.param), ids, C.byref(h)) == 0, lib.pmc_last_error()
    yield h
    assert lib.pmc_shutdown(h) == 0


def mix_from(lib, ctx, g, prefix, weights=None):
    w = c64(g[prefix + "weights"] if weights is None else weights)
    mu, inv, ln = c64(g[prefix + "mu"]), c64(g[prefix + "inv_sigma"]), c64(g[prefix + "log_norm"])
    dof = g[prefix + "dof"] if (prefix + "dof") in g else None
    student = dof is not None and np.size(dof) == len(w) and np.all(np.isfinite(np.asarray(dof, dtype=float))) \
        and np.all(np.asarray(dof, dtype=float) > 0)
    K, D = mu.shape
    h = C.c_void_p()
    rc = lib.pmc_mixture_create(ctx, 1 if student else 0, K, D, dp(w), dp(mu), dp(inv), dp(ln),
                                dp(c64(dof)) if student else None, C.byref(h))
    assert rc == 0, lib.pmc_last_error()
    return h, student


def upload(lib, ctx, x):
    x = c64(x)
    h = C.c_void_p()
    assert lib.pmc_samples_upload(ctx, dp(x), x.shape[0], x.shape[1], C.byref(h)) == 0, lib.pmc_last_error()
    return h


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    same = a == b
    return float(np.max(np.where(same, 0., np.abs(a - b) / np.maximum(np.abs(b), 1e-300))))


@pytest.mark.parametrize("name", ["gauss_d1k2", "gauss_d2k3", "gauss_d5k4", "gauss_d7k1", "gauss_d20k16",
                                  "student_d2k3", "student_d3k2", "student_d30k8"])
def test_multi_evaluate(lib, ctx, name):
    g = load_golden("logpdf_" + name)                       # mixture.pyx:112-156 on the reference
    m, _ = mix_from(lib, ctx, g, "")
    s = upload(lib, ctx, g["x"])
    N, K = g["individual"].shape
    out, ind = np.empty(N), np.empty((N, K))
    assert lib.pmc_mix_logpdf(m, s, dp(out), dp(ind)) == 0, lib.pmc_last_error()
    assert rel(out, g["out"]) < 1e-10 and rel(ind, g["individual"]) < 1e-10
    lib.pmc_mixture_destroy(m)
    lib.pmc_samples_free(s)


@pytest.mark.parametrize("tag", ["d40k32", "d24k64"])
@pytest.mark.parametrize("min_n", [32768, 128])
def test_multi_evaluate_pruned_components_far_away(lib, ctx, tag, min_n):
    """round 5: a mixture with pruned (zero-weight) components, some samples on a dead component 70 sigma from every live one --
    the reference returns log 0 = -inf there (logsumexp2D's maximum runs over ALL components).  Through the handle layer with
    the exact kernels (default threshold) and with the matrix-product form forced on from 128 samples per part."""
    g = load_golden("logpdf_dead_" + tag)
    K = len(g["weights"])
    gg = dict(weights=g["weights"], mu=g["mu"], inv_sigma=np.repeat(g["inv_sigma0"][None], K, axis=0),
              log_norm=np.full(K, float(g["log_norm0"])))
    assert lib.pmc_ctx_configure(ctx, b"maha_gemm_min_n", float(min_n)) == 0, lib.pmc_last_error()
    m, _ = mix_from(lib, ctx, gg, "")
    s = upload(lib, ctx, g["x"])
    N = len(g["x"])
    out = np.empty(N)
    assert lib.pmc_mix_logpdf(m, s, dp(out), None) == 0, lib.pmc_last_error()
    assert np.array_equal(np.isneginf(out), np.isneginf(g["out"])) and not np.isnan(out).any()
    fin = np.isfinite(g["out"])
    assert rel(out[fin], g["out"][fin]) < 1e-10
    lib.pmc_mixture_destroy(m)
    lib.pmc_samples_free(s)


@pytest.mark.parametrize("name", ["gauss_d2k3", "gauss_d5k4", "gauss_d20k16", "gauss_d1k2", "gauss_d7k1"])
def test_multi_evaluate_components_subset(lib, ctx, name):
    """multi_evaluate(x, individual=..., components=subset) on the reference (mixture.pyx:153-156): only the listed
    columns are written, the others keep what the caller had there (verdict r3 missing 3)"""
    g = load_golden("logpdf_" + name)
    m, _ = mix_from(lib, ctx, g, "")
    s = upload(lib, ctx, g["x"])
    N, K = g["individual"].shape
    sub = np.ascontiguousarray(g["subset"], dtype=np.int32)
    ind = np.full((N, K), -7.25)
    rc = lib.pmc_mix_logpdf_components(m, s, sub.ctypes.data_as(C.POINTER(C.c_int32)), len(sub), dp(ind))
    assert rc == 0, lib.pmc_last_error()
    cols = list(sub)
    assert rel(ind[:, cols], np.asarray(g["individual_subset"])[:, cols]) < 1e-10
    rest = [k for k in range(K) if k not in cols]
    assert (ind[:, rest] == -7.25).all()
    bad = np.array([K], dtype=np.int32)
    assert lib.pmc_mix_logpdf_components(m, s, bad.ctypes.data_as(C.POINTER(C.c_int32)), 1, dp(ind)) < 0
    assert lib.pmc_mix_logpdf_components(m, s, None, 1, dp(ind)) < 0
    lib.pmc_mixture_destroy(m)
    lib.pmc_samples_free(s)


@pytest.mark.parametrize("name", ["gauss_d2", "student_d5"])
def test_importance_weights(lib, ctx, name):
    g = load_golden("is_" + name)                           # importance_sampling.py:197-215, convergence.py
    q, _ = mix_from(lib, ctx, g, "prop_")
    s = upload(lib, ctx, g["samples"])
    N = len(g["samples"])
    w, sums = np.empty(N), np.empty(3)
    assert lib.pmc_is_weights(q, s, dp(c64(g["target_values"]).reshape(-1)), None, dp(w), None, dp(sums)) == 0, \
        lib.pmc_last_error()
    assert rel(w, np.asarray(g["weights"]).reshape(-1)) < 1e-10
    perp = np.exp(-(sums[1] / sums[0] - np.log(sums[0]))) / N
    ess = sums[0] ** 2 / sums[2] / N
    assert abs(perp - float(g["perp"])) < 1e-10 and abs(ess - float(g["ess"])) < 1e-10
    lib.pmc_mixture_destroy(q)
    lib.pmc_samples_free(s)


@pytest.mark.parametrize("name,weighted", [("d2k3", False), ("d5k4w", True), ("d20k8", False), ("d3k5first", True)])
def test_vb_estep(lib, ctx, name, weighted):
    g = load_golden("vb_" + name)                           # variational.pyx:116-127 on the reference, two stages
    x = c64(g["data"])
    N, D = x.shape
    sw = None
    if weighted:
        sw = c64(g["sample_weights"])
        sw = c64(N * (sw / sw.sum()))                       # variational.pyx:94
    s = upload(lib, ctx, x)
    for stage in ("e0_", "u1_"):
        K = len(g[stage + "nu"])
        Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
        r, lr = np.empty((N, K)), np.empty((N, K))
        rc = lib.pmc_vb_estep(ctx, s, dp(sw), K, dp(c64(g[stage + "m"])), dp(c64(g[stage + "W"])), dp(c64(g[stage + "nu"])),
                              dp(c64(g[stage + "beta"])), dp(c64(g[stage + "expectation_ln_pi"])),
                              dp(c64(g[stage + "expectation_det_ln_lambda"])), None, dp(Nk), dp(xbar), dp(S), dp(elq),
                              dp(r), dp(lr))
        assert rc == 0, lib.pmc_last_error()
        np.testing.assert_allclose(Nk, g[stage + "N_comp"], rtol=1e-10, err_msg=stage)
        np.testing.assert_allclose(xbar, g[stage + "x_mean_comp"], rtol=1e-10, atol=1e-13, err_msg=stage)
        np.testing.assert_allclose(S, g[stage + "S"], rtol=1e-10, atol=1e-12, err_msg=stage)
        assert rel(r, g[stage + "r"]) < 1e-10, stage
        ref_lr = np.asarray(g[stage + "log_rho"])
        assert np.max(np.abs(lr - ref_lr) / np.maximum(np.abs(ref_lr), 1e-3)) < 1e-10, stage
        if (stage + "log_q_Z") in g:
            assert abs(elq[0] - float(g[stage + "log_q_Z"])) <= 1e-10 * abs(float(g[stage + "log_q_Z"])), stage
    lib.pmc_samples_free(s)


def test_vb_estep_headline_shape(lib, ctx):
    """K = 32, D = 20: the reference's first E-step of the BASELINE metric's shape through the handle layer, and through
    the kernel-level call the Python front-end makes"""
    from pypmc_amd.backend import HipBackend, ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    g = load_golden("vb_d20k32")
    x = c64(g["data"])
    N, D = x.shape
    p = lambda k: c64(g["e0_" + k])
    K = len(p("nu"))
    s = upload(lib, ctx, x)
    Nk, xbar, S, elq, r = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1), np.empty((N, K))
    assert lib.pmc_vb_estep(ctx, s, None, K, dp(p("m")), dp(p("W")), dp(p("nu")), dp(p("beta")), dp(p("expectation_ln_pi")),
                            dp(p("expectation_det_ln_lambda")), None, dp(Nk), dp(xbar), dp(S), dp(elq), dp(r), None) == 0, \
        lib.pmc_last_error()
    np.testing.assert_allclose(Nk, p("N_comp"), rtol=1e-10)
    np.testing.assert_allclose(xbar, p("x_mean_comp"), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(S, p("S"), rtol=1e-10, atol=1e-12)
    assert rel(r, p("r")) < 1e-10
    assert abs(elq[0] - float(g["e0_log_q_Z"])) <= 1e-10 * abs(float(g["e0_log_q_Z"]))
    lib.pmc_samples_free(s)
    # the same step as HipBackend.estep issues it, with the common-shift statistics forced on at this small N
    be = HipBackend()
    cs = ComponentSet(2, p("m"), p("W"), c0=D / p("beta"), c1=p("nu"), c2=p("expectation_ln_pi"),
                      c3=p("expectation_det_ln_lambda") - D * np.log(2. * np.pi))
    be.configure("stats_common_shift_min_n", 0)
    try:
        flat = be.tohost(be.estep(x, cs, 0)["stats"])
    finally:
        be.configure("stats_common_shift_min_n", 524288)
    sc, S0, M1, M2, _, _ = split_stats(flat, K, D)
    xb, Sg = centred_moments(S0, M1, M2, p("m"))
    np.testing.assert_allclose(S0, p("N_comp"), rtol=1e-10)
    np.testing.assert_allclose(xb, p("x_mean_comp"), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(Sg, p("S"), rtol=1e-10, atol=1e-12)
    assert abs(sc[0] - float(g["e0_log_q_Z"])) <= 1e-10 * abs(float(g["e0_log_q_Z"]))


def _check_update(g, prefix, alpha, mu, sigma, live, what):
    ref_w = np.asarray(g[prefix + "weights"], dtype=float)
    np.testing.assert_allclose(alpha[live] / alpha[live].sum(), ref_w[live] / ref_w[live].sum(), rtol=1e-10, err_msg=what)
    np.testing.assert_allclose(mu[live], np.asarray(g[prefix + "mu"])[live], rtol=1e-10, atol=1e-12, err_msg=what)
    np.testing.assert_allclose(sigma[live], np.asarray(g[prefix + "sigma"])[live], rtol=1e-10, atol=1e-12, err_msg=what)


@pytest.mark.parametrize("tag", ["d2k3", "d5k4"])
def test_gaussian_pmc(lib, ctx, tag):
    g = load_golden("pmc_gauss_" + tag)                     # pmc.pyx:120-246 on the reference
    x, iw, latent = c64(g["samples"]), c64(g["weights"]), np.ascontiguousarray(g["latent"], dtype=np.int64)
    K, D = np.asarray(g["in_mu"]).shape
    s = upload(lib, ctx, x)
    q, _ = mix_from(lib, ctx, g, "in_")
    for cname, w, lat, rb in (("rb_w_", iw, None, 1), ("rb_u_", None, None, 1), ("nrb_w_", iw, latent, 0), ("nrb_u_", None, latent, 0)):
        alpha, mu, sigma = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D))
        rc = lib.pmc_pmc_update_stats(ctx, q, s, dp(w), 0, ip(lat), rb, dp(alpha), dp(mu), dp(sigma), None, None, None)
        assert rc == 0, (cname, lib.pmc_last_error())
        _check_update(g, cname, alpha, mu, sigma, list(range(K)), tag + " " + cname)
    lib.pmc_mixture_destroy(q)
    # a dead component (pmc.pyx:66): its rows stay untouched, the others are adapted as the reference adapts them
    q, _ = mix_from(lib, ctx, g, "in_", weights=g["dead_in_weights"])
    alpha, mu, sigma = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D))
    assert lib.pmc_pmc_update_stats(ctx, q, s, dp(iw), 0, None, 1, dp(alpha), dp(mu), dp(sigma), None, None, None) == 0
    live = [k for k in range(K) if k != 1]
    _check_update(g, "dead_rb_w_", alpha, mu, sigma, live, tag + " dead")
    assert alpha[1] == 0. and not sigma[1].any()
    lib.pmc_mixture_destroy(q)
    lib.pmc_samples_free(s)


@pytest.mark.parametrize("tag", ["d2k3", "d4k3"])
def test_student_t_pmc(lib, ctx, tag):
    g = load_golden("pmc_student_" + tag)                   # pmc.pyx:499-739 on the reference
    x, iw, latent = c64(g["samples"]), c64(g["weights"]), np.ascontiguousarray(g["latent"], dtype=np.int64)
    K, D = np.asarray(g["in_mu"]).shape
    s = upload(lib, ctx, x)
    q, student = mix_from(lib, ctx, g, "in_")
    assert student
    for cname, w, lat, rb, solve, lo, hi in (("rb_w_dof_", iw, None, 1, True, 1e-5, 1e3), ("rb_w_nodof_", iw, None, 1, False, 0, 0),
                                            ("rb_u_dof_", None, None, 1, True, 1e-5, 1e3),
                                            ("nrb_w_dof_", iw, latent, 0, True, 1e-5, 1e3),
                                            ("nrb_u_nodof_", None, latent, 0, False, 0, 0),
                                            ("rb_w_clamp_", iw, None, 1, True, 5., 5.5)):
        alpha, mu, sigma, const = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D)), np.zeros(K)
        rc = lib.pmc_pmc_update_stats(ctx, q, s, dp(w), 0, ip(lat), rb, dp(alpha), dp(mu), dp(sigma), dp(const), None, None)
        assert rc == 0, (cname, lib.pmc_last_error())
        _check_update(g, cname, alpha, mu, sigma, list(range(K)), tag + " " + cname)
        ref_dof = np.asarray(g[cname + "dof"], dtype=float)
        if not solve:
            np.testing.assert_array_equal(ref_dof, np.asarray(g["in_dof"], dtype=float))
            continue
        for k in range(K):                                  # the host's part: _DOFCondition + brentq (pmc.pyx:478-497, :693-710)
            cond = lambda nu: const[k] + np.log(.5 * nu) - digamma(.5 * nu)
            try:
                nu = brentq(cond, lo, hi, maxiter=100)
            except ValueError:
                nu = lo if cond(lo) < 0. else hi
            assert abs(nu - ref_dof[k]) <= 1e-8 * ref_dof[k], (tag, cname, k, nu, ref_dof[k])
    lib.pmc_mixture_destroy(q)
    lib.pmc_samples_free(s)
