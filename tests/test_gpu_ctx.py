"""The handle layer of the C ABI (include/pmc_ctx.h): host pointers in, host pointers out, the reference's conventions --
what SURVEY section 8(b) lists as pmc_init / pmc_mixture_create / pmc_samples_upload / pmc_is_weights / pmc_vb_estep /
pmc_pmc_update_stats.  Called through ctypes exactly as a .pyx binding would call it, checked against the oracle (the
restated reference loops) and against the Python front-end on the same HIP kernels."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import brentq
from scipy.special import digamma

pytestmark = pytest.mark.gpu

dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
ip = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_int64))


def mk(K, D, seed, spread=3.0):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, spread, size=(K, D))
    A = rs.normal(size=(K, D, D))
    cov = np.einsum('kij,klj->kil', A, A) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K)
    return mu, cov, w / w.sum()


@pytest.fixture(scope="module")
def lib():
    from pypmc_amd import _lib
    return _lib.load()


@pytest.fixture()
def ctx(lib):
    h = C.c_void_p()
    assert lib.pmc_init(0, C.byref(h)) == 0, lib.pmc_last_error()
    yield h
    assert lib.pmc_shutdown(h) == 0


def make_mix(lib, ctx, mixture):
    """pmc_mix handle from a front-end MixtureDensity: exactly the arrays its components hold"""
    comps = mixture.components
    K, D = len(comps), mixture.dim
    student = hasattr(comps[0], "dof")
    w = np.ascontiguousarray(mixture.weights, dtype=np.float64)
    mu = np.ascontiguousarray([c.mu for c in comps], dtype=np.float64)
    inv = np.ascontiguousarray([c.inv_sigma for c in comps], dtype=np.float64)
    ln = np.ascontiguousarray([c.log_normalization for c in comps], dtype=np.float64)
    dof = np.ascontiguousarray([c.dof for c in comps], dtype=np.float64) if student else None
    h = C.c_void_p()
    rc = lib.pmc_mixture_create(ctx, 1 if student else 0, K, D, dp(w), dp(mu), dp(inv), dp(ln), dp(dof), C.byref(h))
    assert rc == 0, lib.pmc_last_error()
    return h


def upload(lib, ctx, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = C.c_void_p()
    assert lib.pmc_samples_upload(ctx, dp(x), x.shape[0], x.shape[1], C.byref(h)) == 0, lib.pmc_last_error()
    return h


@pytest.mark.parametrize("student,D,K,N", [(False, 5, 3, 1000), (False, 20, 32, 70001), (True, 30, 8, 5003),
                                           (False, 2, 1, 1), (True, 70, 4, 777)])
def test_mix_logpdf_matches_oracle_and_front_end(lib, ctx, student, D, K, N):
    from oracle import oracle as orc
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    mu, cov, w = mk(K, D, 3)
    mixture = create_t_mixture(mu, cov, np.full(K, 6.5), w) if student else create_gaussian_mixture(mu, cov, w)
    np.random.seed(4)
    x = mixture.propose(N)
    m, s = make_mix(lib, ctx, mixture), upload(lib, ctx, x)
    assert lib.pmc_samples_count(s) == N
    out, ind = np.empty(N), np.empty((N, K))
    assert lib.pmc_mix_logpdf(m, s, dp(out), dp(ind)) == 0, lib.pmc_last_error()
    ref_ind = np.empty((N, K))
    ref = mixture.multi_evaluate(x, individual=ref_ind)
    np.testing.assert_array_equal(out, ref)                       # the same kernels, the same pack
    np.testing.assert_array_equal(ind, ref_ind)
    comps = mixture.components
    inv = np.array([c.inv_sigma for c in comps])
    ln = np.array([c.log_normalization for c in comps])
    if student:
        o, _ = orc.mixture_multi_evaluate(1, x, mixture.weights, mu, inv, ln, prefactor=np.full(K, -.5 * (6.5 + D)),
                                          inv_dof=np.full(K, 1. / 6.5))
    else:
        o, _ = orc.mixture_multi_evaluate(0, x, mixture.weights, mu, inv, ln)
    assert np.max(np.abs(out - o) / np.abs(o)) < 1e-10
    back = np.empty_like(x)
    assert lib.pmc_samples_download(s, dp(back)) == 0
    np.testing.assert_array_equal(back, x)
    assert lib.pmc_samples_free(s) == 0 and lib.pmc_mixture_destroy(m) == 0


def test_is_weights_both_target_forms(lib, ctx):
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    from pypmc_amd.tools.convergence import perp, ess
    D, N = 12, 40001
    prop = create_t_mixture(*mk(6, D, 1)[:2], np.full(6, 9.), mk(6, D, 1)[2])
    tgt = create_gaussian_mixture(*mk(3, D, 2, spread=1.0))
    np.random.seed(5)
    x = prop.propose(N)
    q, t, s = make_mix(lib, ctx, prop), make_mix(lib, ctx, tgt), upload(lib, ctx, x)
    w1, lt, sums = np.empty(N), np.empty(N), np.empty(3)
    assert lib.pmc_is_weights(q, s, None, t, dp(w1), dp(lt), dp(sums)) == 0, lib.pmc_last_error()
    log_p, log_q = tgt.multi_evaluate(x), prop.multi_evaluate(x)
    np.testing.assert_array_equal(lt, log_p)
    ref = np.exp(log_p - log_q)                                    # importance_sampling.py:204-207
    assert np.max(np.abs(w1 - ref) / ref) < 1e-10
    np.testing.assert_allclose(sums, [ref.sum(), (ref * np.log(ref)).sum(), (ref ** 2).sum()], rtol=1e-10)
    # perp / ess from the three sums (convergence.py:31-39, :67-72)
    wn = ref / ref.sum()
    assert abs(np.exp(-(sums[1] / sums[0] - np.log(sums[0]))) / N - perp(ref)) < 1e-10
    assert abs(sums[0] ** 2 / sums[2] / N - 1. / (1. + np.mean((N * wn - 1) ** 2))) < 1e-10 and 0 < ess(ref) <= 1
    # the caller's own target values
    w2, sums2 = np.empty(N), np.empty(3)
    assert lib.pmc_is_weights(q, s, dp(np.ascontiguousarray(log_p)), None, dp(w2), None, dp(sums2)) == 0
    np.testing.assert_array_equal(w2, w1)
    np.testing.assert_array_equal(sums2, sums)
    assert lib.pmc_is_weights(q, s, None, None, dp(w2), None, None) < 0 and b"one of them" in lib.pmc_last_error()
    for h in (q, t):
        lib.pmc_mixture_destroy(h)
    lib.pmc_samples_free(s)


def vb_arrays(vb):
    return [np.ascontiguousarray(a, dtype=np.float64) for a in
            (vb.m, vb.W, vb.nu, vb.beta, vb.expectation_ln_pi, vb.expectation_det_ln_lambda)]


@pytest.mark.parametrize("D,K,N,weighted", [(3, 4, 2000, False), (20, 32, 60000, True), (20, 8, 30011, False),
                                            (40, 24, 20000, True)])
def test_vb_estep_reference_conventions(lib, ctx, D, K, N, weighted):
    from oracle import oracle as orc
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.variational import GaussianInference
    mixture = create_gaussian_mixture(*mk(K, D, 7))
    np.random.seed(8)
    x = mixture.propose(N)
    sw = np.random.uniform(0.5, 1.5, N) if weighted else None
    vb = GaussianInference(x, initial_guess=mixture, weights=sw)     # its constructor runs the E-step
    if weighted:
        sw = np.ascontiguousarray(vb.weights, dtype=np.float64)      # as the constructor normalised them (variational.pyx:94)
    m, W, nu, beta, ln_pi, ln_lam = vb_arrays(vb)
    s = upload(lib, ctx, x)
    Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
    r, lr = np.empty((N, K)), np.empty((N, K))
    rc = lib.pmc_vb_estep(ctx, s, dp(sw), K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), None,
                          dp(Nk), dp(xbar), dp(S), dp(elq), dp(r), dp(lr))
    assert rc == 0, lib.pmc_last_error()
    # against the oracle's two-pass loops (variational.pyx:699-932, :1003-1013)
    o = orc.vb_estep(x, sw, m, W, beta, nu, ln_pi, ln_lam)
    np.testing.assert_allclose(Nk, o["N_comp"], rtol=1e-10)
    np.testing.assert_allclose(xbar, o["x_mean_comp"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(S, o["S"], rtol=1e-9, atol=1e-11)
    assert abs(elq[0] - o["expectation_log_q_Z"]) <= 1e-9 * abs(o["expectation_log_q_Z"])
    assert np.max(np.abs(r - o["r"]) / o["r"]) < 1e-10
    # and the Python front-end, which runs the same kernels with the same shifts
    np.testing.assert_allclose(Nk, vb.N_comp, rtol=1e-12)
    np.testing.assert_allclose(xbar, vb.x_mean_comp, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(S, vb.S, rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(S, S.transpose(0, 2, 1))
    np.testing.assert_array_equal(r, vb.r)
    np.testing.assert_array_equal(lr, vb.log_rho)
    # without the N x K matrices (the E-step proper) the K-sized results are the same to rounding
    Nk2, xbar2, S2 = np.empty(K), np.empty((K, D)), np.empty((K, D, D))
    assert lib.pmc_vb_estep(ctx, s, dp(sw), K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), None,
                            dp(Nk2), dp(xbar2), dp(S2), None, None, None) == 0
    np.testing.assert_allclose(Nk2, Nk, rtol=1e-11)
    np.testing.assert_allclose(S2, S, rtol=1e-9, atol=1e-11)
    # moments about the previous x_mean_comp: same numbers to rounding
    assert lib.pmc_vb_estep(ctx, s, dp(sw), K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), dp(xbar),
                            dp(Nk2), dp(xbar2), dp(S2), None, None, None) == 0
    np.testing.assert_allclose(xbar2, xbar, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(S2, S, rtol=1e-9, atol=1e-11)
    lib.pmc_samples_free(s)


def test_vb_estep_far_start_values_take_the_second_pass(lib, ctx):
    """start means 300 sigma off: one-pass moments about m_k would lose 1e-11; the layer repeats them about the mean"""
    rs = np.random.RandomState(3)
    D, K, N = 4, 2, 5000
    x = np.ascontiguousarray(np.concatenate([rs.normal(0, 1, (N // 2, D)) + 300., rs.normal(0, 1, (N // 2, D)) - 300.]))
    m = np.ascontiguousarray(np.array([[1.] * D, [-1.] * D]))
    W = np.ascontiguousarray(np.array([np.eye(D) * 1e-2] * K))     # r = 1 / 1e-52: two clean blocks
    nu, beta = np.full(K, D + 1.), np.full(K, 1.)
    ln_pi, ln_lam = np.log(np.full(K, .5)), np.zeros(K)
    s = upload(lib, ctx, x)
    Nk, xbar, S = np.empty(K), np.empty((K, D)), np.empty((K, D, D))
    assert lib.pmc_vb_estep(ctx, s, None, K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), None,
                            dp(Nk), dp(xbar), dp(S), None, None, None) == 0, lib.pmc_last_error()
    for k, blk in enumerate((x[:N // 2], x[N // 2:])):
        np.testing.assert_allclose(Nk[k], N // 2, rtol=1e-12)
        np.testing.assert_allclose(xbar[k], blk.mean(axis=0), rtol=1e-13)
        c = blk - blk.mean(axis=0)
        np.testing.assert_allclose(S[k], c.T @ c / (N // 2), rtol=2e-12, atol=1e-15)
    lib.pmc_samples_free(s)


@pytest.mark.parametrize("D,K,N,rb,dead", [(5, 4, 3000, True, False), (20, 32, 40000, True, True),
                                           (8, 6, 9000, False, False)])
def test_gaussian_pmc_update_stats(lib, ctx, D, K, N, rb, dead):
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    mu, cov, w = mk(K, D, 11)
    if dead:
        w[2] = 0.
        w /= w.sum()
    proposal = create_gaussian_mixture(mu, cov, w)
    np.random.seed(12)
    x, origin = proposal.propose(N, trace=True, shuffle=False)
    wts = np.random.uniform(0.2, 2.0, N)
    latent = np.ascontiguousarray(origin, dtype=np.int64)
    ref = gaussian_pmc(x, proposal, wts, latent=None if rb else latent, rb=rb, copy=True)
    q, s = make_mix(lib, ctx, proposal), upload(lib, ctx, x)
    alpha, nmu, nsig = np.zeros(K), np.array(mu), np.zeros((K, D, D))
    ll, norm = np.empty(1), np.empty(1)
    rc = lib.pmc_pmc_update_stats(ctx, q, s, dp(wts), 0, None if rb else ip(latent), int(rb), dp(alpha), dp(nmu), dp(nsig),
                                  None, dp(ll), dp(norm))
    assert rc == 0, lib.pmc_last_error()
    assert abs(norm[0] - wts.sum()) <= 1e-12 * wts.sum()
    live = [k for k in range(K) if w[k] != 0]
    np.testing.assert_allclose(alpha[live] / alpha[live].sum(), ref.weights[live] / ref.weights[live].sum(), rtol=1e-11)
    for k in live:
        np.testing.assert_allclose(nmu[k], ref.components[k].mu, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(nsig[k], ref.components[k].sigma, rtol=1e-10, atol=1e-12)
    if dead:
        assert alpha[2] == 0. and not nsig[2].any()                 # rows of dead components are not written
    if rb:
        logq = proposal.multi_evaluate(x)
        assert abs(ll[0] - (wts * logq).sum()) <= 1e-10 * abs((wts * logq).sum())
    lib.pmc_mixture_destroy(q)
    lib.pmc_samples_free(s)


def test_student_t_pmc_update_stats_and_weights_on_device(lib, ctx):
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    from pypmc_amd.mix_adapt.pmc import student_t_pmc
    D, K, N = 6, 5, 20000
    mu, cov, w = mk(K, D, 21, spread=2.0)
    dof = np.array([3., 5., 8., 13., 21.])
    proposal = create_t_mixture(mu, cov, dof, w)
    target = create_gaussian_mixture(*mk(3, D, 22, spread=1.5))
    np.random.seed(23)
    x = proposal.propose(N)
    q, t, s = make_mix(lib, ctx, proposal), make_mix(lib, ctx, target), upload(lib, ctx, x)
    wts, sums = np.empty(N), np.empty(3)
    assert lib.pmc_is_weights(q, s, None, t, dp(wts), None, dp(sums)) == 0
    ref = student_t_pmc(x, proposal, wts, rb=True, copy=True)
    alpha, nmu, nsig, const = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D)), np.zeros(K)
    norm = np.empty(1)
    # the importance weights the weighting call left on the device: no N-sized array crosses the bus again
    rc = lib.pmc_pmc_update_stats(ctx, q, s, None, 1, None, 1, dp(alpha), dp(nmu), dp(nsig), dp(const), None, dp(norm))
    assert rc == 0, lib.pmc_last_error()
    assert abs(norm[0] - sums[0]) <= 1e-12 * sums[0]
    np.testing.assert_allclose(alpha, ref.weights, rtol=1e-10)
    for k in range(K):
        np.testing.assert_allclose(nmu[k], ref.components[k].mu, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(nsig[k], ref.components[k].sigma, rtol=1e-9, atol=1e-11)
        nu = brentq(lambda v: const[k] + np.log(.5 * v) - digamma(.5 * v), 1e-5, 1e3, maxiter=100)   # pmc.pyx:478-497, :696
        assert abs(nu - ref.components[k].dof) <= 1e-8 * ref.components[k].dof
    # host weights give the same update
    a2, m2, s2, c2 = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D)), np.zeros(K)
    assert lib.pmc_pmc_update_stats(ctx, q, s, dp(wts), 0, None, 1, dp(a2), dp(m2), dp(s2), dp(c2), None, None) == 0
    np.testing.assert_allclose(a2, alpha, rtol=1e-13)
    np.testing.assert_allclose(s2, nsig, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(c2, const, rtol=1e-11, atol=1e-13)
    assert lib.pmc_pmc_update_stats(ctx, q, s, dp(wts), 1, None, 1, dp(a2), dp(m2), dp(s2), dp(c2), None, None) < 0
    for h in (q, t):
        lib.pmc_mixture_destroy(h)
    lib.pmc_samples_free(s)


def test_generate_then_latent_update_and_mixture_update(lib, ctx):
    from pypmc_amd.density.mixture import create_gaussian_mixture
    D, K = 7, 4
    mu, cov, w = mk(K, D, 31)
    proposal = create_gaussian_mixture(mu, cov, w)
    counts = np.array([3000, 0, 1201, 4999], dtype=np.int64)        # rng.multinomial of the caller (mixture.pyx:192)
    N = int(counts.sum())
    q = make_mix(lib, ctx, proposal)
    chol = np.ascontiguousarray(np.linalg.cholesky(cov))
    s1, s2, s3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.pmc_samples_generate(ctx, q, dp(chol), ip(counts), 99, 0, C.byref(s1)) == 0, lib.pmc_last_error()
    assert lib.pmc_samples_generate(ctx, q, None, ip(counts), 99, 0, C.byref(s2)) == 0, lib.pmc_last_error()
    assert lib.pmc_samples_generate(ctx, q, dp(chol), ip(counts), 100, 0, C.byref(s3)) == 0
    assert lib.pmc_samples_count(s1) == N
    x1, x2, x3, origin = np.empty((N, D)), np.empty((N, D)), np.empty((N, D)), np.empty(N, dtype=np.int64)
    for h, out in ((s1, x1), (s2, x2), (s3, x3)):
        assert lib.pmc_samples_download(h, dp(out)) == 0
    assert lib.pmc_samples_origin(s1, ip(origin)) == 0
    np.testing.assert_array_equal(origin, np.repeat(np.arange(K), counts))     # bit-exact counts and origins
    np.testing.assert_allclose(x2, x1, rtol=1e-10, atol=1e-11)      # Cholesky factors derived from inv_sigma
    assert np.abs(x3 - x1).max() > 0.1                              # another seed, other numbers
    for k in (0, 2, 3):                                             # right distribution: whitened blocks ~ N(0, I)
        z = np.linalg.solve(chol[k], (x1[origin == k] - mu[k]).T).T
        assert np.abs(z.mean(axis=0)).max() < 5 / np.sqrt(counts[k]) and np.abs(np.cov(z.T) - np.eye(D)).max() < 0.15
    # non-Rao-Blackwell update with the origin the handle kept as latent: component k's block mean / covariance
    alpha, nmu, nsig = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D))
    assert lib.pmc_pmc_update_stats(ctx, q, s1, None, 0, None, 0, dp(alpha), dp(nmu), dp(nsig), None, None, None) == 0, \
        lib.pmc_last_error()
    np.testing.assert_allclose(alpha, counts / N, rtol=1e-12)
    for k in (0, 2, 3):
        blk = x1[origin == k]
        np.testing.assert_allclose(nmu[k], blk.mean(axis=0), rtol=1e-10, atol=1e-12)
        c = blk - blk.mean(axis=0)
        np.testing.assert_allclose(nsig[k], c.T @ c / counts[k], rtol=1e-9, atol=1e-11)
    # uploaded samples carry no origin
    up = upload(lib, ctx, x1)
    assert lib.pmc_samples_origin(up, ip(origin)) < 0
    assert lib.pmc_pmc_update_stats(ctx, q, up, None, 0, None, 0, dp(alpha), dp(nmu), dp(nsig), None, None, None) < 0
    assert b"`rb` must be True" in lib.pmc_last_error()             # the reference's message (pmc.pyx:81-83)
    # pmc_mixture_update: the handle follows the host's parameters
    out1, out2 = np.empty(N), np.empty(N)
    lib.pmc_mix_logpdf(q, up, dp(out1), None)
    newp = create_gaussian_mixture(mu + 0.5, cov * 1.3, w[::-1].copy())
    comps = newp.components
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in
            (newp.weights, [c.mu for c in comps], [c.inv_sigma for c in comps], [c.log_normalization for c in comps])]
    assert lib.pmc_mixture_update(q, dp(arrs[0]), dp(arrs[1]), dp(arrs[2]), dp(arrs[3]), None) == 0
    lib.pmc_mix_logpdf(q, up, dp(out2), None)
    np.testing.assert_array_equal(out2, newp.multi_evaluate(x1))
    assert np.abs(out2 - out1).max() > 1e-3
    # a precision matrix that does not factorise: the status and the component's number
    bad = arrs[2].copy()
    bad[1] = -np.eye(D)
    assert lib.pmc_mixture_update(q, dp(arrs[0]), dp(arrs[1]), dp(bad), dp(arrs[3]), None) == -2
    assert b"component 1" in lib.pmc_last_error()
    for h in (s1, s2, s3, up):
        lib.pmc_samples_free(h)
    lib.pmc_mixture_destroy(q)


def test_one_rank_communicator_changes_nothing(lib):
    """pmc_ctx_join with a world of one: every K-sized result goes through ncclAllReduce on the context's stream"""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    D, K, N = 10, 6, 15000
    mixture = create_gaussian_mixture(*mk(K, D, 41))
    np.random.seed(42)
    x = mixture.propose(N)
    wts = np.random.uniform(0.5, 1.5, N)
    results = []
    for joined in (False, True):
        ctx = C.c_void_p()
        assert lib.pmc_init(0, C.byref(ctx)) == 0
        if joined:
            uid = (C.c_char * 128)()
            assert lib.pmc_comm_unique_id(uid) == 0, lib.pmc_last_error()
            assert lib.pmc_ctx_join(ctx, 0, 1, uid) == 0, lib.pmc_last_error()
            assert lib.pmc_ctx_join(ctx, 0, 1, uid) < 0              # once
        q, s = make_mix(lib, ctx, mixture), upload(lib, ctx, x)
        alpha, nmu, nsig, ll = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D)), np.empty(1)
        assert lib.pmc_pmc_update_stats(ctx, q, s, dp(wts), 0, None, 1, dp(alpha), dp(nmu), dp(nsig), None, dp(ll), None) == 0, \
            lib.pmc_last_error()
        w, sums = np.empty(N), np.empty(3)
        assert lib.pmc_is_weights(q, s, dp(np.zeros(N)), None, dp(w), None, dp(sums)) == 0
        results.append((alpha, nmu, nsig, ll, sums))
        lib.pmc_mixture_destroy(q)
        lib.pmc_samples_free(s)
        assert lib.pmc_shutdown(ctx) == 0
    for a, b in zip(*results):
        np.testing.assert_array_equal(a, b)


def test_argument_errors(lib, ctx):
    h = C.c_void_p()
    assert lib.pmc_init(99, C.byref(h)) < 0 and not h.value
    assert lib.pmc_samples_upload(ctx, None, 5, 3, C.byref(h)) < 0
    assert lib.pmc_samples_upload(ctx, dp(np.zeros(3)), 1, 2000, C.byref(h)) < 0 and b"not supported" in lib.pmc_last_error()
    x = np.zeros((4, 3))
    s = upload(lib, ctx, x)
    from pypmc_amd.density.mixture import create_gaussian_mixture
    q = make_mix(lib, ctx, create_gaussian_mixture(*mk(2, 5, 1)))
    assert lib.pmc_mix_logpdf(q, s, dp(np.empty(4)), None) < 0 and b"do not belong together" in lib.pmc_last_error()
    assert lib.pmc_mixture_create(ctx, 2, 1, 3, dp(np.ones(1)), dp(np.zeros(3)), dp(np.eye(3)), dp(np.zeros(1)), None,
                                  C.byref(h)) < 0
    assert lib.pmc_mixture_create(ctx, 1, 1, 3, dp(np.ones(1)), dp(np.zeros(3)), dp(np.eye(3)), dp(np.zeros(1)), None,
                                  C.byref(h)) < 0 and b"h_dof" in lib.pmc_last_error()
    lib.pmc_mixture_destroy(q)
    lib.pmc_samples_free(s)


def test_out_of_memory_is_an_error_and_leaves_the_context_usable(lib, ctx):
    """a request beyond the device's memory: an error code and a message, no handle, no abort -- and no stale HIP error
    for the next launch to trip over (hipGetLastError behind a launch would report the failed hipMalloc)"""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    D, K = 3, 2
    mu, cov, w = mk(K, D, 77)
    mixture = create_gaussian_mixture(mu, cov, w)
    q = make_mix(lib, ctx, mixture)
    huge = np.array([2 ** 35, 2 ** 35], dtype=np.int64)             # 2^36 samples x 3 x 8 bytes = 1.6 TB
    h = C.c_void_p()
    assert lib.pmc_samples_generate(ctx, q, None, ip(huge), 1, 0, C.byref(h)) < 0 and not h.value
    assert b"hipMalloc" in lib.pmc_last_error() or b"memory" in lib.pmc_last_error().lower(), lib.pmc_last_error()
    x = np.random.RandomState(5).normal(size=(1000, D))
    s = upload(lib, ctx, x)
    out = np.empty(1000)
    assert lib.pmc_mix_logpdf(q, s, dp(out), None) == 0, lib.pmc_last_error()
    from oracle import oracle as orc
    comps = mixture.components
    ref, _ = orc.mixture_multi_evaluate(0, x, mixture.weights, mu, np.array([c.inv_sigma for c in comps]),
                                        np.array([c.log_normalization for c in comps]))
    np.testing.assert_allclose(out, ref, rtol=1e-10)
    counts = np.array([500, 500], dtype=np.int64)
    assert lib.pmc_samples_generate(ctx, q, None, ip(counts), 1, 0, C.byref(h)) == 0, lib.pmc_last_error()
    lib.pmc_samples_free(h)
    lib.pmc_samples_free(s)
    lib.pmc_mixture_destroy(q)


def test_many_contexts_in_sequence(lib):
    """300 contexts, each with a stream of its own (non-blocking), each running a finishing reduction as its very first
    launch: the library's per-stream scratch (at most 256 slots) is released with the context and re-used, and its
    zeroing is complete before the first launch on the new stream (a race between the NULL stream's memset and a
    non-blocking stream's kernel left the scalars unwritten now and then before)."""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    D, K, N = 6, 3, 4000
    mixture = create_gaussian_mixture(*mk(K, D, 51))
    np.random.seed(52)
    x = mixture.propose(N)
    lt = np.ascontiguousarray(np.random.normal(size=N))
    first = None
    for i in range(300):
        ctx = C.c_void_p()
        assert lib.pmc_init(0, C.byref(ctx)) == 0, lib.pmc_last_error()
        q, s = make_mix(lib, ctx, mixture), upload(lib, ctx, x)
        sums = np.empty(3)
        assert lib.pmc_is_weights(q, s, dp(lt), None, None, None, dp(sums)) == 0, (i, lib.pmc_last_error())
        if first is None:
            first = sums.copy()
            w = np.exp(lt - mixture.multi_evaluate(x))
            np.testing.assert_allclose(first, [w.sum(), (w * np.log(w)).sum(), (w * w).sum()], rtol=1e-10)
        np.testing.assert_array_equal(sums, first, err_msg="context %d" % i)
        lib.pmc_mixture_destroy(q)
        lib.pmc_samples_free(s)
        assert lib.pmc_shutdown(ctx) == 0


def test_first_launch_on_new_torch_streams(lib):
    """the kernel-level ABI on streams it has never seen (PyTorch's pool streams are non-blocking): the scalars of the
    first call on each are complete"""
    import torch
    from pypmc_amd.backend import HipBackend
    from pypmc_amd.density.mixture import create_gaussian_mixture, component_set
    be = HipBackend()
    mixture = create_gaussian_mixture(*mk(4, 9, 61))
    np.random.seed(62)
    x = be.asdevice(mixture.propose(3000))
    lt = be.asdevice(np.random.normal(size=3000))
    cs = component_set(mixture.components, mixture.weights)
    ref = be.tohost(be.logpdf(x, cs, log_target=lt, want_scalars=True)["scalars"])
    torch.cuda.synchronize()
    for _ in range(40):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            got = be.tohost(be.logpdf(x, cs, log_target=lt, want_scalars=True)["scalars"])
        np.testing.assert_array_equal(got, ref)
        assert lib.pmc_stream_release(C.c_void_p(st.cuda_stream)) == 0


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PMC_CTX_FUZZ_SEEDS", "16"))))   # (soak: more)
def test_handle_layer_fuzz(lib, ctx, seed):
    """random shapes through every handle-layer entry point against the Python front-end (same kernels underneath):
    all compiled dimensions incl. padded ones and the run-time-dimension unit, K = 1 ..., ragged N, both families,
    weights or none, a dead component now and then"""
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc, student_t_pmc
    from pypmc_amd.mix_adapt.variational import GaussianInference
    rs = np.random.RandomState(1000 + seed)
    D = int(rs.choice([1, 2, 3, 5, 7, 8, 9, 11, 13, 16, 19, 20, 23, 29, 31, 32, 37, 48, 61, 64, 70, 100]))
    K = int(rs.randint(1, 14))
    N = int(rs.choice([1, 63, 64, 65, 700, 2999, 6001]))
    student = bool(rs.randint(2))
    mu, cov, w = mk(K, D, 2000 + seed, spread=2.0)
    if K > 2 and rs.rand() < 0.4:
        w[rs.randint(K)] = 0.
        w /= w.sum()
    dof = rs.uniform(2.5, 30., K)
    mixture = create_t_mixture(mu, cov, dof, w) if student else create_gaussian_mixture(mu, cov, w)
    np.random.seed(seed)
    x = mixture.propose(N)
    wts = rs.uniform(0.1, 3.0, N) if rs.rand() < 0.7 else None
    q, s = make_mix(lib, ctx, mixture), upload(lib, ctx, x)
    tag = "seed %d: D=%d K=%d N=%d student=%d weighted=%d" % (seed, D, K, N, student, wts is not None)
    # log-pdf: bit-equal
    out, ind = np.empty(N), np.empty((N, K))
    assert lib.pmc_mix_logpdf(q, s, dp(out), dp(ind)) == 0, lib.pmc_last_error()
    ref_ind = np.empty((N, K))
    np.testing.assert_array_equal(out, mixture.multi_evaluate(x, individual=ref_ind), err_msg=tag)
    np.testing.assert_array_equal(ind, ref_ind, err_msg=tag)
    # PMC update
    live = [k for k in range(K) if w[k] != 0]
    alpha, nmu, nsig, const = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D)), np.zeros(K)
    rc = lib.pmc_pmc_update_stats(ctx, q, s, dp(wts), 0, None, 1, dp(alpha), dp(nmu), dp(nsig), dp(const) if student else None,
                                  None, None)
    assert rc == 0, (tag, lib.pmc_last_error())
    norm = wts.sum() if wts is not None else float(N)
    from pypmc_amd.backend import get_backend
    from pypmc_amd.density.mixture import component_set
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    be = get_backend(None)
    cs = component_set(mixture.components, mixture.weights, live, K)
    st = split_stats(be.tohost(be.estep(x, cs, 1, max_init_zero=len(live) < K, sample_w=wts)["stats"]), len(live), D)
    shift = mu[live]
    if student:
        m_ref, c_ref = centred_moments(st[1], st[2], st[3], shift, S0_cov=st[4])
        a_ref = st[4] / norm
    else:
        m_ref, c_ref = centred_moments(st[1], st[2], st[3], shift)
        a_ref = st[1] / norm
    from pypmc_amd.mix_adapt._stats import shift_is_far
    if not shift_is_far(st[1], st[2], st[3]):                      # (else the layer took its second pass: checked elsewhere)
        np.testing.assert_allclose(alpha[live], a_ref, rtol=1e-11, atol=1e-300, err_msg=tag)
        np.testing.assert_allclose(nmu[live], m_ref, rtol=1e-10, atol=1e-12, err_msg=tag)
        scale = np.abs(c_ref).max(axis=(1, 2))[:, None, None] + 1e-300
        assert np.max(np.abs(nsig[live] - c_ref) / scale) < 1e-9, tag
    # VB E-step (Gaussian posterior): against GaussianInference with the mixture as its guess
    if not student and N >= K and len(live) == K:
        vb = GaussianInference(x, initial_guess=mixture, weights=wts)
        sw = np.ascontiguousarray(vb.weights) if wts is not None else None
        m, W, nu, beta, ln_pi, ln_lam = vb_arrays(vb)
        Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
        assert lib.pmc_vb_estep(ctx, s, dp(sw), K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), None,
                                dp(Nk), dp(xbar), dp(S), dp(elq), None, None) == 0, (tag, lib.pmc_last_error())
        np.testing.assert_allclose(Nk, vb.N_comp, rtol=1e-11, err_msg=tag)
        np.testing.assert_allclose(xbar, vb.x_mean_comp, rtol=1e-10, atol=1e-12, err_msg=tag)
        scale = np.abs(vb.S).max(axis=(1, 2))[:, None, None] + 1e-300
        assert np.max(np.abs(S - vb.S) / scale) < 1e-9, tag
        assert abs(elq[0] - vb._expectation_log_q_Z) <= 1e-10 * abs(vb._expectation_log_q_Z) + 1e-12, tag
    lib.pmc_mixture_destroy(q)
    lib.pmc_samples_free(s)


def test_vb_estep_large_batch_takes_the_fast_forms(lib, ctx):
    """N large enough for the grouped responsibilities and the common-shift statistics (k_resp_groups + k_stats_gemm):
    the handle layer issues the very call the front-end issues -- same numbers"""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.variational import GaussianInference
    D, K, N = 20, 32, 600_000
    mixture = create_gaussian_mixture(*mk(K, D, 71))
    np.random.seed(72)
    x = mixture.propose(N)
    vb = GaussianInference(x, initial_guess=mixture)
    m, W, nu, beta, ln_pi, ln_lam = vb_arrays(vb)
    s = upload(lib, ctx, x)
    Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
    assert lib.pmc_vb_estep(ctx, s, None, K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), None,
                            dp(Nk), dp(xbar), dp(S), dp(elq), None, None) == 0, lib.pmc_last_error()
    np.testing.assert_array_equal(Nk, vb.N_comp)
    np.testing.assert_array_equal(xbar, vb.x_mean_comp)
    np.testing.assert_allclose(S, vb.S, rtol=1e-13, atol=1e-15)        # (numpy's einsum against the plain loop)
    assert elq[0] == vb._expectation_log_q_Z
    assert abs(Nk.sum() - N) < 1e-6
    lib.pmc_samples_free(s)


@pytest.mark.parametrize("D,N,how", [(2, 1000, "host"), (20, 70001, "device"), (70, 333, "none"), (5, 1, "host")])
def test_weighted_moments(lib, ctx, D, N, how):
    """calculate_mean / calculate_covariance (importance_sampling.py:46-83) through the handle layer: against the
    reference's formulas in numpy and against the front-end's functions"""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import calculate_mean, calculate_covariance
    mixture = create_gaussian_mixture(*mk(3, D, 81))
    np.random.seed(82)
    x = mixture.propose(N) + 40.                                     # far from the origin: the shift matters
    s = upload(lib, ctx, x)
    if how == "host":
        w = np.random.uniform(0.1, 2.0, N)
        args = (dp(w), 0)
    elif how == "device":                                            # the importance weights a weighting call left behind
        q = make_mix(lib, ctx, mixture)
        w = np.empty(N)
        lt = np.ascontiguousarray(mixture.multi_evaluate(x) + np.random.normal(size=N))     # weights = exp(N(0, 1))
        assert lib.pmc_is_weights(q, s, dp(lt), None, dp(w), None, None) == 0, lib.pmc_last_error()
        lib.pmc_mixture_destroy(q)
        args = (None, 1)
    else:
        w = np.ones(N)
        args = (None, 0)
    mean, cov = np.empty(D), np.empty((D, D))
    assert lib.pmc_weighted_moments(ctx, s, args[0], args[1], dp(mean), dp(cov)) == 0, lib.pmc_last_error()
    ref_mean = (w[:, None] * x).sum(axis=0) / w.sum()                # importance_sampling.py:58-61
    np.testing.assert_allclose(mean, ref_mean, rtol=1e-12)
    if N > 1:
        c = x - ref_mean
        sw, q2 = w.sum(), (w * w).sum()
        ref_cov = sw / (sw * sw - q2) * np.einsum('n,ni,nj->ij', w, c, c)      # :76-83
        np.testing.assert_allclose(cov, ref_cov, rtol=1e-9, atol=1e-12 * np.abs(ref_cov).max())
        np.testing.assert_allclose(mean, calculate_mean(x, w), rtol=1e-13)
        np.testing.assert_allclose(cov, calculate_covariance(x, w), rtol=1e-10, atol=1e-13 * np.abs(ref_cov).max())
    assert lib.pmc_weighted_moments(ctx, s, dp(w), 1, dp(mean), dp(cov)) < 0
    lib.pmc_samples_free(s)


def test_two_contexts_two_threads_with_their_own_options(lib):
    """re-entrant per context, serialised per context.  Two contexts in two threads run the same large E-step
    at once, one with the common-shift statistics switched off for ITS context (pmc_ctx_configure), the other with the
    defaults; a third thread hammers the process-wide pmc_configure meanwhile.  Each context's numbers are, bit for bit,
    what it gives alone -- and the two differ (different kernels), so a shared global would show."""
    import threading
    from pypmc_amd import _lib
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.variational import GaussianInference
    D, K, N = 20, 32, 300_000
    mixture = create_gaussian_mixture(*mk(K, D, 81))
    np.random.seed(82)
    x = mixture.propose(N)
    vb = GaussianInference(x[:2000], initial_guess=mixture)
    m, W, nu, beta, ln_pi, ln_lam = vb_arrays(vb)

    def make(limit):
        h = C.c_void_p()
        assert lib.pmc_init(0, C.byref(h)) == 0, lib.pmc_last_error()
        assert lib.pmc_ctx_configure(h, b"stats_common_shift_min_n", 0.0) == 0
        if limit is not None:
            assert lib.pmc_ctx_configure(h, b"stats_common_shift_limit", limit) == 0
        assert lib.pmc_ctx_configure(h, b"no_such_option", 1.0) < 0
        assert lib.pmc_ctx_timing_enable(h, 1) == 0
        return h, upload(lib, h, x)

    def estep(h, s):
        Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
        rc = lib.pmc_vb_estep(h, s, None, K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), None,
                              dp(Nk), dp(xbar), dp(S), dp(elq), None, None)
        assert rc == 0, lib.pmc_last_error()
        return np.concatenate([Nk, xbar.ravel(), S.ravel(), elq])

    (ha, sa), (hb, sb) = make(0.0), make(None)
    alone_a, alone_b = estep(ha, sa), estep(hb, sb)
    assert not np.array_equal(alone_a, alone_b), "the two contexts should run different statistics kernels"
    np.testing.assert_allclose(alone_a, alone_b, rtol=1e-9, atol=1e-9)
    errors, stop = [], threading.Event()

    def worker(h, s, ref):
        try:
            for _ in range(12):
                if not np.array_equal(estep(h, s), ref):
                    errors.append("a context's result changed under concurrency")
        except Exception as exc:                           # noqa: BLE001
            errors.append(repr(exc))

    def meddler():
        flip = 0
        while not stop.is_set():
            lib.pmc_configure(b"stats_common_shift_limit", 0.0 if flip else 1000.0)
            flip ^= 1
    threads = [threading.Thread(target=worker, args=(ha, sa, alone_a)), threading.Thread(target=worker, args=(hb, sb, alone_b)),
               threading.Thread(target=worker, args=(hb, sb, alone_b))]       # (two threads on ONE context: serialised inside)
    med = threading.Thread(target=meddler)
    med.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    med.join()
    lib.pmc_configure(b"stats_common_shift_limit", 1000.0)
    assert not errors, errors
    # each context's own timing record: context a ran 13 E-steps, context b 25; the process-wide record saw none of them
    buf = (_lib.Timing * 16)()
    n = C.c_int(0)
    calls = {}
    for name, h in (("a", ha), ("b", hb)):
        assert lib.pmc_ctx_get_timings(h, C.cast(buf, C.c_void_p), 16, C.byref(n)) == 0
        calls[name] = {buf[i].name.decode(): buf[i].calls for i in range(n.value)}
    assert calls["a"]["k_resp"] == 13 and calls["b"]["k_resp"] == 25, calls
    assert calls["a"]["k_stats"] == 13 and calls["b"]["k_stats"] == 25, calls
    assert lib.pmc_get_timings(C.cast(buf, C.c_void_p), 16, C.byref(n)) == 0 and n.value == 0
    for h, s in ((ha, sa), (hb, sb)):
        lib.pmc_samples_free(s)
        assert lib.pmc_shutdown(h) == 0
