"""SURVEY 8(b) row 1 / verdict r4 #1: ONE context over several devices for one host process (pmc_init_devices) -- lib-owned
contiguous shards (pmc_samples_upload / _generate), every N-sized call on all devices at once (one host thread per
device), the K-sized vectors added in device order.  A one-GPU box runs it with the same ordinal repeated ("virtual
shards": own streams, own scratch, own threads), which is what these tests do:

  * bit-equal to the device-ordered sum ((v_0 + v_1) + v_2) + ... of the per-shard results of the kernel level,
  * within 1e-10 of the oracle (the restated reference loops),
  * the golden vectors generated from the reference through it (tests/test_gpu_ctx_golden.py runs every case through a
    context of three parts as well),
  * N-sized outputs (log q, weights, r, generated samples, origins) bit-equal to a one-device context's.
"""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.special import digamma

from test_gpu_ctx import dp, ip, mk, make_mix

pytestmark = pytest.mark.gpu
LAYOUTS = [[0, 0], [0, 0, 0, 0], [0, 0, 0]]


@pytest.fixture(scope="module")
def lib():
    from pypmc_amd import _lib
    return _lib.load()


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    return HipBackend()


def open_ctx(lib, ids):
    h = C.c_void_p()
    arr = (C.c_int * len(ids))(*ids)
    assert lib.pmc_init_devices(len(ids), arr, C.byref(h)) == 0, lib.pmc_last_error()
    assert lib.pmc_ctx_device_count(h) == len(ids)
    return h


@pytest.fixture(params=LAYOUTS, ids=lambda l: "x".join(map(str, l)))
def mctx(lib, request):
    h = open_ctx(lib, request.param)
    yield h, len(request.param)
    assert lib.pmc_shutdown(h) == 0


@pytest.fixture()
def one(lib):
    h = C.c_void_p()
    assert lib.pmc_init(0, C.byref(h)) == 0
    yield h
    assert lib.pmc_shutdown(h) == 0


def upload(lib, ctx, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = C.c_void_p()
    assert lib.pmc_samples_upload(ctx, dp(x), x.shape[0], x.shape[1], C.byref(h)) == 0, lib.pmc_last_error()
    return h


def shards(lib, s, n):
    out = []
    for p in range(n):
        b, c = C.c_int64(), C.c_int64()
        assert lib.pmc_samples_shard(s, p, C.byref(b), C.byref(c)) == 0          # (the device ordinal)
        out.append((b.value, b.value + c.value))
    return out


def ordered_sum(vs):
    s = vs[0].copy()
    for v in vs[1:]:
        s = s + v
    return s


def vb_problem(K, D, N, seed, weighted):
    rs = np.random.RandomState(seed)
    mu, cov, w = mk(K, D, seed)
    comp = rs.choice(K, N, p=w)
    x = mu[comp] + np.einsum('nij,nj->ni', np.linalg.cholesky(cov)[comp], rs.normal(size=(N, D)))
    nu, beta, alpha = D + 2. + rs.uniform(0, 3, K), 1. + rs.uniform(0, 3, K), 1. + rs.uniform(0, 3, K)
    W = np.linalg.inv(cov) / nu[:, None, None]
    W = np.ascontiguousarray(0.5 * (W + W.transpose(0, 2, 1)))
    ln_lam = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + np.linalg.slogdet(W)[1]
    ln_pi = digamma(alpha) - digamma(alpha.sum())
    sw = rs.uniform(0.5, 1.5, N) if weighted else None
    return np.ascontiguousarray(x), sw, np.ascontiguousarray(mu), W, nu, beta, ln_pi, ln_lam


def run_vb(lib, ctx, s, sw, K, D, m, W, nu, beta, ln_pi, ln_lam, shift=None, want_nk=False, N=0):
    Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
    r = np.empty((N, K)) if want_nk else None
    lr = np.empty((N, K)) if want_nk else None
    rc = lib.pmc_vb_estep(ctx, s, dp(sw), K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), dp(shift), dp(Nk), dp(xbar),
                          dp(S), dp(elq), dp(r), dp(lr))
    assert rc == 0, lib.pmc_last_error()
    return Nk, xbar, S, elq[0], r, lr


@pytest.mark.parametrize("D,K,N,weighted", [(3, 4, 2000, False), (20, 32, 70001, True), (5, 8, 30011, False), (40, 64, 9000, True),
                                            (20, 32, 1200000, False), (2, 3, 5, False)])
def test_vb_estep_is_the_device_ordered_sum_of_its_shards(lib, be, mctx, D, K, N, weighted):
    """pmc_vb_estep over virtual shards: N_k / x-bar / S / E[log q(Z)] bit-equal to the kernel level run per shard
    (pypmc_amd's backend: the same entry point, pmc_estep_about), the raw vectors added in shard order, converted once
    (pmc_host_convert_stats); and within 1e-10 of the oracle.  N = 1.2e6 puts every shard on the large-batch forms
    (k_resp_groups + k_stats_gemm); N = 5 leaves parts without a sample."""
    from oracle import oracle as orc
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd._lib import PMC_KIND_VB, PMC_RESP_VB
    from pypmc_amd.mix_adapt._stats import convert_stats, regularize
    ctx, n = mctx
    x, sw, m, W, nu, beta, ln_pi, ln_lam = vb_problem(K, D, N, 7 + D, weighted)
    s = upload(lib, ctx, x)
    parts = shards(lib, s, n)
    assert parts[0][0] == 0 and parts[-1][1] == N and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert max(e - b for b, e in parts) - min(e - b for b, e in parts) <= 1
    Nk, xbar, S, elq, _, _ = run_vb(lib, ctx, s, sw, K, D, m, W, nu, beta, ln_pi, ln_lam)
    cs = ComponentSet(PMC_KIND_VB, m, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lam - D * np.log(2. * np.pi))
    flats = []
    for b, e in parts:
        if e > b:
            flats.append(be.tohost(be.estep(x[b:e], cs, PMC_RESP_VB, sample_w=None if sw is None else sw[b:e])["stats"]))
        else:
            flats.append(np.zeros(be.stats_len(K, D)))
    flat = ordered_sum(flats)
    scalars, S0, M1, x_mean, S_ref, far, _, _ = convert_stats(flat, K, D, m)
    assert not far
    np.testing.assert_array_equal(Nk, regularize(S0.copy()))
    np.testing.assert_array_equal(xbar, x_mean)
    np.testing.assert_array_equal(S, S_ref)
    assert elq == scalars[0]
    if N <= 100000:
        o = orc.vb_estep(x, sw, m, W, beta, nu, ln_pi, ln_lam)
        np.testing.assert_allclose(Nk, o["N_comp"], rtol=1e-10)
        sd = np.sqrt(np.maximum(np.einsum('kii->ki', o["S"]), 1e-300))
        big = o["N_comp"] > 1e-6 * N
        assert np.max(np.abs(xbar - o["x_mean_comp"])[big] / sd[big]) < 1e-10
        assert np.max((np.abs(S - o["S"]) / (sd[:, :, None] * sd[:, None, :]))[big]) < 1e-10
        assert abs(elq - o["expectation_log_q_Z"]) <= 1e-10 * abs(o["expectation_log_q_Z"]) + 1e-14 * N
    lib.pmc_samples_free(s)


def test_vb_estep_rows_and_shift_and_second_pass(lib, mctx, one):
    """r / log_rho rows of every shard land in the caller's N x K arrays exactly as a one-device context writes them; a
    caller's shift and the far-shift second pass run on all parts"""
    ctx, n = mctx
    D, K, N = 6, 5, 20003
    x, sw, m, W, nu, beta, ln_pi, ln_lam = vb_problem(K, D, N, 3, True)
    s, s1 = upload(lib, ctx, x), upload(lib, one, x)
    got = run_vb(lib, ctx, s, sw, K, D, m, W, nu, beta, ln_pi, ln_lam, want_nk=True, N=N)
    ref = run_vb(lib, one, s1, sw, K, D, m, W, nu, beta, ln_pi, ln_lam, want_nk=True, N=N)
    np.testing.assert_array_equal(got[4], ref[4])
    np.testing.assert_array_equal(got[5], ref[5])
    for a, b in zip(got[:4], ref[:4]):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)           # (another summation order over the samples)
    shift = np.ascontiguousarray(ref[1])
    g2 = run_vb(lib, ctx, s, sw, K, D, m, W, nu, beta, ln_pi, ln_lam, shift=shift)
    r2 = run_vb(lib, one, s1, sw, K, D, m, W, nu, beta, ln_pi, ln_lam, shift=shift)
    for a, b in zip(g2[:4], r2[:4]):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)
    lib.pmc_samples_free(s)
    lib.pmc_samples_free(s1)
    # start means 300 sigma off: the second pass about the mean just found (decided on the summed vector)
    rs = np.random.RandomState(3)
    D, K, N = 4, 2, 5001
    h = N // 2
    x = np.ascontiguousarray(np.concatenate([rs.normal(0, 1, (h, D)) + 300., rs.normal(0, 1, (N - h, D)) - 300.]))
    m = np.ascontiguousarray(np.array([[1.] * D, [-1.] * D]))
    W = np.ascontiguousarray(np.array([np.eye(D) * 1e-2] * K))
    s = upload(lib, ctx, x)
    Nk, xbar, S, _, _, _ = run_vb(lib, ctx, s, None, K, D, m, W, np.full(K, D + 1.), np.full(K, 1.), np.log(np.full(K, .5)), np.zeros(K))
    for k, blk in enumerate((x[:h], x[h:])):
        np.testing.assert_allclose(Nk[k], len(blk), rtol=1e-12)
        np.testing.assert_allclose(xbar[k], blk.mean(axis=0), rtol=1e-13)
        c = blk - blk.mean(axis=0)
        np.testing.assert_allclose(S[k], c.T @ c / len(blk), rtol=2e-12, atol=1e-15)
    lib.pmc_samples_free(s)


@pytest.mark.parametrize("student,D,K,N", [(False, 5, 3, 1001), (False, 20, 32, 70001), (True, 30, 8, 5003), (True, 70, 4, 777),
                                           (False, 40, 64, 70000)])
def test_logpdf_and_importance_weights(lib, mctx, one, student, D, K, N):
    """N-sized outputs are per-sample: the one-device context's numbers (same kernels, same pack: bit-equal where the parts
    run the same form, to rounding otherwise); the three sums of the weighting pass are the ordered sum of the parts' sums;
    everything within 1e-10 of the oracle"""
    from oracle import oracle as orc
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    ctx, n = mctx
    mu, cov, w = mk(K, D, 3)
    mixture = create_t_mixture(mu, cov, np.full(K, 6.5), w) if student else create_gaussian_mixture(mu, cov, w)
    target = create_gaussian_mixture(*mk(3, D, 5, spread=1.0))
    np.random.seed(4)
    x = mixture.propose(N)
    outs = []
    for c in (ctx, one):
        q, t, s = make_mix(lib, c, mixture), make_mix(lib, c, target), upload(lib, c, x)
        out, ind = np.empty(N), np.empty((N, K))
        assert lib.pmc_mix_logpdf(q, s, dp(out), dp(ind)) == 0, lib.pmc_last_error()
        sub = np.full((N, K), -7.0)
        comps = np.array([K - 1, 0], dtype=np.int32)
        assert lib.pmc_mix_logpdf_components(q, s, comps.ctypes.data_as(C.POINTER(C.c_int32)), 2, dp(sub)) == 0
        wts, lt, sums = np.empty(N), np.empty(N), np.empty(3)
        assert lib.pmc_is_weights(q, s, None, t, dp(wts), dp(lt), dp(sums)) == 0, lib.pmc_last_error()
        w2, sums2 = np.empty(N), np.empty(3)
        assert lib.pmc_is_weights(q, s, dp(lt), None, dp(w2), None, dp(sums2)) == 0
        back = np.empty_like(x)
        assert lib.pmc_samples_download(s, dp(back)) == 0
        np.testing.assert_array_equal(back, x)
        mean, cv = np.empty(D), np.empty((D, D))
        assert lib.pmc_weighted_moments(c, s, None, 1, dp(mean), dp(cv)) == 0, lib.pmc_last_error()
        outs.append((out, ind, sub, wts, lt, sums, w2, sums2, mean, cv))
        for h in (q, t):
            lib.pmc_mixture_destroy(h)
        lib.pmc_samples_free(s)
    got, ref = outs
    # Which form a device runs depends on ITS batch: the matrix-product form of the Mahalanobis forms engages from 49152 samples
    # on (D = 40: the whole batch takes it, a shard does not: 1e-11, not bit for bit), and below split_max_rounds rounds of the
    # chip the components of a sample block are walked in pieces whose number follows from the block count (round 6: log q of a
    # block in pieces agrees with the one-workgroup walk to the rounding of the merge, a few ulps) -- the batch-size dependence
    # include/pmc_hip.h documents.  Per-pair outputs (the individual matrix) do not depend on the pieces.
    threshold_case = D == 40 and N // n < 49152 <= N
    for i in (0, 1, 2, 3, 4, 6):
        if threshold_case and i in (0, 1, 3, 6):             # (log q, the individual matrix, the weights twice)
            np.testing.assert_allclose(got[i], ref[i], rtol=1e-10)
        elif i in (0, 3, 4, 6) and n > 1:                    # (log q, weights, log P: the rounding of the merge)
            np.testing.assert_allclose(got[i], ref[i], rtol=1e-13, atol=1e-14)
        else:
            np.testing.assert_array_equal(got[i], ref[i])
    assert (got[2][:, 1:K - 1] == -7.0).all()
    np.testing.assert_allclose(got[5], ref[5], rtol=1e-13)
    np.testing.assert_allclose(got[7], got[5], rtol=1e-13)
    np.testing.assert_allclose(got[8], ref[8], rtol=1e-12, atol=1e-13)
    # (moments about the first sample, several standard deviations from the mean: both carry ~1e-13 of M2 / sum w)
    np.testing.assert_allclose(got[9], ref[9], rtol=1e-10, atol=1e-12 * float(np.abs(ref[9]).max()) * 100)
    comps = mixture.components
    inv = np.array([c.inv_sigma for c in comps])
    ln = np.array([c.log_normalization for c in comps])
    sl = slice(0, N, max(N // 4000, 1))
    if student:
        o, _ = orc.mixture_multi_evaluate(1, x[sl], mixture.weights, mu, inv, ln, prefactor=np.full(K, -.5 * (6.5 + D)),
                                          inv_dof=np.full(K, 1. / 6.5))
    else:
        o, _ = orc.mixture_multi_evaluate(0, x[sl], mixture.weights, mu, inv, ln)
    assert np.max(np.abs(got[0][sl] - o) / np.abs(o)) < 1e-10
    wn = got[3]
    np.testing.assert_allclose(got[5], [wn.sum(), (wn[wn > 0] * np.log(wn[wn > 0])).sum(), (wn ** 2).sum()], rtol=1e-10)


@pytest.mark.parametrize("student", [False, True])
def test_generate_is_the_one_device_stream_split_by_rows(lib, mctx, one, student):
    """pmc_samples_generate on several parts: counts and origins bit-exact (the caller's multinomial counts clipped to the
    parts' row ranges), the samples those one device draws for the same rows (Philox counted by the global row)"""
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    ctx, n = mctx
    D, K = 7, 5
    mu, cov, w = mk(K, D, 31)
    proposal = create_t_mixture(mu, cov, np.array([3., 4., 5., 6., 7.]), w) if student else create_gaussian_mixture(mu, cov, w)
    counts = np.array([3000, 0, 1201, 4999, 3], dtype=np.int64)
    N = int(counts.sum())
    chol = np.ascontiguousarray(np.linalg.cholesky(cov))
    res = []
    for c in (ctx, one):
        q, s = make_mix(lib, c, proposal), C.c_void_p()
        assert lib.pmc_samples_generate(c, q, dp(chol), ip(counts), 99, 1000, C.byref(s)) == 0, lib.pmc_last_error()
        assert lib.pmc_samples_count(s) == N
        x, origin = np.empty((N, D)), np.empty(N, dtype=np.int64)
        assert lib.pmc_samples_download(s, dp(x)) == 0 and lib.pmc_samples_origin(s, ip(origin)) == 0
        # the latent (non-Rao-Blackwellised) update with the origins the handle kept
        alpha, nmu, nsig, dc = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D)), np.zeros(K)
        assert lib.pmc_pmc_update_stats(c, q, s, None, 0, None, 0, dp(alpha), dp(nmu), dp(nsig), dp(dc) if student else None,
                                        None, None) == 0, lib.pmc_last_error()
        res.append((x, origin, alpha, nmu, nsig))
        lib.pmc_mixture_destroy(q)
        lib.pmc_samples_free(s)
    np.testing.assert_array_equal(res[0][1], np.repeat(np.arange(K), counts))
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-13)
    np.testing.assert_allclose(res[0][3], res[1][3], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(res[0][4], res[1][4], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("student,D,K,N,dead", [(False, 5, 4, 3001, False), (False, 20, 32, 40000, True), (True, 6, 5, 20000, False),
                                                (False, 40, 128, 66000, False)])
def test_pmc_update_stats_against_the_front_end_and_the_ordered_sum(lib, be, mctx, student, D, K, N, dead):
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc, student_t_pmc
    ctx, n = mctx
    mu, cov, w = mk(K, D, 11, spread=2.0 if student else 3.0)
    if dead:
        w[2] = 0.
        w /= w.sum()
    dof = 3. + np.arange(K) % 7
    proposal = create_t_mixture(mu, cov, dof, w) if student else create_gaussian_mixture(mu, cov, w)
    np.random.seed(12)
    x = proposal.propose(N)
    wts = np.random.uniform(0.2, 2.0, N)
    ref = (student_t_pmc if student else gaussian_pmc)(x, proposal, wts, rb=True, copy=True)
    q, s = make_mix(lib, ctx, proposal), upload(lib, ctx, x)
    alpha, nmu, nsig, dc = np.zeros(K), np.array(mu), np.zeros((K, D, D)), np.zeros(K)
    ll, norm = np.empty(1), np.empty(1)
    rc = lib.pmc_pmc_update_stats(ctx, q, s, dp(wts), 0, None, 1, dp(alpha), dp(nmu), dp(nsig), dp(dc) if student else None,
                                  dp(ll), dp(norm))
    assert rc == 0, lib.pmc_last_error()
    parts = shards(lib, s, n)
    assert norm[0] == ordered_sum([np.array(float(np.sum(wts[b:e].astype(np.longdouble)))) for b, e in parts])
    live = [k for k in range(K) if w[k] != 0]
    np.testing.assert_allclose(alpha[live] / alpha[live].sum(), ref.weights[live] / ref.weights[live].sum(), rtol=1e-10)
    for k in live:
        np.testing.assert_allclose(nmu[k], ref.components[k].mu, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(nsig[k], ref.components[k].sigma, rtol=1e-9, atol=1e-11)
    logq = proposal.multi_evaluate(x)
    assert abs(ll[0] - (wts * logq).sum()) <= 1e-10 * abs((wts * logq).sum())
    lib.pmc_mixture_destroy(q)
    lib.pmc_samples_free(s)


def test_errors_timings_options_and_threads(lib):
    """a failing part is the call's failure (status + message naming the device); timings merge by kernel name; options are
    the context's; two multi-device contexts run side by side from two threads"""
    import threading
    from pypmc_amd import _lib
    bad = (C.c_int * 2)(0, 99)
    h = C.c_void_p()
    assert lib.pmc_init_devices(2, bad, C.byref(h)) == _lib.PMC_ENODEVICE and not h.value
    ctx = open_ctx(lib, [0, 0, 0])
    ids = (C.c_int * 8)()
    assert lib.pmc_ctx_devices(ctx, ids, 8) == 3 and list(ids)[:3] == [0, 0, 0]
    D, K, N = 5, 3, 9000
    mu, cov, w = mk(K, D, 2)
    prec = np.ascontiguousarray(np.linalg.inv(cov))
    notpd = prec.copy()
    notpd[1] = -np.eye(D)
    ln = np.zeros(K)
    m = C.c_void_p()
    assert lib.pmc_mixture_create(ctx, 0, K, D, dp(w), dp(mu), dp(notpd), dp(ln), None, C.byref(m)) == _lib.PMC_ENOTPOSDEF
    assert lib.pmc_mixture_create(ctx, 0, K, D, dp(w), dp(mu), dp(prec), dp(ln), None, C.byref(m)) == 0
    x = np.ascontiguousarray(np.random.RandomState(0).normal(size=(N, D)))
    s = upload(lib, ctx, x)
    assert lib.pmc_ctx_timing_enable(ctx, 1) == 0
    out = np.empty(N)
    assert lib.pmc_mix_logpdf(m, s, dp(out), None) == 0
    buf = (_lib.Timing * 16)()
    nt = C.c_int(0)
    assert lib.pmc_ctx_get_timings(ctx, C.cast(buf, C.c_void_p), 16, C.byref(nt)) == 0
    assert nt.value == 1 and buf[0].name == b"k_logpdf" and buf[0].calls == 3 and buf[0].ms > 0
    assert lib.pmc_ctx_timing_enable(ctx, 0) == 0
    assert lib.pmc_ctx_configure(ctx, b"maha_gemm_min_n", 1000.) == 0 and lib.pmc_ctx_configure(ctx, b"nope", 1.) < 0
    # a NaN target value reaches the weights of its row only
    lt = np.zeros(N)
    lt[N - 1] = np.nan
    wts, sums = np.empty(N), np.empty(3)
    assert lib.pmc_is_weights(m, s, dp(lt), None, dp(wts), None, dp(sums)) == 0
    assert np.isnan(wts[N - 1]) and np.isfinite(wts[:N - 1]).all()
    lib.pmc_samples_free(s)
    lib.pmc_mixture_destroy(m)
    assert lib.pmc_shutdown(ctx) == 0
    # two contexts of two parts each, driven from two threads
    results = {}

    def work(tag, seed):
        c = open_ctx(lib, [0, 0])
        xx, sw, mm, W, nu, beta, ln_pi, ln_lam = vb_problem(4, 6, 30000, seed, False)
        ss = upload(lib, c, xx)
        outs = [run_vb(lib, c, ss, None, 4, 6, mm, W, nu, beta, ln_pi, ln_lam)[:4] for _ in range(5)]
        lib.pmc_samples_free(ss)
        assert lib.pmc_shutdown(c) == 0
        results[tag] = outs
    th = [threading.Thread(target=work, args=(i, 40 + i)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for tag in (0, 1):
        first = results[tag][0]
        for o in results[tag][1:]:
            for a, b in zip(first, o):
                np.testing.assert_array_equal(a, b)                # bit-reproducible from call to call


@pytest.mark.parametrize("ids", [[0], [0, 0, 0]], ids=["one_device", "three_parts"])
def test_more_than_2_to_31_samples_through_the_handle_layer(lib, ids):
    """2^31 + 5 one-dimensional samples generated on the device(s): sample counts, shard boundaries and origins are 64-bit
    all the way through the handle layer -- closed forms of the K-sized results, nothing N-sized on the host"""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.variational import GaussianInference
    h = open_ctx(lib, ids)
    K, D = 2, 1
    N = 2 ** 31 + 5
    mu = np.array([[-2.0], [1.0]])
    var = np.array([0.25, 2.25])
    w = np.array([0.25, 0.75])                                     # = counts / N: E[rho_k] = w_k below
    mixture = create_gaussian_mixture(mu, var.reshape(K, 1, 1), w)
    q = make_mix(lib, h, mixture)
    counts = np.array([N // 4, N - N // 4], dtype=np.int64)
    s = C.c_void_p()
    assert lib.pmc_samples_generate(h, q, None, ip(counts), 11, 0, C.byref(s)) == 0, lib.pmc_last_error()
    assert lib.pmc_samples_count(s) == N
    total = 0
    for part in range(len(ids)):
        b, c = C.c_int64(), C.c_int64()
        assert lib.pmc_samples_shard(s, part, C.byref(b), C.byref(c)) == 0
        assert b.value == total
        total += c.value
    assert total == N
    # latent form with the origin the handle kept: alpha = counts / N, mu and sigma the blocks' moments
    alpha, nmu, nsig, norm = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D)), np.zeros(1)
    assert lib.pmc_pmc_update_stats(h, q, s, None, 0, None, 0, dp(alpha), dp(nmu), dp(nsig), None, None, dp(norm)) == 0, \
        lib.pmc_last_error()
    assert norm[0] == N
    np.testing.assert_allclose(alpha, counts / N, rtol=1e-13)
    for k in range(K):
        assert abs(nmu[k, 0] - mu[k, 0]) < 6 * np.sqrt(var[k] / counts[k])
        assert abs(nsig[k, 0, 0] / var[k] - 1) < 6 * np.sqrt(2.0 / counts[k])
    # Rao-Blackwellised form: alpha sums to one, and E_q[log q] = -entropy to Monte-Carlo accuracy
    ll = np.zeros(1)
    assert lib.pmc_pmc_update_stats(h, q, s, None, 0, None, 1, dp(alpha), dp(nmu), dp(nsig), None, dp(ll), dp(norm)) == 0
    assert abs(alpha.sum() - 1) < 1e-12 and np.abs(alpha - counts / N).max() < 2e-4
    assert np.isfinite(ll[0]) and -2.5 < ll[0] / N < -1.0
    # VB E-step with the front-end's start values for this mixture (its constructor wants samples: a small host batch)
    np.random.seed(3)
    vb = GaussianInference(mixture.propose(4000), initial_guess=mixture)
    m, W, nu, beta, ln_pi, ln_lam = [np.ascontiguousarray(a, dtype=np.float64) for a in
                                     (vb.m, vb.W, vb.nu, vb.beta, vb.expectation_ln_pi, vb.expectation_det_ln_lambda)]
    Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
    assert lib.pmc_vb_estep(h, s, None, K, dp(m), dp(W), dp(nu), dp(beta), dp(ln_pi), dp(ln_lam), None,
                            dp(Nk), dp(xbar), dp(S), dp(elq), None, None) == 0, lib.pmc_last_error()
    assert abs(Nk.sum() / N - 1) < 1e-11 and np.abs(Nk / N - counts / N).max() < 2e-2
    assert np.isfinite(elq[0]) and elq[0] <= 0
    # weighted moments (w = 1): the mixture's mean and variance
    mean, cov = np.empty(D), np.empty((D, D))
    assert lib.pmc_weighted_moments(h, s, None, 0, dp(mean), dp(cov)) == 0
    f = counts / N
    mix_mean = float((f * mu[:, 0]).sum())
    mix_var = float((f * (var + mu[:, 0] ** 2)).sum() - mix_mean ** 2)
    assert abs(mean[0] - mix_mean) < 6 * np.sqrt(mix_var / N) and abs(cov[0, 0] / mix_var - 1) < 1e-3
    lib.pmc_samples_free(s)
    lib.pmc_mixture_destroy(q)
    assert lib.pmc_shutdown(h) == 0
