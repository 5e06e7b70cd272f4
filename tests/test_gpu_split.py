"""The components of a sample block split over workgroups (k_logpdf_split / k_resp_groups_split, round 6): a small batch
is spread over the chip instead of costing K component steps on a handful of compute units, and the last round of a
larger launch ends on short pieces.  A piece leaves (maximum, sum) per sample, the piece that draws the block's last
ticket combines them in piece order (_regularize.pyx:72-81 about the row maximum).

Held here: oracle parity at N in {1, 63, 64, 65, 255, 256, 257, 4096, 65536} for every piece count the options can
force, bitwise run-to-run determinism, the per-pair outputs (individual, kept forms), the zero-weight maximum rule,
NaN rows, the target mixture in pieces, and -- for the grouped responsibilities -- the bits of the unsplit kernel."""
import numpy as np
import pytest
from scipy.special import digamma

from test_gpu_kernels import mk, draw, gauss_set, student_set, assert_rel

pytestmark = pytest.mark.gpu

SIZES = (1, 63, 64, 65, 255, 256, 257, 4096, 65536)
DEFAULTS = dict(split_components=1, split_min_components=0, split_tail_pieces=4, split_max_rounds=24, split_fill=1.0, split_max_pieces=16,
                split_tail_rounds=0.25, split_tail_min_components=0)


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    b = HipBackend()
    b.configure("maha_gemm_min_n", 2 ** 40)              # the exact kernels: the matrix-product form has tests of its own
    for k, v in DEFAULTS.items():
        assert b.option_default(k) == v, k                   # (this file's idea of the defaults is the library's)
    yield b
    for k in DEFAULTS:
        b.reset_option(k)
    b.reset_option("maha_gemm_min_n")


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


class options(object):
    def __init__(self, be, **kw):
        self.be, self.kw = be, kw

    def __enter__(self):
        for k, v in self.kw.items():
            self.be.configure(k, v)

    def __exit__(self, *exc):
        for k in self.kw:
            self.be.configure(k, DEFAULTS[k])


def plans():
    """(label, options): every piece count of the plan -- one component per piece up to everything in one piece"""
    fine = dict(split_fill=64., split_max_pieces=1024)
    return [("auto", {}), ("1 per piece", dict(split_min_components=1, **fine)),
            ("2 per piece", dict(split_min_components=2, **fine)),
            ("3 per piece", dict(split_min_components=3, **fine)),
            ("7 per piece", dict(split_min_components=7, **fine)),
            ("off", dict(split_components=0))]


@pytest.mark.parametrize("D,K", [(2, 3), (8, 5), (20, 32), (20, 16), (24, 33), (30, 8), (40, 12), (64, 5), (13, 9)])
def test_gauss_logpdf_in_pieces_vs_oracle(be, orc, D, K):
    mu, cov, w = mk(K, D, 300 + D + K)
    cs, inv, ln = gauss_set(mu, cov, w)
    xall, _ = draw(mu, cov, w, max(SIZES), 11)
    ref_all, ind_all = orc.mixture_multi_evaluate(0, xall[:4096], w, mu, inv, ln)
    for N in SIZES:
        if N > 4096 and D * K > 700:
            continue                                         # (the oracle on one core: seconds)
        x = xall[:N]
        if N <= 4096:
            ref, ind = ref_all[:N], ind_all[:N]
        else:
            ref, ind = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
        outs = {}
        for label, opt in plans():
            with options(be, **opt):
                res = be.logpdf(x, cs, want_out=True, want_individual=True)
                out = be.tohost(res["out"]).copy()
                assert_rel(out, ref, what="out N=%d %s" % (N, label))
                assert_rel(be.tohost(res["individual"]), ind, what="individual N=%d %s" % (N, label))
                again = be.logpdf(x, cs, want_out=True)
                np.testing.assert_array_equal(be.tohost(again["out"]), out, err_msg="run to run N=%d %s" % (N, label))
                outs[label] = out
        # whatever the pieces: the same numbers to the rounding of the merge
        for label, out in outs.items():
            np.testing.assert_allclose(out, outs["off"], rtol=2e-15, atol=1e-14, err_msg=label)


@pytest.mark.parametrize("D,K,dof", [(3, 6, 2.5), (20, 16, 8.), (30, 32, 8.), (40, 9, 4.)])
def test_student_logpdf_in_pieces_vs_oracle(be, orc, D, K, dof):
    mu, cov, w = mk(K, D, 400 + D)
    dofs = np.full(K, dof) + 0.25 * np.arange(K)
    cs, inv, ln, pf, idf = student_set(mu, cov, w, dofs)
    xall, _ = draw(mu, cov * 1.5, w, 4096, 12)
    ref_all, ind_all = orc.mixture_multi_evaluate(1, xall, w, mu, inv, ln, pf, idf)
    for N in (1, 65, 256, 257, 4096):
        for label, opt in plans():
            with options(be, **opt):
                res = be.logpdf(xall[:N], cs, want_out=True, want_individual=True)
                assert_rel(be.tohost(res["out"]), ref_all[:N], what="out N=%d %s" % (N, label))
                assert_rel(be.tohost(res["individual"]), ind_all[:N], what="individual N=%d %s" % (N, label))


@pytest.mark.parametrize("D,K,KT,kinds", [(20, 32, 4, "gg"), (5, 9, 7, "gt"), (30, 32, 4, "tg"), (40, 16, 12, "gg"),
                                          (20, 3, 40, "gg"), (8, 1, 1, "gg")])
def test_importance_weights_in_pieces(be, orc, D, K, KT, kinds):
    """proposal AND target mixture in pieces: weights and their sums against the oracle, log P bitwise the target's own
    log-pdf call (include/pmc_hip.h), the two-launch path bitwise"""
    def build(kind, K_, seed):
        mu, cov, w = mk(K_, D, seed)
        if kind == "g":
            cs, inv, ln = gauss_set(mu, cov, w)
            return cs, lambda x: orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)[0]
        dofs = np.full(K_, 5.0)
        cs, inv, ln, pf, idf = student_set(mu, cov, w, dofs)
        return cs, lambda x: orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, pf, idf)[0]
    prop, ref_q = build(kinds[0], K, 70 + D)
    tgt, ref_t = build(kinds[1], KT, 80 + D)
    xall, _ = draw(*mk(K, D, 70 + D), 4096, 3)
    for N in (1, 64, 257, 4096):
        x = xall[:N]
        wref = orc.is_weights(ref_t(x), ref_q(x))
        for label, opt in plans():
            with options(be, **opt):
                fused = be.importance_weights(x, prop, tgt, want_out=True, want_log_target=True)
                lt = be.logpdf(x, tgt)["out"]
                two = be.logpdf(x, prop, log_target=lt, want_scalars=True)
                what = "N=%d %s" % (N, label)
                np.testing.assert_array_equal(be.tohost(fused["log_target"]), be.tohost(lt), err_msg=what)
                np.testing.assert_array_equal(be.tohost(fused["weights"]), be.tohost(two["weights"]), err_msg=what)
                np.testing.assert_array_equal(be.tohost(fused["scalars"]), be.tohost(two["scalars"]), err_msg=what)
                assert_rel(be.tohost(fused["weights"]), wref, what="weights " + what)
                sc = be.tohost(fused["scalars"])
                np.testing.assert_allclose(sc[0], wref.sum(), rtol=1e-11)
                np.testing.assert_allclose(sc[2], (wref ** 2).sum(), rtol=1e-11)


def test_zero_weight_components_and_the_row_maximum(be, orc):
    """_regularize.pyx:73-77: a component without weight still takes part in the row's maximum -- also when it sits in a
    piece of its own, and when it is the only one near the sample (the reference's degraded sum / log 0 there)"""
    D, K = 20, 12
    mu, cov, w = mk(K, D, 77)
    w0 = w.copy()
    w0[[0, 5, 11]] = 0.
    w0 /= w0.sum()
    cs, inv, ln = gauss_set(mu, cov, w0)
    x, _ = draw(mu, cov, w, 700, 5)                           # samples of the dead components too
    x[:5] = mu[5] + 1e-3
    ref, _ = orc.mixture_multi_evaluate(0, x, w0, mu, inv, ln)
    for label, opt in plans():
        with options(be, **opt):
            assert_rel(be.tohost(be.logpdf(x, cs)["out"]), ref, what=label)
    # dead components 70 sigma from everything: -inf rows of the reference
    far = mu.copy()
    far[1:] += 500.
    wl = np.zeros(K)
    wl[1:] = 1. / (K - 1)
    cs2, inv2, ln2 = gauss_set(far, cov, wl)
    xs = far[0] + 0.01 * np.random.RandomState(2).normal(size=(300, D))
    with np.errstate(divide="ignore"):
        ref2, _ = orc.mixture_multi_evaluate(0, xs, wl, far, inv2, ln2)
    assert np.isneginf(ref2).all()
    for label, opt in plans():
        with options(be, **opt):
            np.testing.assert_array_equal(be.tohost(be.logpdf(xs, cs2)["out"]), ref2, err_msg=label)


def test_nan_rows_and_far_points(be):
    """a NaN coordinate poisons its row (and only its row) whichever piece meets it; a component value of -inf drops out"""
    D, K = 20, 16
    mu, cov, w = mk(K, D, 9)
    cs = gauss_set(mu, cov, w)[0]
    x, _ = draw(mu, cov, w, 1000, 5)
    x[17, 3] = np.nan
    x[300, 0] = np.inf
    x[999, 19] = 1e200
    base = None
    for label, opt in plans():
        with options(be, **opt):
            res = be.logpdf(x, cs, log_target=np.zeros(len(x)), want_scalars=True)
            out = be.tohost(res["out"])
            assert np.isnan(out[17]), label
            assert np.isneginf(out[300]) and np.isneginf(out[999]), label          # every component value -inf: log 0
            ok = np.ones(len(x), bool)
            ok[[17, 300, 999]] = False
            assert np.isfinite(out[ok]).all(), label
            if base is None:
                base = out
            np.testing.assert_allclose(out[ok], base[ok], rtol=2e-15, err_msg=label)
            assert np.isnan(be.tohost(res["scalars"])[0]), label


def test_kept_forms_from_pieces(be):
    """pmc_mixture_logpdf_keep: the Mahalanobis forms leave from the pieces themselves -- the same tiles, bitwise"""
    D, K, N = 20, 32, 3000
    mu, cov, w = mk(K, D, 31)
    cs = gauss_set(mu, cov, w)[0]
    x, _ = draw(mu, cov, w, N, 5)
    with options(be, split_components=0):
        ref = be.logpdf(x, cs, keep=True)
        tiles_ref = be.tohost(ref["tiles"].data).copy()
    for label, opt in plans()[:-1]:
        with options(be, **opt):
            res = be.logpdf(x, cs, keep=True)
            n = (N + 63) // 64 * 64 * K
            np.testing.assert_array_equal(be.tohost(res["tiles"].data)[:n], tiles_ref[:n], err_msg=label)


def test_last_round_in_pieces(be, orc):
    """a launch that fills the chip: whole blocks first (k_logpdf's numbers, bitwise), the last round in pieces"""
    D, K = 20, 16
    mu, cov, w = mk(K, D, 55)
    mu *= 0.1                                                 # overlapping components: every piece matters to every sample
    cs, inv, ln = gauss_set(mu, cov, w)
    N = 256 * (2 * 256 * 4 + 300) + 77                        # two rounds of the chip + 300 blocks + a ragged one
    x, _ = draw(mu, cov, w, N, 4)
    xd = be.asdevice(x)
    with options(be, split_components=0):
        whole = be.tohost(be.logpdf(xd, cs)["out"]).copy()
    res = be.logpdf(xd, cs, log_target=be.zeros(N), want_scalars=True)
    out = be.tohost(res["out"])
    np.testing.assert_allclose(out, whole, rtol=2e-15)
    differ = np.nonzero(out != whole)[0]
    assert differ.size > 0 and differ.min() >= 256 * 1000, "whole blocks must keep k_logpdf's bits"
    sel = np.concatenate([np.arange(0, 2000), np.arange(N - 3000, N)])
    ref, _ = orc.mixture_multi_evaluate(0, x[sel], w, mu, inv, ln)
    assert_rel(out[sel], ref, what="last round in pieces")
    np.testing.assert_allclose(be.tohost(res["scalars"])[0], np.exp(-whole).sum(), rtol=1e-11)
    np.testing.assert_array_equal(be.tohost(be.logpdf(xd, cs)["out"]), out)


def _vb_set(K, D, seed):
    from pypmc_amd.backend import ComponentSet
    mu, cov, w = mk(K, D, seed)
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
    return cs, (mu, cov, w), (m, W, beta, nu, ln_pi, ln_lambda)


# (N >= 16384: below, the common-shift statistics -- and with them this pair of kernels -- are never taken, whatever the options)
@pytest.mark.parametrize("D,K,N,weighted", [(20, 32, 20000, True), (20, 64, 18000, False), (20, 70, 17000, True),
                                            (8, 48, 20000, False), (40, 48, 16500, True), (20, 64, 300000, True)])
def test_grouped_responsibilities_in_pieces_keep_their_bits(be, orc, D, K, N, weighted):
    """k_resp_groups_split: groups of 16 components in pieces, k_resp_groups' recurrence over the groups by the piece
    that finishes the block -- the statistics and E[log q(Z)] of the E-step, BITWISE, whatever the pieces; the oracle's
    numbers at 1e-10"""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    cs, (mu, cov, w), (m, W, beta, nu, ln_pi, ln_lambda) = _vb_set(K, D, 600 + K)
    x, _ = draw(mu, cov, w, N, 21)
    sw = np.random.RandomState(3).uniform(0.5, 1.5, N) if weighted else None
    xd = be.asdevice(x)
    be.configure("stats_common_shift_min_n", 0)              # the grouped form at every batch size
    be.configure("estep_grouped_responsibilities", 2)
    try:
        with options(be, split_components=0):
            whole = be.tohost(be.estep(xd, cs, 0, sample_w=sw)["stats"]).copy()
        for label, opt in (("auto", {}), ("fine", dict(split_fill=64., split_max_pieces=1024)), ("tail 2", dict(split_tail_pieces=2))):
            with options(be, **opt):
                got = be.tohost(be.estep(xd, cs, 0, sample_w=sw)["stats"])
                np.testing.assert_array_equal(got, whole, err_msg=label)
        # Gaussian Rao-Blackwell PMC kind
        gs = gauss_set(mu, cov, w)[0]
        with options(be, split_components=0):
            whole_g = be.tohost(be.estep(xd, gs, 1, sample_w=sw)["stats"]).copy()
        with options(be, split_fill=64., split_max_pieces=1024):
            np.testing.assert_array_equal(be.tohost(be.estep(xd, gs, 1, sample_w=sw)["stats"]), whole_g)
    finally:
        be.configure("stats_common_shift_min_n", 524288)
        be.configure("estep_grouped_responsibilities", 1)
    if N <= 20000:
        ref = orc.vb_estep(x, sw, m, W, beta, nu, ln_pi, ln_lambda)
        sc, S0, M1, M2, _, _ = split_stats(whole, K, D)
        assert_rel(S0, ref["N_comp"], rtol=1e-10, what="N_comp")
        assert abs(sc[0] - ref["expectation_log_q_Z"]) <= 1e-10 * abs(ref["expectation_log_q_Z"]) + 1e-11


def test_plan_is_a_function_of_the_shape(be):
    """the pieces depend on (N, K, D) and the options only: two contexts, two streams, two calls -- the same bits"""
    import torch
    D, K, N = 20, 32, 10000
    mu, cov, w = mk(K, D, 1)
    cs = gauss_set(mu, cov, w)[0]
    x = be.asdevice(draw(mu, cov, w, N, 5)[0])
    a = be.tohost(be.logpdf(x, cs)["out"]).copy()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b = be.tohost(be.logpdf(x, cs)["out"]).copy()
    np.testing.assert_array_equal(a, b)
    for _ in range(20):
        np.testing.assert_array_equal(be.tohost(be.logpdf(x, cs)["out"]), a)


@pytest.mark.parametrize("D,K,N", [(20, 32, 10000), (40, 64, 4096), (8, 32, 65536), (20, 16, 600000)])
def test_no_stale_pieces(be, D, K, N):
    """The pieces of a block meet through memory that crosses XCDs (agent-scope accesses, no cache-wide fences): alternate
    two DIFFERENT inputs of one shape -- what the previous call left in the pieces' region is then wrong for this one -- and
    compare every call with the unsplit kernel's numbers"""
    mu, cov, w = mk(K, D, 123)
    cs = gauss_set(mu, cov, w)[0]
    rs = np.random.RandomState(4)
    xs = [be.asdevice(rs.normal(size=(N, D)) * s) for s in (2.0, 3.5)]
    with options(be, split_components=0):
        refs = [be.tohost(be.logpdf(x, cs)["out"]).copy() for x in xs]
        sums = [be.tohost(be.logpdf(x, cs, log_target=be.zeros(N), want_scalars=True)["scalars"]).copy() for x in xs]
    first = [None, None]
    for it in range(30):
        i = it & 1
        res = be.logpdf(xs[i], cs, log_target=be.zeros(N), want_scalars=True)
        out = be.tohost(res["out"])
        np.testing.assert_allclose(out, refs[i], rtol=2e-15, atol=1e-14, err_msg="call %d" % it)
        np.testing.assert_allclose(be.tohost(res["scalars"])[:4], sums[i][:4], rtol=1e-12)
        if first[i] is None:
            first[i] = out.copy()
        np.testing.assert_array_equal(out, first[i], err_msg="call %d: run to run" % it)


@pytest.mark.parametrize("D,K,N,weighted", [(20, 32, 10000, True), (20, 17, 257, False), (8, 64, 4096, True), (40, 128, 3000, False),
                                            (20, 70, 30000, True), (30, 33, 1000, False), (12, 48, 65, True)])
def test_small_batch_estep_in_pieces(be, orc, D, K, N, weighted):
    """pmc_estep of a batch that does not fill the chip (round 6): grouped responsibilities, the groups of 16 components in
    pieces, the factors multiplied into u by the workgroup that finishes a block, the per-component statistics kernel
    behind.  Against the oracle (variational.pyx:675-932, pmc.pyx:23-43 + :188-222), against the one-workgroup walk
    (k_resp), run to run."""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    cs, (mu, cov, w), (m, W, beta, nu, ln_pi, ln_lambda) = _vb_set(K, D, 800 + K + D)
    x, _ = draw(mu, cov, w, N, 27)
    sw = np.random.RandomState(4).uniform(0.5, 1.5, N) if weighted else None
    xd = be.asdevice(x)
    be.configure("estep_small_batch_pieces", 0)
    try:
        walk = be.tohost(be.estep(xd, cs, 0, sample_w=sw)["stats"]).copy()
    finally:
        be.reset_option("estep_small_batch_pieces")
    got = be.tohost(be.estep(xd, cs, 0, sample_w=sw)["stats"]).copy()
    np.testing.assert_array_equal(be.tohost(be.estep(xd, cs, 0, sample_w=sw)["stats"]), got)
    ps = 1 + D + D * (D + 1) // 2
    a, b = got[8:8 + K * ps].reshape(K, ps), walk[8:8 + K * ps].reshape(K, ps)
    assert (np.abs(a - b) / (np.abs(b).max(axis=1, keepdims=True) + 1e-300)).max() < 1e-11
    assert abs(got[0] - walk[0]) <= 1e-11 * abs(walk[0]) + 1e-12
    ref = orc.vb_estep(x, sw, m, W, beta, nu, ln_pi, ln_lambda)
    sc, S0, M1, M2, _, _ = split_stats(got, K, D)
    assert_rel(S0, ref["N_comp"], rtol=1e-10, what="N_comp")
    x_mean, S = centred_moments(S0, M1, M2, m)
    live = ref["N_comp"] > 1e-6
    np.testing.assert_allclose(x_mean[live], ref["x_mean_comp"][live], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(S[live], ref["S"][live], rtol=1e-8, atol=1e-10)
    assert abs(sc[0] - ref["expectation_log_q_Z"]) <= 1e-10 * abs(ref["expectation_log_q_Z"]) + 1e-11
    # the Gaussian Rao-Blackwell update's statistics (pmc.pyx:23-43, :188-222)
    gs, inv, ln = gauss_set(mu, cov, w)
    rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
    g = be.tohost(be.estep(xd, gs, 1, sample_w=sw)["stats"])
    swv = np.ones(N) if sw is None else sw
    _, S0g, M1g, _, _, _ = split_stats(g, K, D)
    np.testing.assert_allclose(S0g, (swv[:, None] * rho).sum(axis=0), rtol=1e-10, atol=1e-300)
    d = x[:, None, :] - mu[None]
    np.testing.assert_allclose(M1g, np.einsum('n,nk,nki->ki', swv, rho, d), rtol=1e-9, atol=1e-10)
    be.configure("estep_small_batch_pieces", 0)
    try:
        gw = be.tohost(be.estep(xd, gs, 1, sample_w=sw)["stats"])
    finally:
        be.reset_option("estep_small_batch_pieces")
    np.testing.assert_allclose(g[3], gw[3], rtol=1e-12)                         # sum w log q


def test_per_sample_outputs_independent_of_the_batch_when_pinned(be):
    """include/pmc_hip.h: with "split_components" 0 and the matrix-product form off, log q of a sample does not depend on the
    batch it arrives in -- bit for bit; with the defaults it agrees to the rounding of the merge"""
    D, K = 20, 32
    mu, cov, w = mk(K, D, 42)
    mu *= 0.2
    cs = gauss_set(mu, cov, w)[0]
    x = be.asdevice(draw(mu, cov, w, 70000, 9)[0])
    with options(be, split_components=0):
        big = be.tohost(be.logpdf(x, cs)["out"]).copy()
        for n in (1, 255, 1000, 4097, 30000):
            np.testing.assert_array_equal(be.tohost(be.logpdf(x[:n].contiguous(), cs)["out"]), big[:n])
    for n in (1, 255, 1000, 4097, 30000, 70000):
        np.testing.assert_allclose(be.tohost(be.logpdf(x[:n].contiguous(), cs)["out"]), big[:n], rtol=2e-15, atol=1e-14)
