"""The matrix-product form of the Mahalanobis forms (csrc/pmc_mgemm.hip; compiled D = 32, 40, 48, 64): against the oracle
(bilinear_sym -> multi_evaluate -> logsumexp2D, pypmc/tools/_linalg.pyx:10-39, density/gauss.pyx:146-151,
student_t.pyx:154-164, tools/_regularize.pyx:57-84; rho, pmc.pyx:23-43; the VB E-step, variational.pyx:675-1013),
against the exact kernels it stands in for, and on the cases its guard exists for -- means 30 sigma from the centre,
cond(Sigma) = 1e10, one far outlier per workgroup, NaN / inf coordinates, zero-weight components."""
import numpy as np
import pytest
from scipy.special import digamma

from test_gpu_kernels import mk, draw, gauss_set, student_set, assert_rel

pytestmark = pytest.mark.gpu
TOL = 5e-11          # the library's default "maha_gemm_tolerance"
def EPS_G(D):
    """the guard's error constant (csrc/pmc_api.hip::mgemm_eps): 3.5e-17 sqrt(number of monomials) of the compiled dimension"""
    Dc = 20 if D <= 20 else 24 if D <= 24 else 32 if D <= 32 else 40 if D <= 40 else 48 if D <= 48 else 64
    return 3.5e-17 * np.sqrt(0.5 * (Dc + 1.) * (Dc + 2.))


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    b = HipBackend()
    yield b
    b.configure("maha_gemm_tolerance", TOL)
    b.reset_option("maha_gemm_min_n")


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture()
def small(be):
    """the form from 1000 samples on, whatever an earlier test left behind (the default is 256 since round 5)"""
    be.configure("maha_gemm_min_n", 1000)
    be.configure("maha_gemm_tolerance", TOL)
    yield
    be.reset_option("maha_gemm_min_n")
    be.configure("maha_gemm_tolerance", TOL)


def exact(be, fn):
    """the exact kernels, one workgroup per block of samples (what runs behind the form for the workgroups it refused)"""
    be.configure("maha_gemm_tolerance", 0.0)
    be.configure("split_components", 0)
    try:
        return fn()
    finally:
        be.configure("maha_gemm_tolerance", TOL)
        be.reset_option("split_components")


def report(be, N, K, D):
    rep = be.maha_gemm_report(N, K, D)
    assert rep is not None, "the shape does not take the matrix-product form"
    return rep


def guard_bound(rep, mu, x):
    cen = 0.5 * (mu.min(axis=0) + mu.max(axis=0))
    dn = np.linalg.norm(x - cen, axis=1)
    return EPS_G(x.shape[1]) * (rep["norms"][0] * dn ** 2 + rep["norms"][1] * dn + rep["norms"][2])


CASES = [(32, 32, 3000), (32, 64, 2049), (31, 32, 1500), (40, 128, 2500), (40, 32, 1111), (40, 96, 1300), (37, 64, 1290),
         (48, 32, 1500), (48, 64, 1027), (44, 96, 1100), (40, 28, 1200), (40, 120, 1100),
         (64, 64, 1300), (64, 32, 1029), (56, 64, 1100), (49, 32, 1200), (64, 128, 1100), (61, 96, 1050),   # round 5: Dc = 64
         # round 5: Dc = 20 / 24, the non-emitting passes of mixtures with four full tiles per pass
         (20, 128, 3000), (19, 112, 1500), (17, 192, 1100), (24, 64, 2000), (24, 128, 1100), (22, 56, 1300), (21, 120, 1050)]


@pytest.mark.parametrize("D,K,N", CASES)
def test_gauss_logpdf_vs_oracle(be, orc, small, D, K, N):
    mu, cov, w = mk(K, D, 300 + D + K)
    x, _ = draw(mu, cov, w, N, 17)
    cs, inv, ln = gauss_set(mu, cov, w)
    ref, _ = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    got = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
    rep = report(be, N, K, D)
    assert rep["refused"] == 0, "the guard refused healthy data: %r" % rep
    assert_rel(got, ref, what="log q through the matrix product")
    # the guard's price covers what the form really costs (the difference to the exact kernel), with room to spare
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    assert_rel(ex, ref, rtol=1e-12, what="exact kernel")
    ratio = (np.abs(got - ex) / guard_bound(rep, mu, x)).max()
    assert ratio < 0.75, "difference / bound = %.3f" % ratio


@pytest.mark.parametrize("D,K,N,dof", [(32, 32, 2000, 8.), (40, 64, 1500, 3.), (40, 128, 1100, 50.), (48, 32, 1200, 5.),
                                       (36, 32, 1300, 1.5), (64, 64, 1200, 6.), (56, 32, 1100, 4.), (24, 64, 1500, 5.),
                                       (20, 128, 1200, 9.)])
def test_student_logpdf_vs_oracle(be, orc, small, D, K, N, dof):
    mu, cov, w = mk(K, D, 400 + D + K)
    x, _ = draw(mu, cov * 1.3, w, N, 18)
    dofs = np.full(K, dof) + 0.25 * (np.arange(K) % 7)
    cs, inv, ln, pf, idf = student_set(mu, cov, w, dofs)
    ref, _ = orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, pf, idf)
    # at the DEFAULT tolerance since round 5: the guard prices maha a priori and applies the pair's own slope
    # (nu + D) / (2 (nu + maha)) behind the product, instead of the worst slope (maha = 0) for every pair
    got = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
    rep = report(be, N, K, D)
    assert rep["refused"] == 0
    assert_rel(got, ref, what="Student-t log q through the matrix product")
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    assert 0 < np.abs(got - ex).max() < TOL


@pytest.mark.parametrize("D,K,N,nu,pruned", [(40, 64, 2600, 8., False), (40, 128, 1500, 3., False), (32, 32, 2500, 1.5, False),
                                             (64, 64, 1200, 8., False), (40, 64, 2600, 5., True), (48, 64, 1300, 30., True)])
def test_student_t_emitting_pass(be, orc, small, D, K, N, nu, pruned):
    """round 6: the emitting pass of a Student-t mixture through the matrix-product form (pmc.pyx:602-610: u = w rho gamma,
    gamma = (nu + D) / (nu + maha) from the t the epilogue holds) and the two sums of the degree-of-freedom condition
    (pmc.pyx:612, :654-691) from k_dof_sums behind it, which recovers log t from u itself -- against the oracle's rho and the
    exact kernel's sums; with a workgroup the guard refuses; with pruned components"""
    mu, cov, w = mk(K, D, 700 + D + K)
    x, _ = draw(mu, cov * 1.3, w, N, 23)
    dofs = np.full(K, nu) + 0.5 * (np.arange(K) % 4)
    wl = w.copy()
    if pruned:
        wl[[1, 17, K - 1]] = 0.
        wl /= wl.sum()
    live = [k for k in range(K) if wl[k] != 0]
    prop, inv, ln, pf, idf = student_set(mu, cov, wl, dofs)
    tmu, tcov, tw = mk(4, D, 78)
    target, tinv, tln = gauss_set(0.5 * tmu, tcov, tw)
    logq, _ = orc.mixture_multi_evaluate(1, x, wl, mu, inv, ln, pf, idf)
    logp, _ = orc.mixture_multi_evaluate(0, x, tw, 0.5 * tmu, tinv, tln)
    em = be.importance_weights(x, prop, target, want_out=True, emit=True)
    rep = report(be, N, K, D)
    assert rep["refused"] == 0, rep
    assert_rel(be.tohost(em["out"]), logq, what="log q")
    wts = be.tohost(em["weights"])
    assert_rel(wts, orc.is_weights(logp, logq), what="importance weights")
    resp = em["responsibilities"]
    assert resp.gscale is not None and resp.vsums is not None and resp.K == len(live)
    rho = orc.rho_rb(1, x, wl, mu, inv, ln, pf, idf, live)[:, live]
    dl = x[:, None, :] - mu[None, live]
    maha = np.einsum('nki,kij,nkj->nk', dl, inv[live], dl)
    wr = wts[:, None] * rho
    ref = wr * (dofs[live] + D) / (dofs[live] + maha)
    u = resp.host_matrix(be)
    normal = ref > 1e-280
    assert_rel(u[normal], ref[normal], what="u = w rho gamma")
    vs = be.tohost(resp.vsums).reshape(len(live), 2)
    np.testing.assert_allclose(vs[:, 0], wr.sum(axis=0), rtol=1e-10)
    np.testing.assert_allclose(vs[:, 1], (wr * np.log(.5 * (maha + dofs[live]))).sum(axis=0), rtol=1e-10, atol=1e-12 * np.abs(wr).sum())
    # the exact kernel's numbers (its own sums per tile)
    ex = exact(be, lambda: be.importance_weights(x, prop, target, emit=True)["responsibilities"])
    np.testing.assert_allclose(vs, be.tohost(ex.vsums).reshape(len(live), 2), rtol=1e-10, atol=1e-13 * np.abs(wr).sum())
    np.testing.assert_allclose(u, ex.host_matrix(be), rtol=1e-10, atol=1e-280)
    # a far outlier: its workgroup is refused and done by the exact kernel; the sums still cover every sample
    xo = x.copy()
    xo[300] = mu[0] + 1e5
    em2 = be.importance_weights(xo, prop, target, emit=True)
    rep2 = report(be, N, K, D)
    assert rep2["refused"] >= 1, rep2
    ex2 = exact(be, lambda: be.importance_weights(xo, prop, target, emit=True)["responsibilities"])
    v2, e2 = be.tohost(em2["responsibilities"].vsums), be.tohost(ex2.vsums)
    np.testing.assert_allclose(v2, e2, rtol=1e-10, atol=1e-13 * np.abs(e2).max())
    np.testing.assert_allclose(em2["responsibilities"].host_matrix(be), ex2.host_matrix(be), rtol=1e-10, atol=1e-280)
    # run to run
    em3 = be.importance_weights(xo, prop, target, emit=True)
    np.testing.assert_array_equal(be.tohost(em3["responsibilities"].vsums), v2)


@pytest.mark.parametrize("D,K,N", [(32, 32, 2500), (40, 128, 1500), (40, 64, 1100), (48, 64, 1300), (35, 32, 1200), (64, 64, 1200),
                                   (56, 32, 1100)])
def test_importance_weights_and_emitted_responsibilities(be, orc, small, D, K, N):
    """configuration 5's pair of calls: weights + u = w rho (grouped: values x factors) + the statistics"""
    mu, cov, w = mk(K, D, 500 + D + K)
    x, _ = draw(mu, cov, w, N, 19)
    tmu, tcov, tw = mk(4, D, 78)
    prop, inv, ln = gauss_set(mu, cov, w)
    target, tinv, tln = gauss_set(0.5 * tmu, tcov, tw)
    logq, _ = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    logp, _ = orc.mixture_multi_evaluate(0, x, tw, 0.5 * tmu, tinv, tln)
    em = be.importance_weights(x, prop, target, want_out=True, want_log_target=True, emit=True)
    rep = report(be, N, K, D)
    assert rep["refused"] == 0
    assert_rel(be.tohost(em["out"]), logq, what="log q")
    assert_rel(be.tohost(em["log_target"]), logp, what="log P")
    wts = be.tohost(em["weights"])
    assert_rel(wts, orc.is_weights(logp, logq), what="importance weights")
    sc = be.tohost(em["scalars"])
    assert_rel(sc[:3], [wts.sum(), (wts * np.log(wts)).sum(), (wts ** 2).sum()], rtol=1e-11, what="weight sums")
    resp = em["responsibilities"]
    assert resp.gscale is not None
    rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
    u, ref = resp.host_matrix(be), wts[:, None] * rho
    normal = ref > 1e-290
    assert_rel(u[normal], ref[normal], what="u = w rho")
    assert np.all(u[~normal] <= 1e-280)
    # the statistics of these responsibilities against the exact pair of calls
    a = be.tohost(be.estep_from_u(x, prop, resp)["stats"])
    b = exact(be, lambda: be.tohost(be.estep_from_u(x, prop, be.importance_weights(x, prop, target, emit=True)["responsibilities"])
                                    ["stats"]))
    ps = 1 + D + D * (D + 1) // 2
    a, b = a[8:8 + K * ps].reshape(K, ps), b[8:8 + K * ps].reshape(K, ps)
    assert (np.abs(a - b) / (np.abs(b).max(axis=1, keepdims=True) + 1e-300)).max() < 1e-11
    # ... and used twice (the far-shift second pass of an update does that): the pair (u, factors) still means the same u
    a2 = be.tohost(be.estep_from_u(x, prop, resp)["stats"])[8:8 + K * ps].reshape(K, ps)
    assert (np.abs(a2 - a) / (np.abs(a).max(axis=1, keepdims=True) + 1e-300)).max() < 1e-12
    np.testing.assert_allclose(resp.host_matrix(be), u, rtol=1e-14, atol=0)


@pytest.mark.parametrize("D,K,N", [(24, 64, 2100), (20, 128, 1500), (23, 120, 1100)])
def test_small_dimensions_take_the_form_for_the_weighting_pass_and_not_for_emitting_passes(be, orc, small, D, K, N):
    """round 5: compiled dimensions 20 / 24 -- importance weights (no u asked for) run through the matrix product and agree
    with the oracle and, within the guard's price, with the exact kernels; the emitting pass and the E-step stay with the
    vector kernels (k_resp_groups is the faster one there): bit for bit the exact path's numbers"""
    mu, cov, w = mk(K, D, 900 + D + K)
    x, _ = draw(mu, cov, w, N, 23)
    tmu, tcov, tw = mk(4, D, 79)
    prop, inv, ln = gauss_set(mu, cov, w)
    target, tinv, tln = gauss_set(0.5 * tmu, tcov, tw)
    logq, _ = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    logp, _ = orc.mixture_multi_evaluate(0, x, tw, 0.5 * tmu, tinv, tln)
    res = be.importance_weights(x, prop, target, want_out=True, want_log_target=True)
    rep = report(be, N, K, D)
    assert rep["refused"] == 0 and rep["workgroups"] == -(-N // 256)
    got = be.tohost(res["out"])
    assert_rel(got, logq, what="log q")
    assert_rel(be.tohost(res["log_target"]), logp, what="log P")
    wts = be.tohost(res["weights"])
    assert_rel(wts, orc.is_weights(logp, logq), what="importance weights")
    assert_rel(be.tohost(res["scalars"])[:3], [wts.sum(), (wts * np.log(wts)).sum(), (wts ** 2).sum()], rtol=1e-11, what="weight sums")
    ex = exact(be, lambda: be.tohost(be.importance_weights(x, prop, target, want_out=True)["out"]))
    diff = np.abs(got - ex)
    assert diff.max() > 0, "the form did not run"
    assert (diff / guard_bound(rep, mu, x)).max() < 0.75
    # emitting pass and E-step: the exact path, whatever the tolerance
    em = be.importance_weights(x, prop, target, want_out=True, emit=True)
    em_out, em_u = be.tohost(em["out"]).copy(), em["responsibilities"].host_matrix(be)
    be.configure("estep_small_batch_pieces", 0)          # (round 6: a small batch's E-step goes in pieces of its own, to rounding)
    try:
        st = be.tohost(be.estep(x, prop, 1)["stats"]).copy()
    finally:
        be.reset_option("estep_small_batch_pieces")
    ex_em = exact(be, lambda: be.importance_weights(x, prop, target, want_out=True, emit=True))
    np.testing.assert_array_equal(em_out, be.tohost(ex_em["out"]))
    np.testing.assert_array_equal(em_u, ex_em["responsibilities"].host_matrix(be))
    np.testing.assert_array_equal(st, exact(be, lambda: be.tohost(be.estep(x, prop, 1)["stats"])))


@pytest.mark.parametrize("tag", ["d40k32", "d24k64"])
@pytest.mark.parametrize("tol", [TOL, 1.0])
def test_dead_components_far_away_against_the_reference_golden(be, small, tag, tol):
    """tests/golden/logpdf_dead_*.npz, generated from the reference: a mixture with pruned components, 384 samples of the
    live mixture and 128 that approach a dead component 70 sigma away -- there the reference's terms exp(a - max over ALL
    components) underflow and it returns log 0 = -inf (81 rows).  Through the matrix-product form: the first workgroup
    stays with it, the second is sent to the exact kernel -- a priori at the default tolerance (the samples are far from the
    centre), a posteriori (dead maximum - live maximum > 700) with the a-priori guard opened wide -- and every row agrees
    with the reference: the same -inf rows, 1e-10 elsewhere."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logpdf_dead_%s.npz" % tag))
    from pypmc_amd.backend import ComponentSet
    x, w, mu = g["x"], g["weights"], g["mu"]
    K, D = mu.shape
    inv = np.repeat(g["inv_sigma0"][None], K, axis=0)
    cs = ComponentSet(0, mu, inv, c0=np.full(K, float(g["log_norm0"])), weight=w)
    be.configure("maha_gemm_min_n", 256)
    be.configure("maha_gemm_tolerance", tol)
    try:
        assert be.lib.pmc_maha_gemm_tiles(len(x), K, D) > 0
        got = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
        rep = report(be, len(x), K, D)
    finally:
        be.configure("maha_gemm_tolerance", TOL)
        be.reset_option("maha_gemm_min_n")
    assert rep["workgroups"] == 2 and rep["refused"] == 1, rep
    ref = g["out"]
    assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and not np.isnan(got).any()
    fin = np.isfinite(ref)
    assert_rel(got[fin], ref[fin], what="log q against the reference's golden vector")
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    np.testing.assert_array_equal(got[256:], ex[256:])               # the redone workgroup: the exact kernel's numbers
    assert np.abs(got[:256] - ex[:256]).max() > 0                     # the other one: the form's


@pytest.mark.parametrize("tag", ["d40k32", "d64k64"])
def test_student_t_against_the_reference_golden(be, small, tag):
    """tests/golden/logpdf_student_shared_*.npz (StudentT.multi_evaluate + logsumexp2D of the reference, student_t.pyx:154-164)
    through the matrix-product form: every workgroup stays with it, 1e-10 against the reference's numbers"""
    import os
    from pypmc_amd.backend import ComponentSet
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logpdf_student_shared_%s.npz" % tag))
    x, w, mu, dof = g["x"], g["weights"], g["mu"], g["dof"]
    K, D = mu.shape
    inv = np.repeat(g["inv_sigma0"][None], K, axis=0)
    cs = ComponentSet(1, mu, inv, c0=g["log_norm"], c1=-.5 * (dof + D), c2=1. / dof, c3=dof, weight=w)
    be.configure("maha_gemm_min_n", 256)
    be.configure("maha_gemm_tolerance", TOL)
    try:
        got = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
        rep = report(be, len(x), K, D)
    finally:
        be.configure("maha_gemm_tolerance", TOL)
        be.reset_option("maha_gemm_min_n")
    assert rep["refused"] == 0 and rep["workgroups"] == 2, rep
    assert_rel(got, g["out"], what="Student-t log q against the reference's golden vector")
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    assert np.abs(got - ex).max() > 0, "the form did not run"
    assert_rel(ex, g["out"], rtol=1e-12, what="exact kernel against the golden vector")


def test_student_t_pairs_beyond_the_tolerance_are_found_behind_the_product(be, orc, small):
    """round 5: the Student-t guard.  A priori only the price of maha is known; the slope |da / dmaha| =
    (nu + D) / (2 (nu + maha)) is applied pair by pair in the epilogue.  Here: nu = 1.2, a mixture spread over +-30 (a large
    price far from the centre), and in ONE workgroup a few samples placed exactly on a component's mean (maha = 0: the largest
    slope there is).  A tolerance just above the planted pairs' bound lets every workgroup through; just below it, that
    workgroup -- and only that one -- is redone by the exact kernel."""
    D, K, N = 40, 32, 2048
    mu, cov, w = mk(K, D, 77)
    mu *= 3.0
    dofs = np.full(K, 1.2)
    x, _ = draw(mu, cov * 1.5, w, N, 41)
    x[1300:1304] = mu[7]                                           # workgroup 5
    cs, inv, ln, pf, idf = student_set(mu, cov, w, dofs)
    ref, _ = orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, pf, idf)
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    assert_rel(ex, ref, what="exact kernel, nu = 1.2")

    def run(tol):
        be.configure("maha_gemm_tolerance", tol)
        try:
            return be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]), report(be, N, K, D)
        finally:
            be.configure("maha_gemm_tolerance", TOL)

    got, rep = run(1.0)                                              # everything through the form: the norms, the real error
    assert rep["refused"] == 0
    assert_rel(got, ref, what="Student-t, nu = 1.2, through the form")
    # the bound of the planted samples' own pair: eps_g x Theta-sum (the price of maha) x slope(maha = 0) ...
    cen = 0.5 * (mu.min(axis=0) + mu.max(axis=0))
    dn = np.linalg.norm(x - cen, axis=1)
    price = EPS_G(D) * (rep["norms"][0] * dn ** 2 + rep["norms"][1] * dn + rep["norms"][2])
    planted = price[1300] * (dofs[7] + D) / (2 * dofs[7])
    # ... and of every other pair: drawn samples have maha >= D / 4 to every component
    others = np.delete(price, np.arange(1300, 1304)).max() * (dofs[0] + D) / (2 * (dofs[0] + 0.25 * D))
    assert planted > 1.5 * others, (planted, others)
    assert np.abs(got - ex).max() < others                           # (what the form really costs: far below its bound)
    got1, rep1 = run(1.25 * planted)
    assert rep1["refused"] == 0
    got2, rep2 = run(0.8 * planted)
    assert rep2["refused"] == 1, rep2
    np.testing.assert_array_equal(got2[1280:1536], ex[1280:1536])
    np.testing.assert_array_equal(np.delete(got2, np.arange(1280, 1536)), np.delete(got1, np.arange(1280, 1536)))


@pytest.mark.parametrize("D,K,N,student,dead", [(40, 64, 2100, False, False), (32, 32, 1500, False, True), (48, 64, 1300, True, False),
                                                (64, 64, 1100, False, True), (24, 64, 1500, False, False), (37, 128, 1200, True, True)])
def test_kept_forms_through_the_matrix_product(be, orc, small, D, K, N, student, dead):
    """round 5: the *_keep variants (the weighting pass leaves the Mahalanobis forms behind for pmc_estep_from_tiles: a PMC
    iteration that cannot emit -- pruned components -- evaluates its proposal once) no longer send the pass to the exact engine:
    the tiles are written from the accumulator layout, maha recovered from the value (Gauss: 2 ((c0 + log w) - value); Student-t:
    nu (t - 1)).  The kept forms against the exact kernel's and against the oracle's bilinear forms; the update's statistics
    from them against the exact path's."""
    from pypmc_amd.backend import ComponentSet
    mu, cov, w = mk(K, D, 990 + D + K)
    if dead:
        w = np.where(np.arange(K) % 4 == 1, 0.0, w)
        w /= w.sum()
    x, _ = draw(mu, cov * (1.3 if student else 1.0), np.full(K, 1.0 / K), N, 33)
    tmu, tcov, tw = mk(4, D, 84)
    target = gauss_set(0.5 * tmu, tcov, tw)[0]
    if student:
        dofs = np.full(K, 5.0) + 0.5 * (np.arange(K) % 3)
        cs, inv, ln, pf, idf = student_set(mu, cov, w, dofs)
    else:
        cs, inv, ln = gauss_set(mu, cov, w)
    kept = be.importance_weights(x, cs, target, want_out=True, keep=True)
    rep = report(be, N, K, D)
    assert rep["refused"] == 0, rep
    ex = exact(be, lambda: be.importance_weights(x, cs, target, want_out=True, keep=True))
    t_form, t_ex = be.tohost(kept["tiles"].data), be.tohost(ex["tiles"].data)
    ntiles = -(-N // 64)
    form = t_form[:ntiles * K * 64].reshape(ntiles, K, 64).transpose(0, 2, 1).reshape(-1, K)[:N]
    exa = t_ex[:ntiles * K * 64].reshape(ntiles, K, 64).transpose(0, 2, 1).reshape(-1, K)[:N]
    d = x[:, None, :] - mu[None]
    ref = np.einsum('nki,kij,nkj->nk', d, inv, d)                    # bilinear_sym, _linalg.pyx:10-39
    assert_rel(exa, ref, rtol=1e-11, what="kept forms of the exact kernel")
    assert np.abs(form - exa).max() > 0, "the form did not run"
    assert np.abs(form - ref).max() < 2 * TOL                        # (absolute: the guard's unit; maha = -2 (a - c0 - log w))
    assert 0 < np.abs(be.tohost(kept["out"]) - be.tohost(ex["out"])).max() < TOL
    # the update from the kept forms: live components only, as _prepare_pmc_update passes them
    live = np.flatnonzero(w > 0)
    sub = ComponentSet(cs.kind, mu[live], inv[live], c0=ln[live], c1=(pf[live] if student else None),
                       c2=(idf[live] if student else None), c3=(dofs[live] if student else None), weight=w[live], column=live, ld=K)
    wts = be.tohost(kept["weights"])
    a_ = be.tohost(be.estep_from_tiles(x, sub, kept["tiles"], max_init_zero=dead, sample_w=wts)["stats"])
    b_ = be.tohost(be.estep_from_tiles(x, sub, ex["tiles"], max_init_zero=dead, sample_w=be.tohost(ex["weights"]))["stats"])
    ps = 1 + D + D * (D + 1) // 2
    Kl = len(live)
    a2, b2 = a_[8:8 + Kl * ps].reshape(Kl, ps), b_[8:8 + Kl * ps].reshape(Kl, ps)
    assert (np.abs(a2 - b2) / (np.abs(b2).max(axis=1, keepdims=True) + 1e-300)).max() < 1e-10


def vb_set(mu, cov, D, K, seed):
    from pypmc_amd.backend import ComponentSet
    rs = np.random.RandomState(seed)
    inv = np.linalg.inv(cov)
    inv = 0.5 * (inv + inv.transpose(0, 2, 1))
    nu, beta, alpha = D + 2. + rs.uniform(0, 3, K), 1. + rs.uniform(0, 3, K), 1. + rs.uniform(0, 3, K)
    W = inv / nu[:, None, None]
    ln_lambda = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + np.linalg.slogdet(W)[1]
    ln_pi = digamma(alpha) - digamma(alpha.sum())
    cs = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    return cs, W, beta, nu, ln_pi, ln_lambda


@pytest.mark.parametrize("D,K,N,weighted", [(32, 32, 20000, False), (40, 64, 17000, True), (40, 128, 16500, False),
                                            (48, 32, 18000, True), (64, 64, 17000, False), (57, 32, 16500, True)])
def test_estep_vs_oracle(be, orc, small, D, K, N, weighted):
    """pmc_estep (VB and Gaussian Rao-Blackwell PMC): k_mgemm's grouped responsibilities + the common-shift statistics.
    Overlapping components (responsibilities that are not one-hot), so that the soft-max itself is tested."""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    mu, cov, w = mk(K, D, 600 + D + K)
    mu = 0.2 * mu
    x, _ = draw(mu, cov, w, N, 21)
    sw = np.random.RandomState(5).uniform(0.5, 1.5, N) if weighted else None
    cs, W, beta, nu, ln_pi, ln_lambda = vb_set(mu, cov, D, K, 3)
    o = orc.vb_estep(x, sw, mu, W, beta, nu, ln_pi, ln_lambda, mt=True)
    assert (o["r"].max(axis=1) < 0.999).mean() > 0.05, "the test data should not be one-hot"
    be.configure("stats_common_shift_min_n", 0)
    try:
        e = be.tohost(be.estep(x, cs, 0, sample_w=sw)["stats"])
        rep = report(be, N, K, D)
        assert rep["refused"] == 0
        sc, S0, M1, M2, _, _ = split_stats(e, K, D)
        xbar, S = centred_moments(S0, M1, M2, mu)
        assert_rel(S0, o["N_comp"], what="N_k")
        assert abs(sc[0] - o["expectation_log_q_Z"]) <= 1e-10 * abs(o["expectation_log_q_Z"])
        np.testing.assert_allclose(xbar, o["x_mean_comp"], rtol=1e-10, atol=1e-11)
        dg = np.sqrt(np.einsum('kii->ki', o["S"]))
        assert (np.abs(S - o["S"]) / (dg[:, :, None] * dg[:, None, :])).max() < 1e-10
        # Gaussian Rao-Blackwell PMC through the same kernel
        gs, inv, ln = gauss_set(mu, cov, w)
        eg = be.tohost(be.estep(x, gs, 1, sample_w=sw)["stats"])
        ex = exact(be, lambda: be.tohost(be.estep(x, gs, 1, sample_w=sw)["stats"]))
        ps = 1 + D + D * (D + 1) // 2
        a, b = eg[8:8 + K * ps].reshape(K, ps), ex[8:8 + K * ps].reshape(K, ps)
        assert (np.abs(a - b) / (np.abs(b).max(axis=1, keepdims=True) + 1e-300)).max() < 1e-11
        assert abs(eg[3] - ex[3]) <= 1e-11 * abs(ex[3])      # sum of sample_w log q
        rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
        wr = rho if sw is None else sw[:, None] * rho
        assert_rel(a[:, 0], wr.sum(axis=0), what="sum w rho")
    finally:
        be.configure("stats_common_shift_min_n", 524288)


# ---------------------------------------------------------------------------------------------------------------
# the guard
# ---------------------------------------------------------------------------------------------------------------
def test_far_outliers_send_their_workgroups_to_the_exact_kernel(be, orc, small):
    """one far outlier per wavefront in some workgroups: exactly those workgroups fall back (bit for bit the exact
    kernel's numbers there), the others keep the matrix product"""
    D, K, N = 40, 64, 256 * 12 + 100
    mu, cov, w = mk(K, D, 777)
    x, _ = draw(mu, cov, w, N, 23)
    bad_blocks = [1, 4, 5, 11]
    for b in bad_blocks:
        for wave in range(4):
            x[256 * b + 64 * wave + 7 * wave + 3] += 4000.0 * (1 + wave)
    x[256 * 12 + 5] -= 1e5                                 # ... and one in the ragged last workgroup
    bad_blocks.append(12)
    cs, inv, ln = gauss_set(mu, cov, w)
    ref, _ = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    lt = np.random.RandomState(1).normal(size=N)
    res = be.logpdf(x, cs, want_scalars=True, log_target=lt)
    rep = report(be, N, K, D)
    assert rep["refused"] == len(bad_blocks) and rep["workgroups"] == 13
    got = be.tohost(res["out"])
    assert_rel(got, ref, what="log q with outliers")
    ex = exact(be, lambda: be.logpdf(x, cs, want_scalars=True, log_target=lt))
    exo = be.tohost(ex["out"])
    inbad = np.zeros(N, dtype=bool)
    for b in bad_blocks:
        inbad[256 * b:256 * (b + 1)] = True
    np.testing.assert_array_equal(got[inbad], exo[inbad])
    assert (got[~inbad] != exo[~inbad]).any()
    np.testing.assert_array_equal(be.tohost(res["weights"])[inbad], be.tohost(ex["weights"])[inbad])
    assert_rel(be.tohost(res["scalars"])[:4], be.tohost(ex["scalars"])[:4], rtol=1e-11, what="scalar sums across both kernels")
    # the emitting pass and the E-step with the same outliers: complete u (factors one) in the refused workgroups
    target = gauss_set(*mk(3, D, 79))[0]
    em = be.importance_weights(x, cs, target, emit=True)
    assert report(be, N, K, D)["refused"] == len(bad_blocks)
    eme = exact(be, lambda: be.importance_weights(x, cs, target, emit=True))
    np.testing.assert_array_equal(be.tohost(em["weights"])[inbad], be.tohost(eme["weights"])[inbad])
    u, ue = em["responsibilities"].host_matrix(be), eme["responsibilities"].host_matrix(be)
    np.testing.assert_array_equal(u[inbad], ue[inbad])
    big = ue > 1e-250
    assert (np.abs(u - ue)[big] / ue[big]).max() < 1e-10
    f = be.tohost(em["responsibilities"].gscale).reshape(-1, (K + 15) // 16, 64)
    assert np.all(f[4 * 4:4 * 6] == 1.0) and not np.all(f[0:4] == 1.0)
    st = be.tohost(be.estep_from_u(x, cs, em["responsibilities"])["stats"])
    ste = exact(be, lambda: be.tohost(be.estep_from_u(x, cs, eme["responsibilities"])["stats"]))
    ps = 1 + D + D * (D + 1) // 2
    a, b = st[8:8 + K * ps].reshape(K, ps), ste[8:8 + K * ps].reshape(K, ps)
    assert (np.abs(a - b) / (np.abs(b).max(axis=1, keepdims=True) + 1e-300)).max() < 1e-10


def test_means_30_sigma_from_the_centre(be, orc, small):
    """components 30 sigma and more from the common centre: either the guard prices every sample out (then the numbers
    are the exact kernel's) or what it lets through holds the contract"""
    D, K, N = 40, 32, 3000
    mu, cov, w = mk(K, D, 31)
    mu = mu * 12.0                                        # |mu_k - c| ~ 36 sigma per coordinate
    x, _ = draw(mu, cov, w, N, 5)
    cs, inv, ln = gauss_set(mu, cov, w)
    ref, _ = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    got = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
    rep = report(be, N, K, D)
    assert_rel(got, ref, what="30 sigma")
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    bound = guard_bound(rep, mu, x)
    assert rep["refused"] == rep["workgroups"] == 12 and bound.min() > TOL
    np.testing.assert_array_equal(got, ex)
    # the same mixture with the tolerance opened wide: the form runs, and its error stays inside its price
    be.configure("maha_gemm_tolerance", 1e-6)
    wide = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
    assert report(be, N, K, D)["refused"] == 0
    be.configure("maha_gemm_tolerance", TOL)
    assert (np.abs(wide - ex) / bound).max() < 0.75


def test_condition_number_1e10(be, orc, small):
    D, K, N = 32, 32, 2000
    mu, cov, w = mk(K, D, 41)
    rs = np.random.RandomState(2)
    for k in range(K):
        q, _ = np.linalg.qr(rs.normal(size=(D, D)))
        cov[k] = (q * np.logspace(-5, 5, D)).dot(q.T)
        cov[k] = 0.5 * (cov[k] + cov[k].T)
    x, _ = draw(mu, cov, w, N, 6)
    cs, inv, ln = gauss_set(mu, cov, w)
    got = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
    rep = report(be, N, K, D)
    assert rep["refused"] == rep["workgroups"], "precisions of norm 1e5 and more cannot be priced in"
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    np.testing.assert_array_equal(got, ex)


def test_non_finite_coordinates_and_zero_weights(be, orc, small):
    D, K, N = 40, 32, 1500
    mu, cov, w = mk(K, D, 51)
    x, _ = draw(mu, cov, w, N, 7)
    x[300, 3] = np.nan
    x[800, 17] = np.inf
    cs, inv, ln = gauss_set(mu, cov, w)
    got = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
    rep = report(be, N, K, D)
    assert rep["refused"] == 2
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    assert np.isnan(got[300]) and np.array_equal(np.isnan(got), np.isnan(ex))
    np.testing.assert_array_equal(got[256:512], ex[256:512])
    np.testing.assert_array_equal(got[768:1024], ex[768:1024])
    ref, _ = orc.mixture_multi_evaluate(0, x[:256], w, mu, inv, ln)
    assert_rel(got[:256], ref, what="the clean workgroups")
    # a negative weight has no logarithm and no meaning: the whole call stays with the exact kernel
    w0 = w.copy()
    w0[5] = -0.1
    cs0 = gauss_set(mu, cov, w0)[0]
    xc, _ = draw(mu, cov, w, N, 8)
    got0 = be.tohost(be.logpdf(xc, cs0, want_scalars=True)["out"])
    rep0 = report(be, N, K, D)
    assert rep0["refused"] == rep0["workgroups"]
    np.testing.assert_array_equal(got0, exact(be, lambda: be.tohost(be.logpdf(xc, cs0, want_scalars=True)["out"])))


@pytest.mark.parametrize("D,K,N,student", [(40, 64, 2500, False), (32, 32, 1500, False), (48, 64, 1300, True), (64, 64, 1200, False),
                                           (24, 64, 1500, False), (37, 128, 1100, True)])
def test_components_without_weight(be, orc, small, D, K, N, student):
    """round 5: a pruned component (weight 0, still in the mixture: pmc.pyx:109-117) no longer sends the whole call to the
    exact engine.  It takes part in the reference's row maximum with its unweighted value (logsumexp2D,
    _regularize.pyx:73-77) and adds nothing to the sum: the matrix kernel keeps it out of both and tests a posteriori that
    no dead component's value lies more than 700 above the live maximum -- the only case in which the reference's number
    differs (its terms exp(a - max) underflow); such workgroups go to the exact kernel.  Log-density, `individual` (the dead
    columns too, as the reference fills them), importance weights; against the oracle's reference loops."""
    mu, cov, w = mk(K, D, 950 + D + K)
    dead = np.arange(K) % 5 == 2
    w = np.where(dead, 0.0, w)
    w /= w.sum()
    x, _ = draw(mu, cov, np.full(K, 1.0 / K), N, 31)                # samples around the dead components too
    if student:
        dofs = np.full(K, 7.0) + 0.25 * (np.arange(K) % 5)
        cs, inv, ln, pf, idf = student_set(mu, cov, w, dofs)
        ev = lambda xs: orc.mixture_multi_evaluate(1, xs, w, mu, inv, ln, pf, idf)
    else:
        cs, inv, ln = gauss_set(mu, cov, w)
        ev = lambda xs: orc.mixture_multi_evaluate(0, xs, w, mu, inv, ln)
    try:
        ref, ref_ind = ev(x)
        res = be.logpdf(x, cs, want_individual=True, want_scalars=True)
        rep = report(be, N, K, D)
        assert rep["refused"] == 0, rep
        got, ind = be.tohost(res["out"]), be.tohost(res["individual"])
        assert_rel(got, ref, what="log q with dead components")
        assert_rel(ind, ref_ind, what="individual with dead components")
        ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
        assert np.abs(got - ex).max() > 0, "the form did not run"
        if student:
            assert np.abs(got - ex).max() < TOL
        else:
            assert (np.abs(got - ex) / guard_bound(rep, mu, x)).max() < 0.75
        # importance weights against a small target: the same pass with the weights behind it
        tmu, tcov, tw = mk(4, D, 83)
        target, tinv, tln = gauss_set(0.5 * tmu, tcov, tw)
        logp, _ = orc.mixture_multi_evaluate(0, x, tw, 0.5 * tmu, tinv, tln)
        iw = be.importance_weights(x, cs, target, want_out=True)
        assert report(be, N, K, D)["refused"] == 0
        assert_rel(be.tohost(iw["weights"]), orc.is_weights(logp, ref), what="importance weights with dead components")
        # a dead component FAR above every live one at some samples: points 45 sigma from all live components, right on a
        # dead one -- the reference's terms underflow there (log 0 = -inf or a degraded sum); those workgroups take the exact
        # kernel and give its numbers bit for bit, the others stay with the form
        far = mu[2] + 60.0 * np.sqrt(np.diag(cov[2]).max()) * np.ones(D) / np.sqrt(D) * np.sqrt(D)
        mu2 = mu.copy()
        mu2[2] = far                                                 # component 2 (dead) moved far away from everything
        xs = x.copy()
        xs[700:703] = far + 0.01
        if student:
            cs2 = student_set(mu2, cov, w, dofs)[0]
            ref2, _ = orc.mixture_multi_evaluate(1, xs, w, mu2, inv, ln, pf, idf)
        else:
            cs2 = gauss_set(mu2, cov, w)[0]
            ref2, _ = orc.mixture_multi_evaluate(0, xs, w, mu2, inv, ln)
        be.configure("maha_gemm_tolerance", 1.0)                     # (the a-priori guard out of the way: the centre moved)
        got2 = be.tohost(be.logpdf(xs, cs2, want_scalars=True)["out"])
        rep2 = report(be, N, K, D)
        ex2 = exact(be, lambda: be.tohost(be.logpdf(xs, cs2, want_scalars=True)["out"]))
        if not student:                                              # (Student-t tails are too heavy for an underflow at 60 sigma)
            assert rep2["refused"] == 1, rep2
            np.testing.assert_array_equal(got2[512:768], ex2[512:768])
        both = np.isfinite(ref2) & np.isfinite(got2)
        assert np.array_equal(np.isfinite(ref2), np.isfinite(got2))
        assert_rel(got2[both], ref2[both], rtol=1e-9, what="log q, dead component far away")
    finally:
        be.configure("maha_gemm_tolerance", TOL)


def test_bitwise_determinism_and_selection(be, small):
    D, K, N = 40, 128, 5000
    mu, cov, w = mk(K, D, 61)
    x, _ = draw(mu, cov, w, N, 9)
    x[1000] += 1e4                                        # (a refused workgroup in the mix)
    cs = gauss_set(mu, cov, w)[0]
    target = gauss_set(*mk(4, D, 62))[0]
    xd = be.asdevice(x)
    first = None
    for _ in range(6):
        em = be.importance_weights(xd, cs, target, want_out=True, emit=True)
        st = be.estep_from_u(xd, cs, em["responsibilities"])
        cur = [be.tohost(t).copy() for t in (em["out"], em["weights"], em["scalars"], em["responsibilities"].data,
                                            em["responsibilities"].gscale, st["stats"])]
        if first is None:
            first = cur
        for a, b in zip(first, cur):
            np.testing.assert_array_equal(a, b)
    # which shapes take the form: compiled D = 32, 40, 48, 64 (padded 31 ... 64), K within ~20 % of a multiple of 32 / 64,
    # N from the threshold on
    lib = be.lib
    assert lib.pmc_maha_gemm_tiles(N, 128, 40) == 4 and lib.pmc_maha_gemm_tiles(N, 32, 40) == 2
    assert lib.pmc_maha_gemm_tiles(N, 96, 33) == 2 and lib.pmc_maha_gemm_tiles(N, 64, 48) == 4 and lib.pmc_maha_gemm_tiles(N, 32, 48) == 2
    assert lib.pmc_maha_gemm_tiles(N, 100, 40) == 0 and lib.pmc_maha_gemm_tiles(N, 16, 40) == 0
    assert lib.pmc_maha_gemm_tiles(N, 128, 30) == 0 and lib.pmc_maha_gemm_tiles(N, 128, 64) == 4 and lib.pmc_maha_gemm_tiles(N, 64, 57) == 4 and lib.pmc_maha_gemm_tiles(N, 32, 64) == 2
    assert lib.pmc_maha_gemm_tiles(N, 128, 72) == 0 and lib.pmc_maha_gemm_tiles(999, 128, 40) == 0
    # D <= 24: four full tiles per pass or nothing (and never for an emitting pass: test_small_dimensions_... below)
    assert lib.pmc_maha_gemm_tiles(N, 128, 20) == 4 and lib.pmc_maha_gemm_tiles(N, 64, 20) == 0 and lib.pmc_maha_gemm_tiles(N, 32, 20) == 0
    assert lib.pmc_maha_gemm_tiles(N, 64, 24) == 4 and lib.pmc_maha_gemm_tiles(N, 32, 24) == 0 and lib.pmc_maha_gemm_tiles(N, 128, 16) == 0
    assert lib.pmc_maha_gemm_tiles(N, 120, 18) == 4 and lib.pmc_maha_gemm_tiles(N, 100, 18) == 0 and lib.pmc_maha_gemm_tiles(N, 128, 28) == 0
    be.configure("maha_gemm_min_n", 32768)                # (the option: round 4's default)
    assert lib.pmc_maha_gemm_tiles(N, 128, 40) == 0 and lib.pmc_maha_gemm_tiles(32768, 128, 40) == 4
    be.configure("maha_gemm_min_n", 256)                  # (round 5's default)
    assert lib.pmc_maha_gemm_tiles(255, 128, 40) == 0 and lib.pmc_maha_gemm_tiles(256, 128, 40) == 4
    be.reset_option("maha_gemm_min_n")                    # round 6: below 49152 samples the exact kernels in pieces are faster
    assert be.option("maha_gemm_min_n") == 49152
    assert lib.pmc_maha_gemm_tiles(49151, 128, 40) == 0 and lib.pmc_maha_gemm_tiles(49152, 128, 40) == 4


def test_front_end_iteration_takes_the_form(be):
    """ImportanceSampler.run_device(prepare_update=True) + gaussian_pmc at D = 40, K = 64, N above the default threshold:
    the update equals the one computed with the exact kernels to 1e-10"""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import ImportanceSampler
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    D, K, N = 40, 64, 60000
    tmu, tcov, tw = mk(4, D, 11)
    tmu /= 3.0
    target = create_gaussian_mixture(tmu, tcov, tw)
    which = np.arange(K) % 4

    def run(tol):
        be.configure("maha_gemm_tolerance", tol)
        try:
            proposal = create_gaussian_mixture(tmu[which] + np.random.RandomState(5).normal(0, 0.15, (K, D)), 1.5 * tcov[which])
            np.random.seed(100)
            sampler = ImportanceSampler(target.evaluate, proposal)
            r = sampler.run_device(N, trace_sort=True, prepare_update=True)
            new = gaussian_pmc(r["samples"], sampler.proposal, r["weights"], r["origin"], mincount=0, rb=True, copy=True,
                               mahalanobis=r["mahalanobis"], responsibilities=r["responsibilities"])
            return (np.array(new.weights), np.array([c.mu for c in new.components]),
                    np.array([c.sigma for c in new.components]), be.tohost(r["weights"]))
        finally:
            be.configure("maha_gemm_tolerance", TOL)
    a = run(TOL)
    from pypmc_amd.backend import get_backend
    assert report(get_backend(), N, K, D)["refused"] == 0     # (the front-end's own backend object and workspace)
    b = run(0.0)
    assert_rel(a[3], b[3], what="importance weights")
    np.testing.assert_allclose(a[0], b[0], rtol=1e-10)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-10, atol=1e-12)
    dg = np.sqrt(np.einsum('kii->ki', b[2]))
    assert (np.abs(a[2] - b[2]) / (dg[:, :, None] * dg[:, None, :])).max() < 1e-10


def test_grouped_pair_completed_in_place_bit_for_bit(be, small):
    """advice r3: a bitwise regression for the path behind grouped responsibilities.  When the statistics kernel that
    applies the per-(sample, group) factors is not the one the shape gets, pmc_estep_from_u_grouped completes u in place
    (k_apply_scale), sets the factors to one (k_reset_scale) and runs the per-component kernel: bit for bit the statistics
    of pmc_sufficient_stats on the product u' f formed on the host (one multiplication per pair either way)."""
    import ctypes as C
    from pypmc_amd import _lib
    D, K, N = 40, 64, 3000
    mu, cov, w = mk(K, D, 91)
    x, _ = draw(mu, cov, w, N, 10)
    prop = gauss_set(mu, cov, w)[0]
    target = gauss_set(*mk(3, D, 92))[0]
    em = be.importance_weights(x, prop, target, emit=True)
    assert report(be, N, K, D)["refused"] == 0
    resp = em["responsibilities"]
    tile, nt, ng = be.tile, (N + 63) // 64, (K + 15) // 16
    vals = be.tohost(resp.data)[:nt * K * tile].reshape(nt, K, tile)
    fac = be.tohost(resp.gscale)[:nt * ng * tile].reshape(nt, ng, tile)
    assert not np.all(fac == 1.0)
    complete = vals * np.repeat(fac, 16, axis=1)[:, :K, :]          # u = u' f, the same single product the kernel forms
    ps = 1 + D + D * (D + 1) // 2
    xd = be.asdevice(x)
    ref = be.zeros(K * ps)
    ud = be.asdevice(complete.reshape(-1))
    _lib.check(be.lib.pmc_sufficient_stats(be._p(xd), N, D, be._p(be.pack(prop)), K, be._p(ud), be._p(ref),
                                           be._p(be._workspace(N, K, D)), be._stream()), "pmc_sufficient_stats")
    be.configure("stats_common_shift_min_k", 1e9)          # the common-shift statistics off: factors without their consumer
    try:
        got = be.tohost(be.estep_from_u(xd, prop, resp)["stats"])[8:8 + K * ps]
    finally:
        be.configure("stats_common_shift_min_k", 17)
    np.testing.assert_array_equal(got, be.tohost(ref))
    np.testing.assert_array_equal(be.tohost(resp.data)[:nt * K * tile].reshape(nt, K, tile), complete)
    assert np.all(be.tohost(resp.gscale)[:nt * ng * tile] == 1.0)
    # ... and a second use of the (now complete) pair gives the same bits
    be.configure("stats_common_shift_min_k", 1e9)
    try:
        again = be.tohost(be.estep_from_u(xd, prop, resp)["stats"])[8:8 + K * ps]
    finally:
        be.configure("stats_common_shift_min_k", 17)
    np.testing.assert_array_equal(again, got)


@pytest.mark.parametrize("D,K,Kt,N", [(40, 32, 33, 200000), (32, 32, 40, 65536), (48, 64, 65, 70000), (40, 32, 33, 40000)])
def test_target_with_more_components_than_the_proposal_on_a_fresh_workspace(orc, small, D, K, Kt, N):
    """advice r4: the workspace is sized for max(K, K_target) while the matrix-product form places its region by the
    PROPOSAL's K, and the layout was not monotone in K (K = 32 needed 62 MB, K = 33 36 MB at D = 40, N = 2e5): 25 MB were
    written past the allocation.  A fresh backend (a workspace of exactly the contract's size) with a poisoned fence
    behind it: nothing beyond pmc_workspace_bytes(N, max(K, K_target), D) may be touched, and the weights are right."""
    import torch
    from pypmc_amd.backend import HipBackend
    b = HipBackend()
    need = int(b.lib.pmc_workspace_bytes(N, max(K, Kt), D))
    assert need >= int(b.lib.pmc_workspace_bytes(N, K, D))
    fence = 1 << 20
    raw = torch.full((need + fence,), 0x5a, dtype=torch.uint8, device=b.device)
    key = torch.cuda.current_stream(b.device).cuda_stream
    b._ws[key] = raw[:need]                              # the workspace the calls below get: exactly the contract's size
    mu, cov, w = mk(K, D, 900 + D + K)
    x, _ = draw(mu, cov, w, N, 23)
    tmu, tcov, tw = mk(Kt, D, 91)
    prop, inv, ln = gauss_set(mu, cov, w)
    target, tinv, tln = gauss_set(0.5 * tmu, tcov, tw)
    res = b.importance_weights(x, prop, target, want_out=True, want_log_target=True)
    torch.cuda.synchronize()
    assert b._ws[key].data_ptr() == raw.data_ptr(), "the backend replaced the workspace"
    assert bool((raw[need:] == 0x5a).all()), "bytes behind the workspace were written"
    rep = b.maha_gemm_report(N, K, D)
    assert rep is not None and rep["refused"] == 0
    sub = slice(0, N, max(N // 3000, 1))
    logq, _ = orc.mixture_multi_evaluate(0, x[sub], w, mu, inv, ln)
    logp, _ = orc.mixture_multi_evaluate(0, x[sub], tw, 0.5 * tmu, tinv, tln)
    assert_rel(b.tohost(res["out"])[sub], logq, what="log q")
    assert_rel(b.tohost(res["log_target"])[sub], logp, what="log P")
    assert_rel(b.tohost(res["weights"])[sub], orc.is_weights(logp, logq), what="weights")


@pytest.mark.parametrize("D", [17, 20, 21, 24, 31, 32, 33, 37, 40, 41, 45, 48, 49, 57, 64])
@pytest.mark.parametrize("cond", [1e2, 1e4, 1e6])
def test_guard_price_covers_ill_conditioned_covariances_in_every_compiled_dimension(be, orc, small, D, cond):
    """advice r4: eps_g is a probabilistic constant (sqrt(n) u growth), so it is held against the cases that stress it --
    covariances of condition number up to 1e6 with random orientations, means off the centre, every compiled dimension and
    real dimensions below the padded one -- with the tolerance opened wide so that the FORM runs on all of them: its
    difference to the exact kernel must stay below 0.75 of the price eps_g (Theta-sum) it quotes for each sample."""
    K, N = (128 if D <= 20 else 64 if D <= 24 else 32), 1200          # (D <= 24: the form wants four full tiles per pass)
    mu, cov, w = mk(K, D, 700 + D)
    rs = np.random.RandomState(int(D + np.log10(cond)))
    for k in range(K):
        q, _ = np.linalg.qr(rs.normal(size=(D, D)))
        cov[k] = (q * np.logspace(-np.log10(cond) / 2, np.log10(cond) / 2, D)).dot(q.T)
        cov[k] = 0.5 * (cov[k] + cov[k].T)
    x, _ = draw(mu, cov, w, N, 8)
    cs, inv, ln = gauss_set(mu, cov, w)
    be.configure("maha_gemm_tolerance", 1.0)               # price everything in: the form runs on every workgroup
    try:
        got = be.tohost(be.logpdf(x, cs, want_scalars=True)["out"])
        rep = report(be, N, K, D)
    finally:
        be.configure("maha_gemm_tolerance", TOL)
    assert rep["refused"] == 0
    ex = exact(be, lambda: be.tohost(be.logpdf(x, cs, want_scalars=True)["out"]))
    ratio = (np.abs(got - ex) / guard_bound(rep, mu, x)).max()
    assert ratio < 0.75, "D = %d, cond = %g: difference / price = %.3f" % (D, cond, ratio)
    ref, _ = orc.mixture_multi_evaluate(0, x[:300], w, mu, inv, ln)
    assert_rel(ex[:300], ref, rtol=1e-9 * max(1.0, cond / 1e4), what="exact kernel vs oracle")


@pytest.mark.parametrize("D,K,N,student", [(32, 32, 2500, False), (40, 128, 1500, False), (40, 64, 1111, True), (48, 64, 1300, False),
                                           (37, 96, 1290, False), (44, 32, 1100, True), (64, 64, 1200, False), (56, 32, 1100, True),
                                           (24, 64, 1300, False), (20, 128, 1100, True)])
def test_individual_through_the_matrix_product(be, orc, small, D, K, N, student):
    """verdict r4 #5: multi_evaluate(x, individual=...) -- the N x K component log-densities, the reference's own
    intermediate (mixture.pyx:138-151) -- no longer sends the call to the exact engine: the matrix is written from the
    accumulator layout, log w_k (folded into the coefficient image) taken off again.  Against the oracle and the exact
    kernel; ragged N (the last tile is partial); the components' output columns honoured."""
    mu, cov, w = mk(K, D, 800 + D + K)
    x, _ = draw(mu, cov * (1.3 if student else 1.0), w, N, 21)
    if student:
        dofs = np.full(K, 6.0) + 0.25 * (np.arange(K) % 7)
        cs, inv, ln, pf, idf = student_set(mu, cov, w, dofs)
        ref, ref_ind = orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, pf, idf)
    else:
        cs, inv, ln = gauss_set(mu, cov, w)
        ref, ref_ind = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)
    try:
        res = be.logpdf(x, cs, want_individual=True, want_scalars=True)
        rep = report(be, N, K, D)
    finally:
        be.configure("maha_gemm_tolerance", TOL)
    assert rep["refused"] == 0, rep
    got, ind = be.tohost(res["out"]), be.tohost(res["individual"])
    assert_rel(got, ref, what="log q")
    assert_rel(ind, ref_ind, what="individual through the matrix product")
    ex = exact(be, lambda: be.logpdf(x, cs, want_individual=True, want_scalars=True))
    if student:
        assert np.abs(ind - be.tohost(ex["individual"])).max() < TOL
    else:
        assert (np.abs(ind - be.tohost(ex["individual"])).max(axis=1) / guard_bound(rep, mu, x)).max() < 0.75
    # a wider output matrix with permuted columns (a subset's columns in a K_total-wide array)
    if not student:
        from pypmc_amd.backend import ComponentSet
        col = np.random.RandomState(3).permutation(K + 5)[:K]
        cs2 = ComponentSet(0, mu, inv, c0=ln, weight=w, column=col, ld=K + 5)
        import torch
        buf = torch.full((N, K + 5), -7.0, dtype=torch.float64, device=be.device)
        r2 = be.logpdf(x, cs2, individual=buf, want_scalars=True)
        assert report(be, N, K, D)["refused"] == 0
        h = be.tohost(buf)
        assert_rel(h[:, col], ref_ind, what="individual, permuted columns")
        untouched = np.setdiff1d(np.arange(K + 5), col)
        assert (h[:, untouched] == -7.0).all()
        assert_rel(be.tohost(r2["out"]), ref, what="log q")
