"""BASELINE.json's configurations at full size on one MI355X, checked through properties that do
not need a full-size CPU run: parity with the oracle on a random subsample of rows, exact
invariants of the algorithm (sum_k r_nk = 1, sum_k N_k = sum_n w_n, additivity of the statistics over
sample blocks = what the multi-GPU all-reduce relies on), closed-form consistency of the fused
reductions, and bitwise run-to-run determinism."""
import numpy as np
import pytest
from scipy.special import digamma, gammaln

pytestmark = pytest.mark.gpu


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


def device_samples(be, mu, cov, w, N, seed, scale=1.0):
    """x ~ mixture, generated on the device"""
    import torch
    dev = be.device
    g = torch.Generator(device=dev).manual_seed(seed)
    K, D = mu.shape
    comp = torch.multinomial(torch.tensor(w, device=dev), N, replacement=True, generator=g)
    comp, _ = torch.sort(comp)
    counts = torch.bincount(comp, minlength=K).tolist()
    x = torch.randn(N, D, dtype=torch.float64, device=dev, generator=g)
    L = torch.tensor(np.linalg.cholesky(cov) * scale, device=dev)
    m = torch.tensor(mu, device=dev)
    start = 0
    for k in range(K):
        seg = x[start:start + counts[k]]
        seg.copy_(seg @ L[k].T + m[k])
        start += counts[k]
    return x[torch.randperm(N, device=dev, generator=g)].contiguous(), comp


def prec(cov):
    inv = np.linalg.inv(cov)
    return 0.5 * (inv + inv.transpose(0, 2, 1))


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def test_cfg2_gauss_mixture_logpdf_1e6(be, orc):
    """MixtureDensity.multi_evaluate: D=20, K=16, N=1e6"""
    from pypmc_amd.backend import ComponentSet
    K, D, N = 16, 20, 1_000_000
    mu, cov, w = mk(K, D, 1)
    x, _ = device_samples(be, mu, cov, w, N, 7)
    inv = prec(cov)
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
    cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
    res = be.logpdf(x, cs, want_out=True, want_individual=True)
    rows = np.random.RandomState(0).choice(N, 3000, replace=False)
    xs = x[rows].cpu().numpy()
    ref_out, ref_ind = orc.mixture_multi_evaluate(0, xs, w, mu, inv, ln)
    assert rel(res["out"][rows].cpu().numpy(), ref_out) < 1e-10
    assert rel(res["individual"][rows].cpu().numpy(), ref_ind) < 1e-10
    # log q(x) >= log(w_k q_k(x)) for every k, with equality never exceeded
    import torch
    lw = torch.tensor(np.log(w), device=be.device)
    assert bool((res["out"][:, None] >= res["individual"] + lw[None, :] - 1e-12).all())
    res2 = be.logpdf(x, cs)
    assert bool((res2["out"] == res["out"]).all())          # bitwise, with or without `individual`


def test_cfg3_student_importance_weights_1e7(be, orc):
    """Student-t mixture (nu=8) D=30, K=32, N=1e7: importance weights + perplexity/ESS"""
    import torch
    from pypmc_amd.backend import ComponentSet
    K, D, N = 32, 30, 10_000_000
    mu, cov, w = mk(K, D, 2)
    dof = np.full(K, 8.)
    inv = prec(cov)
    logdet = np.linalg.slogdet(cov)[1]
    ln = gammaln(.5 * (dof + D)) - gammaln(.5 * dof) - 0.5 * D * np.log(dof * np.pi) - 0.5 * logdet
    cs = ComponentSet(1, mu, inv, c0=ln, c1=-.5 * (dof + D), c2=1. / dof, c3=dof, weight=w)
    x, _ = device_samples(be, mu, cov, w, N, 8, scale=1.2)
    # target: the same mixture with slightly shifted means (Gaussian) -> healthy weights
    tln = -0.5 * D * np.log(2 * np.pi) - 0.5 * logdet
    tmu = mu + 0.05
    target = ComponentSet(0, tmu, inv, c0=tln, weight=w)
    lt = be.logpdf(x, target)["out"]
    res = be.logpdf(x, cs, want_out=True, log_target=lt, want_scalars=True)
    wts, sc = res["weights"], res["scalars"].cpu().numpy()
    rows = np.random.RandomState(1).choice(N, 2000, replace=False)
    xs = x[rows].cpu().numpy()
    ref_q, _ = orc.mixture_multi_evaluate(1, xs, w, mu, inv, ln, -.5 * (dof + D), 1. / dof)
    ref_t, _ = orc.mixture_multi_evaluate(0, xs, w, tmu, inv, tln)
    assert rel(res["out"][rows].cpu().numpy(), ref_q) < 1e-10
    assert rel(wts[rows].cpu().numpy(), orc.is_weights(ref_t, ref_q)) < 1e-10
    # fused reductions == reductions of the weight vector (torch fp64 on the same data)
    S, Q = float(wts.sum()), float((wts * wts).sum())
    L = float((wts * torch.log(wts)).sum())
    assert abs(sc[0] / S - 1) < 1e-12 and abs(sc[2] / Q - 1) < 1e-12 and abs(sc[1] - L) < 1e-9 * abs(L) + 1e-6
    assert sc[4] == 0
    perp = np.exp(-(sc[1] / sc[0] - np.log(sc[0]))) / N
    ess = sc[0] ** 2 / (N * sc[2])
    assert 0 < perp <= 1 and 0 < ess <= 1 and ess <= perp + 1e-12
    sc2 = be.weight_sums(wts).cpu().numpy()
    np.testing.assert_allclose(sc2[:3], sc[:3], rtol=1e-12)


def _vb_set(mu, cov, w, N, D):
    from pypmc_amd.backend import ComponentSet
    K = len(w)
    alpha = w * (K * 1e-5 + N - K) + 1
    beta = 1e-5 + N * w
    nu = D - 1. + 1e-5 + N * w
    W = np.linalg.inv(cov * (nu - D)[:, None, None])
    W = 0.5 * (W + W.transpose(0, 2, 1))
    ln_lambda = sum(digamma(0.5 * (nu + 1. - i)) for i in range(1, D + 1)) + D * np.log(2.) + np.linalg.slogdet(W)[1]
    ln_pi = digamma(alpha) - digamma(alpha.sum())
    return ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi)), \
        (W, beta, nu, ln_pi, ln_lambda)


@pytest.mark.parametrize("K,N", [(64, 10_000_000), (32, 10_000_000)])
def test_cfg4_vb_estep_1e7(be, orc, K, N):
    """GaussianInference E-step: N=1e7, K=64 (and the metric's K=32), D=20"""
    import torch
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    D = 20
    mu, cov, w = mk(K, D, 3)
    x, _ = device_samples(be, mu, cov, w, N, 9)
    cs, (W, beta, nu, ln_pi, ln_lambda) = _vb_set(mu, cov, w, N, D)
    pack = be.pack(cs)
    sw = torch.rand(N, dtype=torch.float64, device=be.device) + 0.5
    sw *= N / sw.sum()
    for weights in (None, sw):
        full = be.estep(x, cs, 0, sample_w=weights, pack=pack)["stats"].clone()
        again = be.estep(x, cs, 0, sample_w=weights, pack=pack)["stats"]
        assert bool((full == again).all())                              # bitwise deterministic
        sc, S0, M1, M2, _, _ = split_stats(full.cpu().numpy(), K, D)
        # sum_k r_nk = 1  =>  sum_k N_k = sum_n w_n = N
        assert abs(S0.sum() / N - 1) < 1e-11
        # additivity over sample blocks (what the all-reduce across GPUs relies on)
        cut = (N // 3 // 64) * 64 + 17
        a = be.estep(x[:cut], cs, 0, sample_w=None if weights is None else weights[:cut], pack=pack)["stats"].clone()
        b = be.estep(x[cut:].contiguous(), cs, 0,
                     sample_w=None if weights is None else weights[cut:].contiguous(), pack=pack)["stats"]
        np.testing.assert_allclose((a + b).cpu().numpy(), full.cpu().numpy(), rtol=1e-9, atol=1e-7)
        # covariance estimates are symmetric positive definite and close to the generating ones
        xbar, S = centred_moments(S0, M1, M2, mu)
        assert np.all(np.linalg.eigvalsh(S) > 0)
        assert np.max(np.abs(xbar - mu)) < 0.05 and np.max(np.abs(S - cov)) < 0.05
    # subsample parity of r / log_rho with the oracle (unweighted)
    n_sub = 640
    rows = np.arange(0, n_sub)
    res = be.estep(x[:n_sub].contiguous(), cs, 0, want_r=True, want_log_rho=True, pack=pack)
    ref = orc.vb_estep(x[:n_sub].cpu().numpy(), None, mu, W, beta, nu, ln_pi, ln_lambda)
    assert rel(res["r"].cpu().numpy(), ref["r"]) < 1e-10
    assert np.max(np.abs(res["r"].cpu().numpy().sum(axis=1) - 1)) < 1e-13


def _moment_errors(xbar, S, ref_xbar, ref_S):
    """errors of means and second moments on the components' own scale: a mean against its standard deviation, S_ij
    against sqrt(S_ii S_jj) -- the reference's normalisation (both are divided by N_k / the weight sum)"""
    sd = np.sqrt(np.einsum('kii->ki', ref_S))
    return (float(np.max(np.abs(xbar - ref_xbar) / sd)),
            float(np.max(np.abs(S - ref_S) / (sd[:, :, None] * sd[:, None, :]))))


@pytest.mark.parametrize("K,spread", [(64, 1.0), (32, 1.0), (64, 0.25)])
def test_cfg4_statistics_vs_oracle_1e6(be, orc, K, spread):
    """The statistics the headline runs -- responsibilities in groups (k_resp_groups) + the common-shift matrix product
    (k_stats_gemm), which only engage from N * ceil(K / 32) >= 524288 on -- against the oracle's loops
    (variational.pyx:699-709, :806-932, :1003-1013) on ALL host cores at N = 1e6: N_k, x-bar, S, E[log q(Z)] at 1e-10.
    spread 0.25: overlapping components (responsibilities far from one-hot)."""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    D, N = 20, 1_000_000
    mu, cov, w = mk(K, D, 3)
    mu = spread * mu
    x, _ = device_samples(be, mu, cov, w, N, 9)
    xh = x.cpu().numpy()
    cs, (W, beta, nu, ln_pi, ln_lambda) = _vb_set(mu, cov, w, N, D)
    sw = np.random.RandomState(4).uniform(0.5, 1.5, N)
    sw *= N / sw.sum()                                   # variational.pyx:86-100
    for weights in (None, sw):
        ref = orc.vb_estep(xh, weights, mu, W, beta, nu, ln_pi, ln_lambda, mt=True)
        if spread < 1:
            assert (ref["r"].max(axis=1) < 0.99).mean() > 0.3
        flat = be.estep(x, cs, 0, sample_w=weights)["stats"].cpu().numpy()
        sc, S0, M1, M2, _, _ = split_stats(flat, K, D)
        assert rel(S0, ref["N_comp"]) < 1e-10
        xbar, S = centred_moments(S0, M1, M2, mu)
        e1, e2 = _moment_errors(xbar, S, ref["x_mean_comp"], ref["S"])
        assert e1 < 1e-10 and e2 < 1e-10, (e1, e2)
        elq = ref["expectation_log_q_Z"]
        assert abs(sc[0] - elq) <= 1e-10 * abs(elq) + 1e-14 * N, (sc[0], elq)
        del ref


@pytest.mark.parametrize("emit", [False, True])
def test_cfg5_update_vs_oracle_2e5(be, orc, emit):
    """alpha / mu / Sigma of configuration 5's Rao-Blackwell update (pmc.pyx:188-222) at N = 2e5, D = 40, K = 128 against
    the oracle on all host cores, at 1e-10 in the reference's normalisation -- through pmc_estep (the responsibilities'
    matrix-product form + the common-shift statistics) and through the emitting weighting pass + pmc_estep_from_u"""
    import torch
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    K, D, N = 128, 40, 200_000
    tmu, tcov, tw = mk(4, D, 11)
    tmu /= 3.0
    which = np.arange(K) % 4
    mu = tmu[which] + np.random.RandomState(5).normal(0, 0.15, (K, D))
    cov = 1.5 * tcov[which]
    w = np.full(K, 1. / K)
    x, _ = device_samples(be, mu, cov, w, N, 10)
    xh = x.cpu().numpy()
    inv = prec(cov)
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
    cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
    rho = orc.rho_rb(0, xh, w, mu, inv, ln, None, None, list(range(K)), mt=True)
    if emit:
        tcs = ComponentSet(0, tmu, prec(tcov), c0=-0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(tcov)[1], weight=tw)
        em = be.importance_weights(x, cs, tcs, emit=True)
        assert em["responsibilities"] is not None and be.maha_gemm_report(N, K, D)["refused"] == 0
        iw = em["weights"].cpu().numpy()
        flat = be.estep_from_u(x, cs, em["responsibilities"])["stats"].cpu().numpy()
    else:
        iw = np.random.RandomState(6).uniform(0.5, 1.5, N)
        flat = be.estep(x, cs, 1, sample_w=iw)["stats"].cpu().numpy()
        assert be.maha_gemm_report(N, K, D)["refused"] == 0
    alpha, o_mu, o_cov = orc.pmc_reductions(xh, rho, None, iw, list(range(K)), mt=True)
    sc, S0, M1, M2, _, _ = split_stats(flat, K, D)
    assert rel(S0, alpha) < 1e-10
    mean, sigma = centred_moments(S0, M1, M2, mu)
    e1, e2 = _moment_errors(mean, sigma, o_mu, o_cov)
    assert e1 < 1e-10 and e2 < 1e-10, (e1, e2)


def test_cfg5_pmc_update_d40_k128(be, orc):
    """PMC Rao-Blackwell update D=40, K=128 (one GPU's share of N=1e8 / 8)"""
    import torch
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments
    K, D, N = 128, 40, 2_000_000
    mu, cov, w = mk(K, D, 5)
    x, _ = device_samples(be, mu, cov, w, N, 10)
    inv = prec(cov)
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
    cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
    iw = torch.rand(N, dtype=torch.float64, device=be.device) + 0.5
    res = be.estep(x, cs, 1, sample_w=iw)
    sc, S0, M1, M2, _, _ = split_stats(res["stats"].cpu().numpy(), K, D)
    # sum_k rho_nk = 1 (up to the +tiny)  =>  sum_k alpha_k = 1
    assert abs(S0.sum() / float(iw.sum()) - 1) < 1e-11
    mean, sigma = centred_moments(S0, M1, M2, mu)
    assert np.all(np.linalg.eigvalsh(sigma) > 0)
    assert np.max(np.abs(mean - mu)) < 0.3 and np.max(np.abs(S0 / float(iw.sum()) - w)) < 5e-3
    # log-likelihood scalar == sum_n w_n log q(x_n) from the log-pdf kernel
    lq = be.logpdf(x, cs)["out"]
    assert abs(sc[3] / float((iw * lq).sum()) - 1) < 1e-12
    # subsample parity of rho with the oracle
    n_sub = 320
    r = be.estep(x[:n_sub].contiguous(), cs, 1, want_r=True)["r"].cpu().numpy()
    ref = orc.rho_rb(0, x[:n_sub].cpu().numpy(), w, mu, inv, ln, None, None, list(range(K)))
    assert rel(r, ref) < 1e-10


def test_cfg5_full_loop_1p25e7_per_gpu(be, orc):
    """BASELINE config 5 at its stated size: one GPU's share (1.25e7) of N=1e8 samples per iteration,
    D=40, K=128, as the full propose -> weight -> Rao-Blackwell update loop with every N-sized array
    resident on the device.  Checked through invariants and oracle parity on a subsample."""
    import torch
    import pypmc_amd as pypmc
    from pypmc_amd.density.mixture import create_gaussian_mixture, component_set
    from pypmc_amd.tools.convergence import perp_from_sums
    K, D, K_T, N = 128, 40, 4, 12_500_000
    tmu, tcov, tw = mk(K_T, D, 11)
    tmu /= 3.0                                           # modes ~1 sigma apart per coordinate
    target = create_gaussian_mixture(tmu, tcov, tw)
    rs = np.random.RandomState(5)
    which = np.arange(K) % K_T
    proposal = create_gaussian_mixture(tmu[which] + rs.normal(0, 0.15, (K, D)), 1.5 * tcov[which])
    sampler = pypmc.sampler.importance_sampling.ImportanceSampler(target.evaluate, proposal,
                                                                  rng=np.random.RandomState(100))
    perps = []
    for it in range(2):
        old = sampler.proposal
        old_set = component_set(old.components, old.weights)
        run = sampler.run_device(N, trace_sort=True, keep_mahalanobis=it == 1)
        x, wts, origin = run["samples"], run["weights"], run["origin"]
        assert tuple(x.shape) == (N, D) and tuple(wts.shape) == (N,) and tuple(origin.shape) == (N,)
        # counts / origins: exactly the host generator's multinomial draw, ordered by component
        assert bool((origin[1:] >= origin[:-1]).all())
        assert int(torch.bincount(origin, minlength=K).sum()) == N
        S, L, Q = run["weight_sums"]
        assert abs(S / float(wts.sum()) - 1) < 1e-12 and abs(Q / float((wts * wts).sum()) - 1) < 1e-12
        perp = perp_from_sums(S, L, N)
        ess = S * S / (N * Q)
        assert 0 < ess <= perp <= 1
        perps.append(perp)
        # subsample parity of log q, log P and the weights with the oracle
        rows = np.random.RandomState(it).choice(N, 256, replace=False)
        xs = x[rows].cpu().numpy()
        ref_q = orc.mixture_multi_evaluate(0, xs, old_set.weight, old_set.mu, old_set.precision, old_set.c0)[0]
        tset = component_set(target.components, target.weights)
        ref_p = orc.mixture_multi_evaluate(0, xs, tset.weight, tset.mu, tset.precision, tset.c0)[0]
        assert rel(wts[rows].cpu().numpy(), orc.is_weights(ref_p, ref_q)) < 1e-10
        # rho of the update on the first rows against the oracle
        n_sub = 192
        r = be.estep(x[:n_sub].contiguous(), old_set, 1, want_r=True)["r"].cpu().numpy()
        ref = orc.rho_rb(0, x[:n_sub].cpu().numpy(), old_set.weight, old_set.mu, old_set.precision, old_set.c0,
                         None, None, list(range(K)))
        assert rel(r, ref) < 1e-10
        old_mu = np.array([c.mu for c in old.components])
        if it == 1:
            # second iteration: the update reuses the component log-densities the weighting pass kept (12.8 GB
            # here) -- and gives what the update that evaluates the proposal again gives
            twice = pypmc.mix_adapt.pmc.gaussian_pmc(x, sampler.proposal, wts, origin, mincount=0, rb=True, copy=True)
            pypmc.mix_adapt.pmc.gaussian_pmc(x, sampler.proposal, wts, origin, mincount=0, rb=True, copy=False,
                                             mahalanobis=run["mahalanobis"])
            # (to rounding: at this shape the update that evaluates again runs its responsibilities in groups of 16
            # with the factors applied by the statistics kernel, k_resp_groups -- one rounding more per pair; the
            # bit-for-bit identity of k_resp and k_resp_tiles is tests/test_gpu_kernels.py::test_estep_from_kept_logpdf)
            np.testing.assert_allclose(sampler.proposal.weights, twice.weights, rtol=1e-13)
            for a_, b_ in zip(sampler.proposal.components, twice.components):
                np.testing.assert_allclose(a_.mu, b_.mu, rtol=1e-12, atol=1e-14)
                np.testing.assert_allclose(a_.sigma, b_.sigma, rtol=1e-11, atol=1e-14)
            del twice
        else:
            pypmc.mix_adapt.pmc.gaussian_pmc(x, sampler.proposal, wts, origin, mincount=0, rb=True, copy=False)
        new = sampler.proposal
        assert new.normalized() and (new.weights > 0).all() and len(new) == K
        for c in new.components:
            assert np.all(np.linalg.eigvalsh(c.sigma) > 0)
        # the adapted means moved towards their target mode
        new_mu = np.array([c.mu for c in new.components])
        assert np.linalg.norm(new_mu - tmu[which]) < np.linalg.norm(old_mu - tmu[which])
        del run, x, wts, origin
    assert perps[1] > perps[0]                           # the adapted proposal is the better one
    be.release()
    torch.cuda.empty_cache()


def test_offsets_beyond_32_bits(be):
    """N x D = 2.4e9 elements (19 GB of samples, 7.7 GB of responsibilities): element and byte
    offsets beyond 2^31 / 2^32 in every kernel of the path."""
    import torch
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats
    D, K, N = 20, 8, 120_000_000
    rs = np.random.RandomState(0)
    mu = rs.normal(0, 3, (K, D))
    inv = np.tile(np.eye(D), (K, 1, 1))
    ln = np.full(K, -0.5 * D * np.log(2 * np.pi))
    w = np.full(K, 1. / K)
    g = torch.Generator(device=be.device).manual_seed(1)
    x = torch.randn(N, D, dtype=torch.float64, device=be.device, generator=g)
    comp = torch.randint(0, K, (N,), device=be.device, generator=g)
    x += torch.tensor(mu, device=be.device)[comp]
    cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
    out = be.logpdf(x, cs)["out"]
    for sl in (slice(0, 100), slice(N // 2, N // 2 + 100), slice(N - 100, N)):
        d = x[sl].cpu().numpy()[:, None, :] - mu[None]
        a = ln[None] - 0.5 * (d ** 2).sum(-1)
        ref = np.log((w * np.exp(a - a.max(1, keepdims=True))).sum(1)) + a.max(1)
        np.testing.assert_allclose(out[sl].cpu().numpy(), ref, rtol=1e-12)
    del out
    S0, M1, M2 = split_stats(be.estep(x, cs, 1)["stats"].cpu().numpy(), K, D)[1:4]
    assert abs(S0.sum() / N - 1) < 1e-11
    counts = torch.bincount(comp, minlength=K).cpu().numpy()
    # components 3 sigma apart in 20 dimensions: rho is one-hot to ~1e-9
    np.testing.assert_allclose(S0, counts, rtol=1e-6)
    assert np.abs(M1 / S0[:, None]).max() < 5e-3                              # x - mu_k averages to 0
    assert np.abs(np.array([np.diag(M2[k]) / S0[k] for k in range(K)]) - 1).max() < 5e-3
    be.release()
    torch.cuda.empty_cache()


def test_offsets_beyond_32_bits_common_shift_statistics(be):
    """the same with K = 32: the statistics run as the component x monomial product (k_stats_gemm) over 2.4e9 sample
    elements and 3.8e9 responsibilities (31 GB) -- against the per-component-shift kernel on the same u"""
    import torch
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd.mix_adapt._stats import split_stats
    D, K, N = 20, 32, 120_000_000
    rs = np.random.RandomState(1)
    mu = rs.normal(0, 3, (K, D))
    inv = np.tile(np.eye(D), (K, 1, 1))
    ln = np.full(K, -0.5 * D * np.log(2 * np.pi))
    w = np.full(K, 1. / K)
    g = torch.Generator(device=be.device).manual_seed(2)
    x = torch.randn(N, D, dtype=torch.float64, device=be.device, generator=g)
    comp = torch.randint(0, K, (N,), device=be.device, generator=g)
    x += torch.tensor(mu, device=be.device)[comp]
    cs = ComponentSet(0, mu, inv, c0=ln, weight=w)
    fast = be.estep(x, cs, 1)["stats"].cpu().numpy().copy()
    be.configure("stats_common_shift_limit", 0.0)
    try:
        slow = be.estep(x, cs, 1)["stats"].cpu().numpy().copy()
    finally:
        be.configure("stats_common_shift_limit", 1000.0)
    assert not np.array_equal(fast, slow)
    a, b = split_stats(fast, K, D), split_stats(slow, K, D)
    counts = torch.bincount(comp, minlength=K).cpu().numpy()
    np.testing.assert_allclose(a[1], counts, rtol=1e-4)         # 32 random means: a few pairs are only ~2 sigma apart
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12)
    for k in range(K):
        assert np.abs(a[2][k] - b[2][k]).max() <= 1e-10 * (np.abs(b[2][k]).max() + np.sqrt(b[1][k] * np.abs(np.diag(b[3][k])).max()))
        assert np.abs(a[3][k] - b[3][k]).max() <= 1e-10 * np.abs(np.diag(b[3][k])).max()
    be.release()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("D,K", [(2, 32), (3, 17), (4, 8), (5, 32), (7, 24), (2, 3)])
def test_fused_estep_many_rounds(be, D, K):
    """The one-kernel E-step where every workgroup loops over many rounds (LDS buffers reused behind barriers):
    2e6 samples against the two-kernel path, VB (weighted) and Gaussian PMC, and bitwise reproducible."""
    import torch
    from pypmc_amd.backend import ComponentSet
    N = 2_000_003
    mu, cov, w = mk(K, D, 40 + D)
    x, _ = device_samples(be, mu, cov, w, N, 11)
    sw = torch.rand(N, dtype=torch.float64, device=be.device) + 0.5
    inv = prec(cov)
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1]
    vb, _ = _vb_set(mu, cov, w, N, D)
    for cs, mode in ((vb, 0), (ComponentSet(0, mu, inv, c0=ln, weight=w), 1)):
        assert be.lib.pmc_estep_is_fused(K, D, cs.kind, mode) == 1
        fused = be.estep(x, cs, mode, sample_w=sw)["stats"].clone()
        again = be.estep(x, cs, mode, sample_w=sw)["stats"]
        assert bool((fused == again).all())
        two = be.estep(x, cs, mode, sample_w=sw, want_r=True)["stats"]        # the two kernels (+ an N x K matrix)
        np.testing.assert_allclose(fused.cpu().numpy(), two.cpu().numpy(), rtol=1e-10, atol=1e-9)
        del two
    be.release()
    torch.cuda.empty_cache()


def test_more_than_2_to_31_samples(be):
    """The sample COUNT beyond 2^31 (the tests above take byte offsets there, not the row index): 2^31 + 70 001
    one-dimensional samples through propose, log-pdf, importance weights, the one-kernel E-step and the two-kernel
    E-step.  Closed forms at both ends and in the middle, the global sums against torch, and additivity of the statistics
    over two blocks that are each on tested ground."""
    import torch
    from pypmc_amd.backend import ComponentSet
    dev = be.device
    D, K = 1, 2
    N = 2 ** 31 + 70_001
    mu = np.array([[-1.5], [2.0]])
    sig = np.array([0.7, 1.3])
    w = np.array([0.3, 0.7])
    counts = np.array([N // 3, N - N // 3], dtype=np.int64)
    x, origin = be.propose(mu, sig.reshape(K, 1, 1), None, counts, seed=5)
    assert x.shape == (N, 1) and origin.shape == (N,)
    edge = int(counts[0])
    assert int(origin[0]) == 0 and int(origin[edge - 1]) == 0 and int(origin[edge]) == 1 and int(origin[N - 1]) == 1
    assert int(origin.sum()) == int(counts[1])                                     # bit-exact origins, all 2^31 of them
    del origin
    for k, sl in ((0, slice(0, edge)), (1, slice(edge, N))):
        seg = x[sl, 0]
        assert abs(float(seg.mean()) - mu[k, 0]) < 6 * sig[k] / np.sqrt(counts[k])
        assert abs(float(seg.var()) / sig[k] ** 2 - 1) < 6 * np.sqrt(2.0 / counts[k])
    assert bool(torch.isfinite(x).all())
    ln = -0.5 * np.log(2 * np.pi) - np.log(sig)
    inv = (1.0 / sig ** 2).reshape(K, 1, 1)
    cs = ComponentSet(0, mu, inv, c0=ln, weight=w)

    def closed_form(xs):
        a = torch.tensor(np.log(w) + ln, device=dev) - 0.5 * (xs - torch.tensor(mu[:, 0], device=dev)) ** 2 * torch.tensor(inv[:, 0, 0], device=dev)
        return torch.logsumexp(a, dim=1)

    res = be.logpdf(x, cs, want_scalars=True)
    out = res["out"]
    worst = 0.0
    step = 2 ** 28
    for b in range(0, N, step):                                  # every row, block by block (the N x 2 temporaries)
        ref = closed_form(x[b:b + step])
        worst = max(worst, float(((out[b:b + step] - ref).abs() / ref.abs().clamp_min(1e-300)).max()))
        del ref
    assert worst < 1e-10, worst
    # importance weights against a one-component target: log w = log p - log q, the fused sums against torch
    tm, ts = 0.5, 2.0
    target = ComponentSet(0, np.array([[tm]]), np.array([[[1 / ts ** 2]]]), c0=np.array([-0.5 * np.log(2 * np.pi) - np.log(ts)]),
                          weight=np.ones(1))
    iw = be.importance_weights(x, cs, target, want_out=True)
    assert float((iw["out"] - out).abs().max()) == 0.0           # the proposal's log-density: the same kernel body
    logp = -0.5 * np.log(2 * np.pi) - np.log(ts) - 0.5 * ((x[:, 0] - tm) / ts) ** 2
    logp -= out
    lw = logp.exp_()
    wts = iw["weights"]
    assert float(((wts - lw).abs() / lw).max()) < 1e-10
    sc = iw["scalars"].cpu().numpy()
    assert abs(sc[0] / float(wts.sum()) - 1) < 1e-11             # sum w: the fused reduction over 2^31 rows
    assert abs(sc[2] / float((wts * wts).sum()) - 1) < 1e-11     # sum w^2
    del wts
    del logp, iw, lw, out, res
    torch.cuda.empty_cache()
    # E-step, Gaussian PMC Rao-Blackwell mode: one kernel at D = 1; additivity over [0, 2^30) and [2^30, N)
    assert be.lib.pmc_estep_is_fused(K, D, cs.kind, 1) == 1
    h = 2 ** 30
    whole = be.estep(x, cs, 1)["stats"].cpu().numpy().copy()
    parts = be.estep(x[:h], cs, 1)["stats"].cpu().numpy() + be.estep(x[h:], cs, 1)["stats"].cpu().numpy()
    from pypmc_amd.mix_adapt._stats import split_stats
    a, b = split_stats(whole, K, D), split_stats(parts, K, D)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-11)
    assert abs(a[1].sum() / N - 1) < 1e-11                       # sum_k sum_n rho_nk = N
    np.testing.assert_allclose(a[2], b[2], rtol=0, atol=1e-11 * N)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-10)
    # ... and the two kernels with the N x K matrix written out (34 GB): rows sum to one at both ends, same statistics
    two = be.estep(x, cs, 1, want_r=True)
    r = two["r"]
    for sl in (slice(0, 1000), slice(N - 1000, N)):
        assert float((r[sl].sum(dim=1) - 1).abs().max()) < 1e-12
    c = split_stats(two["stats"].cpu().numpy(), K, D)
    np.testing.assert_allclose(c[1], a[1], rtol=1e-11)
    np.testing.assert_allclose(c[3], a[3], rtol=1e-10)
    del r, two, x
    be.release()
    torch.cuda.empty_cache()
