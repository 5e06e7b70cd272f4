"""Device-side MixtureDensity.propose (pmc_propose): component counts / origins bit-exact with the
reference's construction, sample values statistically equivalent ("RNG parity"),
streams reproducible and independent of sharding, and the device-resident PMC iteration equal to
the host-array path on the same samples."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    return HipBackend()


def mk(K, D, seed):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, 3, size=(K, D))
    cov = np.empty((K, D, D))
    for k in range(K):
        A = rs.normal(0, 1, size=(D, D))
        cov[k] = A.dot(A.T) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K)
    return mu, cov, w / w.sum()


@pytest.mark.parametrize("D,K", [(2, 3), (5, 4), (9, 2), (20, 8), (33, 2), (64, 2), (70, 3), (130, 2)])
def test_gauss_propose(be, D, K):
    from pypmc_amd.density.mixture import create_gaussian_mixture
    mu, cov, w = mk(K, D, 40 + D)
    mix = create_gaussian_mixture(mu, cov, w)
    mix._backend = be
    N = 400_000 if D <= 64 else 150_000
    np.random.seed(123)
    x, origin = mix.propose(N, trace=True, shuffle=False, device=True)
    np.random.seed(123)
    counts = np.random.mtrand.multinomial(N, mix.weights)          # mixture.pyx:192
    origin = origin.cpu().numpy()
    np.testing.assert_array_equal(origin, np.repeat(np.arange(K), counts))     # bit-exact
    x = x.cpu().numpy()
    assert x.shape == (N, D) and np.isfinite(x).all()
    for k in range(K):
        xs = x[origin == k]
        n = len(xs)
        sd = np.sqrt(np.diag(cov[k]))
        assert np.all(np.abs(xs.mean(axis=0) - mu[k]) < 6 * sd / np.sqrt(n))
        emp = np.cov(xs.T).reshape(D, D)
        assert np.max(np.abs(emp - cov[k])) < 8 * np.max(np.diag(cov[k])) / np.sqrt(n) + 1e-12
        d = xs - mu[k]
        maha = np.einsum('ni,ij,nj->n', d, np.linalg.inv(cov[k]), d)          # ~ chi^2(D)
        assert abs(maha.mean() - D) < 6 * np.sqrt(2 * D / n)
        assert abs(maha.var() - 2 * D) < 0.15 * 2 * D
    # a third of the samples is beyond 1 sigma in the first whitened coordinate (two-sided)
    z0 = (x[origin == 0] - mu[0]).dot(np.linalg.inv(np.linalg.cholesky(cov[0])).T)[:, 0]
    assert abs((np.abs(z0) > 1).mean() - 0.3173) < 0.01
    # shuffle / no trace: same multiset of rows
    np.random.seed(123)
    xs2 = mix.propose(N, device=True).cpu().numpy()
    assert xs2.shape == (N, D) and not np.array_equal(xs2, x)
    np.testing.assert_allclose(np.sort(xs2[:, 0]), np.sort(x[:, 0]))


@pytest.mark.parametrize("D,dof", [(65, None), (100, None), (257, None), (80, 6.0)])
def test_big_dimension_propose_is_the_affine_map_of_its_normals(be, D, dof):
    """Run-time-dimension unit (D > 64): the kernel writes a sample's normals into its row and transforms them in
    place, 16 coordinates at a time.  With L = I, mu = 0 it returns the normals themselves [times the Student-t
    scale]; the general x must then be mu_k + L_k z to rounding -- also in wavefronts that mix components."""
    K, N = 4, 3000
    mu, cov, w = mk(K, D, 300 + D)
    chol = np.linalg.cholesky(cov)
    counts = np.array([1000, 37, 0, N - 1037])
    dofs = None if dof is None else np.full(K, dof)
    z, o = be.propose(np.zeros((K, D)), np.tile(np.eye(D), (K, 1, 1)), dofs, counts, seed=11)
    x, o2 = be.propose(mu, chol, dofs, counts, seed=11)
    z, x, o = z.cpu().numpy(), x.cpu().numpy(), o.cpu().numpy()
    np.testing.assert_array_equal(o, np.repeat(np.arange(K), counts))
    np.testing.assert_array_equal(o2.cpu().numpy(), o)
    ref = mu[o] + np.einsum('nij,nj->ni', chol[o], z)
    np.testing.assert_allclose(x, ref, rtol=1e-12, atol=1e-12)
    if dof is None:
        assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.var() - 1) < 0.01
        assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.08 and abs(np.corrcoef(z[:, 0], z[:, D - 1])[0, 1]) < 0.08


def test_student_t_propose(be):
    from pypmc_amd.density.mixture import create_t_mixture
    K, D, N = 3, 4, 600_000
    mu, cov, w = mk(K, D, 77)
    dof = np.array([5., 9., 0.8])                  # incl. nu < 2 (Gamma shape < 1 branch)
    mix = create_t_mixture(mu, cov, dof, w)
    mix._backend = be
    np.random.seed(5)
    x, origin = mix.propose(N, trace=True, shuffle=False, device=True)
    x, origin = x.cpu().numpy(), origin.cpu().numpy()
    assert np.isfinite(x).all()
    for k in range(K):
        xs = x[origin == k]
        d = xs - mu[k]
        f = np.einsum('ni,ij,nj->n', d, np.linalg.inv(cov[k]), d) / D          # ~ F(D, nu)
        med = np.median(f)
        from scipy.stats import f as fdist
        assert abs(med - fdist.median(D, dof[k])) < 0.02 * fdist.median(D, dof[k]) + 0.01
        if dof[k] > 4:
            assert abs(f.mean() - dof[k] / (dof[k] - 2)) < 0.05
            np.testing.assert_allclose(np.cov(xs.T), dof[k] / (dof[k] - 2) * cov[k], atol=0.08 * np.max(cov[k]) * dof[k] / (dof[k] - 2))


def test_streams_reproducible_and_shardable(be):
    K, D, N = 3, 6, 100_001
    mu, cov, w = mk(K, D, 9)
    chol = np.linalg.cholesky(cov)
    counts = np.array([40_000, 1, 60_000])
    x1, o1 = be.propose(mu, chol, None, counts, seed=42)
    x2, _ = be.propose(mu, chol, None, counts, seed=42)
    x3, _ = be.propose(mu, chol, None, counts, seed=43)
    assert bool((x1 == x2).all()) and not bool((x1 == x3).all())
    # the tail [s, N) generated by "another rank" with first_sample = s equals the tail of the whole
    s = 40_000 + 1 + 777
    tail_counts = np.array([0, 0, N - s])
    xt, _ = be.propose(mu, chol, None, tail_counts, seed=42, first_sample=s)
    assert bool((xt == x1[s:]).all())


def test_device_resident_pmc_iteration_equals_host_path(be):
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import ImportanceSampler
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    from pypmc_amd.tools.convergence import perp_from_sums
    K, D, N = 4, 5, 200_000
    mu, cov, w = mk(K, D, 3)
    prop = create_gaussian_mixture(mu, cov * 1.5, w)
    rs = np.random.RandomState(1)
    target = create_gaussian_mixture(mu + 0.3 * rs.normal(size=mu.shape), cov, w[::-1])
    prop._backend = target._backend = be
    np.random.seed(11)
    sampler = ImportanceSampler(target.evaluate, prop, backend=be)
    run = sampler.run_device(N, trace_sort=True)
    x, wts, origin = run["samples"], run["weights"], run["origin"]
    assert x.is_cuda and wts.is_cuda and origin.is_cuda
    S, L, Q = run["weight_sums"]
    assert 0.2 < perp_from_sums(S, L, N) <= 1.0
    dev = gaussian_pmc(x, sampler.proposal, wts, origin, mincount=10, backend=be)          # device tensors
    host = gaussian_pmc(x.cpu().numpy(), sampler.proposal, wts.cpu().numpy(), origin.cpu().numpy(),
                        mincount=10, backend=be)                                            # host arrays
    np.testing.assert_array_equal(dev.weights, host.weights)
    for a, b in zip(dev.components, host.components):
        np.testing.assert_array_equal(a.mu, b.mu)
        np.testing.assert_array_equal(a.sigma, b.sigma)
    # the update moved the proposal towards the target
    assert np.abs(np.array([c.mu for c in dev.components]) - np.array([c.mu for c in target.components])).max() < \
        np.abs(mu - np.array([c.mu for c in target.components])).max()


def test_latent_update_block_path_equals_dense_and_is_cheaper(be):
    """rb=False on samples ordered by component takes the per-component block path; it must equal
    the dense one-hot path (same samples shuffled -> dense) and scale with N, not N x K."""
    import time
    import torch
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc, student_t_pmc
    K, D, N = 24, 10, 600_000
    mu, cov, w = mk(K, D, 21)
    for student in (False, True):
        if student:
            prop = create_t_mixture(mu, cov, np.full(K, 6.), w)
            update = lambda *a, **kw: student_t_pmc(*a, dof_solver_steps=20, **kw)
        else:
            prop = create_gaussian_mixture(mu, cov, w)
            update = gaussian_pmc
        prop._backend = be
        np.random.seed(4)
        x, origin = prop.propose(N, trace=True, shuffle=False, device=True)
        wts = torch.rand(N, dtype=torch.float64, device=x.device) + 0.5
        torch.cuda.synchronize()
        t0 = time.time()
        blocks = update(x, prop, wts, origin, rb=False, backend=be)
        torch.cuda.synchronize()
        t_blocks = time.time() - t0
        perm = torch.randperm(N, device=x.device)
        t0 = time.time()
        dense = update(x[perm].contiguous(), prop, wts[perm].contiguous(), origin[perm].contiguous(),
                       rb=False, backend=be)
        torch.cuda.synchronize()
        t_dense = time.time() - t0
        np.testing.assert_allclose(blocks.weights, dense.weights, rtol=1e-12)
        for a, b in zip(blocks.components, dense.components):
            np.testing.assert_allclose(a.mu, b.mu, rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(a.sigma, b.sigma, rtol=1e-9, atol=1e-12)
            if student:
                assert abs(a.dof - b.dof) < 1e-8 * b.dof


def test_vb_on_device_resident_samples(be):
    """GaussianInference accepts device tensors (e.g. the samples/weights of run_device) and gives the
    same posterior as with host arrays."""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.variational import GaussianInference
    K, D, N = 3, 4, 50_000
    mu, cov, w = mk(K, D, 31)
    mix = create_gaussian_mixture(mu, cov, w)
    mix._backend = be
    np.random.seed(2)
    x = mix.propose(N, device=True)
    import torch
    torch.manual_seed(3)        # unseeded weights made this fail once in ~20 runs at rtol 1e-12
    sw = torch.rand(N, dtype=torch.float64, device=x.device) + 0.5
    a = GaussianInference(x, components=5, weights=sw, backend=be)
    b = GaussianInference(x.cpu().numpy(), components=5, weights=sw.cpu().numpy(), backend=be)
    for vb in (a, b):
        vb.run(15, prune=10.)
    # the two normalise the weights by sums taken in different orders (torch's reduction on the device, numpy's
    # pairwise sum on the host): a 1e-16 difference in the inputs, which 15 VB iterations amplify
    assert a.K == b.K
    np.testing.assert_allclose(a.m, b.m, rtol=1e-10)
    np.testing.assert_allclose(a.W, b.W, rtol=1e-10)
    np.testing.assert_allclose(a.alpha, b.alpha, rtol=1e-10)


@pytest.mark.parametrize("D,K", [(4, 3), (20, 12), (40, 24)])
def test_pmc_update_reuses_the_mahalanobis_forms_of_the_weighting_pass(be, D, K):
    """run_device(keep_mahalanobis=True) + gaussian_pmc(mahalanobis=...): the same new proposal as the
    update that evaluates the components again (bitwise from D = 8 on, where both are the two-kernel path; to
    rounding below, where the plain update is the one-kernel E-step), with mincount pruning, and a ValueError
    for a density the values do not belong to."""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import ImportanceSampler
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    mu, cov, w = mk(K, D, 61 + D)
    prop = create_gaussian_mixture(mu, cov * 1.5, w)
    target = create_gaussian_mixture(mu[:3], cov[:3], [0.5, 0.3, 0.2])     # inside the proposal: healthy weights at any D
    prop._backend = target._backend = be
    np.random.seed(11)
    sampler = ImportanceSampler(target.evaluate, prop, backend=be)
    run = sampler.run_device(20_000, trace_sort=True, keep_mahalanobis=True)
    assert run["mahalanobis"] is not None and run["mahalanobis"].N == 20_000
    for kwargs in (dict(), dict(latent=run["origin"], mincount=int(0.8 * 20_000 / K))):      # some components die
        assert float(run["weights"].sum()) > 0
        plain = gaussian_pmc(run["samples"], sampler.proposal, run["weights"], backend=be, **kwargs)
        reuse = gaussian_pmc(run["samples"], sampler.proposal, run["weights"], backend=be,
                             mahalanobis=run["mahalanobis"], **kwargs)
        np.testing.assert_array_equal(plain.weights == 0, reuse.weights == 0)
        check = np.testing.assert_array_equal if D >= 8 else (lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-13))
        check(reuse.weights, plain.weights)
        for a, b in zip(reuse.components, plain.components):
            check(a.mu, b.mu)
            check(a.sigma, b.sigma)
    moved = create_gaussian_mixture(mu + 0.01, cov * 1.5, w)
    with pytest.raises(ValueError):
        gaussian_pmc(run["samples"], moved, run["weights"], backend=be, mahalanobis=run["mahalanobis"])
    with pytest.raises(ValueError):
        gaussian_pmc(run["samples"][:-1], sampler.proposal, run["weights"][:-1], backend=be,
                     mahalanobis=run["mahalanobis"])
    # a host target function takes the other weighting call (pmc_mixture_logpdf_keep)
    np.random.seed(12)
    centre = mu.mean(axis=0)                     # a broad Gaussian over the proposal: healthy weights at D = 40 too
    s2 = ImportanceSampler(lambda x: -0.5 * float(np.dot(x - centre, x - centre)) / 16., prop, backend=be)
    run2 = s2.run_device(3_000, trace_sort=True, keep_mahalanobis=True)
    a = gaussian_pmc(run2["samples"], s2.proposal, run2["weights"], backend=be)
    b = gaussian_pmc(run2["samples"], s2.proposal, run2["weights"], backend=be, mahalanobis=run2["mahalanobis"])
    np.testing.assert_allclose(b.weights, a.weights, rtol=1e-10, atol=1e-13)
    # Student-t proposal: student_t_pmc (means, covariances, degrees of freedom) from the kept forms
    from pypmc_amd.density.mixture import create_t_mixture
    from pypmc_amd.mix_adapt.pmc import student_t_pmc
    tprop = create_t_mixture(mu, cov * 1.5, 4. + np.arange(K) % 3, w)
    tprop._backend = be
    np.random.seed(13)
    s3 = ImportanceSampler(target.evaluate, tprop, backend=be)
    run3 = s3.run_device(20_000, trace_sort=True, keep_mahalanobis=True)
    plain = student_t_pmc(run3["samples"], s3.proposal, run3["weights"], backend=be)
    reuse = student_t_pmc(run3["samples"], s3.proposal, run3["weights"], backend=be, mahalanobis=run3["mahalanobis"])
    # (bitwise from D = 8 on; below, the plain update is the one-kernel Student-t E-step since round 3, which recovers
    # gamma and the dof sums from the parked a_nk: to rounding)
    check(reuse.weights, plain.weights)
    for a_, b_ in zip(reuse.components, plain.components):
        check(a_.mu, b_.mu)
        check(a_.sigma, b_.sigma)
        assert a_.dof == b_.dof if D >= 8 else abs(a_.dof - b_.dof) <= 1e-8 * b_.dof


def test_device_sampler_keeps_its_history_on_the_gpu(be):
    """ImportanceSampler(device=True): run() proposes straight into a DeviceHistory, weights there;
    host indexing gives lazy copies whose weights equal the oracle's on those very samples."""
    from oracle import oracle as orc
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import ImportanceSampler, combine_weights
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    from pypmc_amd.tools import DeviceHistory
    K, D = 3, 4
    mu, cov, w = mk(K, D, 21)
    prop = create_gaussian_mixture(mu, cov * 1.3, w)
    tmu, tcov, tw = mk(2, D, 22)
    target = create_gaussian_mixture(tmu * 0.3, tcov * 2, tw)
    prop._backend = target._backend = be

    def oracle_logpdf(mix, x):
        inv = np.array([c.inv_sigma for c in mix.components])
        ln = np.array([-0.5 * D * np.log(2 * np.pi) - 0.5 * c.log_det_sigma for c in mix.components])
        return orc.mixture_multi_evaluate(0, x, mix.weights, np.array([c.mu for c in mix.components]), inv, ln)[0]

    np.random.seed(5)
    sampler = ImportanceSampler(target.evaluate, prop, backend=be, device=True, save_target_values=True,
                                prealloc=1000)
    assert isinstance(sampler.samples, DeviceHistory)
    assert sampler.run(1000) is None
    np.random.seed(6)
    origin = sampler.run(700, trace_sort=True)                 # outgrows the preallocation
    np.random.seed(6)
    np.testing.assert_array_equal(origin, np.repeat(np.arange(K), np.random.mtrand.multinomial(700, prop.weights)))
    assert len(sampler.samples) == len(sampler.weights) == len(sampler.target_values) == 2
    assert sampler.samples.device(0).is_cuda and tuple(sampler.samples.device(0).shape) == (1000, D)
    assert tuple(sampler.samples[:].shape) == (1700, D) and tuple(sampler.weights[-1].shape) == (700, 1)
    for run in (0, 1):
        x = sampler.samples[run]
        lt, lq = oracle_logpdf(target, x), oracle_logpdf(prop, x)
        np.testing.assert_allclose(sampler.target_values[run][:, 0], lt, rtol=1e-10)
        np.testing.assert_allclose(sampler.weights[run][:, 0], orc.is_weights(lt, lq), rtol=1e-10)
    # a host callable as target: only it sees host copies of the samples
    calls = []

    def py_target(x):
        calls.append(1)
        return -0.5 * float(x.dot(x))
    np.random.seed(7)
    s2 = ImportanceSampler(py_target, prop, backend=be, device=True)
    s2.run(300)
    x = s2.samples[0]
    assert len(calls) == 300
    np.testing.assert_allclose(s2.weights[0][:, 0], np.exp(-0.5 * (x * x).sum(axis=1) - oracle_logpdf(prop, x)),
                               rtol=1e-10)
    # the stored device runs feed combine_weights and the PMC update without leaving the GPU
    comb = combine_weights([sampler.samples.device(0), sampler.samples.device(1)],
                           [sampler.weights.device(0)[:, 0], sampler.weights.device(1)[:, 0]],
                           [prop, prop], backend=be)
    assert isinstance(comb, DeviceHistory)
    # same proposal twice: the deterministic-mixture weight is the ordinary importance weight
    np.testing.assert_allclose(comb[:][:, 0], sampler.weights[:][:, 0], rtol=1e-12)
    new = gaussian_pmc(sampler.samples.device(slice(None)), prop, comb.device(slice(None))[:, 0], backend=be)
    ref = gaussian_pmc(sampler.samples[:], prop, comb[:][:, 0], backend=be)
    # (the weight normalisation is summed on the device for device inputs, by numpy for host inputs: one ulp)
    np.testing.assert_allclose(new.weights, ref.weights, rtol=1e-14)
    sampler.clear()
    assert len(sampler.samples) == 0 and sampler.samples[:].size == 0
    with pytest.raises(ValueError, match='device=True'):
        ImportanceSampler(target.evaluate, prop, backend=be).run_device(10, store=True)


@pytest.mark.parametrize("T,N", [(1, 1), (2, 1000), (5, 100_003)])
def test_combine_weights_kernel_vs_oracle(be, T, N):
    """pmc_combine_weights against the restated reference loops, both branches, plus the
    non-finite counter behind the reference's final assert (importance_sampling.py:310)."""
    from oracle import oracle as orc
    rs = np.random.RandomState(T * 7 + N % 13)
    q = rs.normal(-20, 8, size=(T, N))
    q[0, ::17] = -800.                                        # exp underflows on the linear scale
    counts = rs.randint(1, 5000, size=T).astype(float)
    omega = np.exp(rs.normal(0, 2, size=N))
    n_total = counts.sum()
    for t in range(T):
        for log_scale in (True, False):
            om = omega.copy()
            if not log_scale:
                om[::5] = 0.
            got, flag = be.combine_weights(q, counts, t, om, n_total, log_scale)
            with np.errstate(all='ignore'):
                ref = orc.combine_weights_run(np.ascontiguousarray(q.T), counts, t, om, n_total, log_scale)
            got = got.cpu().numpy()
            fin = np.isfinite(ref)
            np.testing.assert_array_equal(np.isfinite(got), fin)
            np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-10, atol=0)
            assert float(flag.cpu()[0]) == np.count_nonzero(~fin)
    if T > 1:
        qbad = q.copy()
        qbad[:, 3] = -1e4                                     # 0 / 0 on the linear scale
        _, flag = be.combine_weights(qbad, counts, 0, omega, n_total, False)
        assert float(flag.cpu()[0]) >= 1
