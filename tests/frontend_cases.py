"""Front-end test cases, written against the reference's own tests and the
golden vectors.  Every case takes the backend to use: the CPU suite runs them on the oracle-backed
checker (host logic only), the GPU suite on the HIP backend (the product path)."""
import numpy as np
import pytest

from conftest import load_golden

RTOL = 1e-10


def assert_rel(a, b, rtol=RTOL, atol=0.0, what=""):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, "%s: shape %s != %s" % (what, a.shape, b.shape)
    same = a == b
    a, b = np.where(same, 0., a), np.where(same, 0., b)
    err = np.abs(a - b) - atol
    bad = err > rtol * np.abs(b)
    assert not bad.any(), "%s: max rel err %.3e" % (what, np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


# ------------------------------------------------------------------------------------------------
def case_tools_kat(be):
    from pypmc_amd.tools import bilinear_sym, logsumexp, logsumexp2D, chol_inv_det
    from pypmc_amd.tools.convergence import perp, ess
    g = load_golden("kat")
    assert abs(bilinear_sym(g["bil_matrix"], g["bil_vector"], backend=be) - 504.23200000000003) < 1e-9
    assert abs(logsumexp(g["lse_values"], g["lse_weights"], backend=be) - 2.28205254) < 1e-7
    np.testing.assert_allclose(logsumexp2D(g["lse2_values"], g["lse2_weights"], backend=be), g["lse2_ref"], rtol=1e-13)
    assert abs(perp(g["perp_weights"], backend=be) - 0.71922309332486445) < 1e-13
    assert abs(perp(range(5), backend=be) - 0.71922309332486445) < 1e-13
    assert abs(ess(g["perp_weights"], backend=be) - 2. / 3.) < 1e-13
    with pytest.raises(AssertionError, match='negative weight'):
        logsumexp2D(np.zeros((2, 2)), np.array([1., -1.]), backend=be)
    # chol_inv_det: numpy agreement and the reference's error classes (linalg_test.py:17-46)
    for m in (np.array([[1.0, 0.5], [0.5, 1.0]]),
              np.array([[1.0, 0.5, 0.1], [0.5, 2.1, -0.4], [0.1, -0.4, 1.8]])):
        keep = m.copy()
        l, inv, log_det = chol_inv_det(m)
        np.testing.assert_array_equal(m, keep)
        np.testing.assert_allclose(l, np.linalg.cholesky(m))
        np.testing.assert_allclose(inv, np.linalg.inv(m))
        assert abs(log_det - np.log(np.linalg.det(m))) < 1e-13
    with pytest.raises(np.linalg.LinAlgError, match='not symmetric'):
        chol_inv_det(np.array([[0.01, 0.003], [0.001, 0.0025]]))
    for m in (np.array([[0., 0., 0.], [0., 0.0025, 0.], [0., 0., 0.6]]), -np.eye(13)):
        with pytest.raises(np.linalg.LinAlgError, match='not positive definite'):
            chol_inv_det(m)


def case_gauss_student_components(be):
    from pypmc_amd.density.gauss import Gauss
    from pypmc_amd.density.student_t import StudentT
    g = load_golden("kat")
    comp = Gauss(g["gauss_mean"], g["gauss_sigma"], backend=be)
    assert abs(comp.evaluate(g["gauss_point"]) - 1.30077135) < 1e-7
    assert abs(comp.evaluate(g["gauss_point"]) - float(g["gauss_ref"])) < 1e-12
    out1 = np.empty(2)
    out2 = comp.multi_evaluate(np.array([g["gauss_point"]] * 2), out1)
    assert out1 is out2
    np.testing.assert_allclose(out1, [float(g["gauss_ref"])] * 2, rtol=1e-12)
    t = StudentT(g["t_mean"], g["t_sigma"], 5., backend=be)
    np.testing.assert_allclose(t.multi_evaluate(g["t_points"]), g["t_ref"], rtol=1e-12)
    np.testing.assert_allclose(t.multi_evaluate(g["t_points"]), [2.200202941, 2.174596526], atol=1e-9)
    cauchy = StudentT([0.], 1, 1, backend=be)                       # scalar sigma -> 1 x 1
    assert abs(cauchy.evaluate(np.array([3.2])) - (-3.5642087303149452)) < 1e-13
    # failed update leaves the object untouched (gauss_test.py:24-36, student_t_test.py:36-49)
    offdiag = np.array([[0.01, 0.003], [0.003, 0.0025]])
    singular = np.array([[0., 0.], [0., 0.0025]])
    asym = np.array([[0.01, 0.003], [0.001, 0.0025]])
    for obj, extra in ((Gauss(g["gauss_mean"], offdiag, backend=be), ()),
                       (StudentT(g["gauss_mean"], offdiag, 1.5, backend=be), (5.3,))):
        before = obj.evaluate(g["gauss_point"])
        for bad in (singular, asym):
            with pytest.raises(np.linalg.LinAlgError):
                obj.update(g["gauss_point"], bad, *extra)
        np.testing.assert_array_equal(obj.sigma, offdiag)
        np.testing.assert_array_equal(obj.mu, g["gauss_mean"])
        assert obj.dim == 2 and obj.evaluate(g["gauss_point"]) == before
    with pytest.raises(AssertionError, match=r'Dimensions of mean \(2\) and covariance matrix \(3\) do not match!'):
        Gauss(np.ones(2), np.eye(3), backend=be)
    with pytest.raises(AssertionError, match=r'Dimensions of mean \(2\) and covariance matrix \(3\) do not match!'):
        StudentT(np.ones(2), np.eye(3), 4., backend=be)
    # statistical check of propose (gauss_test.py:162-198 style)
    np.random.seed(1)
    s = Gauss([-3., 3.], offdiag, backend=be).propose(20000, np.random.mtrand)
    np.testing.assert_allclose(s.mean(axis=0), [-3., 3.], atol=3e-3)
    np.testing.assert_allclose(np.cov(s.T), offdiag, atol=3e-4)


class DummyComponent(object):
    """evaluates to a constant, proposes a constant (mixture_test.py:15-23)"""
    dim = 1

    def __init__(self, eval_to=42., propose=(0.,)):
        self.eval_to, self.to_propose = eval_to, np.array(propose)

    def evaluate(self, x):
        return self.eval_to

    def multi_evaluate(self, x, out=None):
        if out is None:
            out = np.empty(len(x))
        out[:] = self.eval_to
        return out

    def propose(self, N=1, rng=None):
        return np.array([self.to_propose for _ in range(N)])


def case_mixture_api(be):
    from pypmc_amd.density.mixture import MixtureDensity
    comps = [DummyComponent(10., [-5.]), DummyComponent(42., [5.])]
    mix = MixtureDensity(comps, backend=be)
    assert len(mix) == 2 and mix.normalized()
    np.testing.assert_allclose(mix.weights, 0.5, rtol=1e-15)
    mix.weights[0] = 2
    assert not mix.normalized()
    mix.normalize()
    assert mix.normalized()
    target = 39.69741490700607                    # mixture_test.py:33: log(.9 e^10 + .1 e^42)
    bad = [DummyComponent() for _ in range(5)]
    bad[2].dim = 100
    with pytest.raises(AssertionError):
        MixtureDensity(bad, backend=be)
    mix = MixtureDensity(comps, (.9, .1), backend=be)
    assert abs(mix.evaluate(np.array([1.])) - target) < 1e-12
    samples = np.array([[1.]] * 2)
    individual, out1, out2 = np.zeros((2, 2)), np.zeros(2), np.zeros(2)
    res1 = mix.multi_evaluate(samples, individual=individual)
    res2 = mix.multi_evaluate(samples, individual=individual, out=out1)
    res3 = mix.multi_evaluate(samples, out=out2)
    for other in (res2, res3, out1, out2):
        np.testing.assert_array_equal(res1, other)           # bitwise, whatever else is computed
    np.testing.assert_allclose(res1, target, rtol=1e-14)
    np.testing.assert_array_equal(individual, [[10., 42.]] * 2)
    # error messages (mixture_test.py:106-126)
    s3 = np.array([[1.], [2.], [3.]])
    for exc, pattern, args, kw in (
            ('x.*wrong dim.*', None, (np.array([[1., 1.2], [2., 32.], [2, 3.]]),), dict(individual=np.empty((3, 2)))),
            ('individual.*must.*shape', None, (s3,), dict(individual=np.empty((2, 2)))),
            ('individual.*must.*shape', None, (s3,), dict(individual=np.empty((3, 3)))),
            ('components.*not None.*out.*must be None', None, (s3, np.empty(3)), dict(components=[0])),
            ('out.*must.*len.*3', None, (s3, np.empty(9)), {})):
        with pytest.raises(AssertionError, match=exc):
            mix.multi_evaluate(*args, **kw)
    # prune: range(K) makes the first weight zero (mixture_test.py:70-77)
    mix = MixtureDensity(comps, range(2), backend=be)
    removed = mix.prune()
    assert len(removed) == 1 and removed[0][0] == 0 and removed[0][2] == 0.
    assert len(mix.weights) == 1 and mix.normalized()
    # propose: trace / shuffle semantics (mixture_test.py:128-168)
    mix = MixtureDensity(comps, [0.3, 0.7], backend=be)
    np.random.seed(7)
    s, origin = mix.propose(50, trace=True, shuffle=False)
    np.random.seed(7)
    counts = np.random.mtrand.multinomial(50, mix.weights)
    np.testing.assert_array_equal(origin, np.repeat([0, 1], counts))
    np.testing.assert_array_equal(s[:, 0], np.repeat([-5., 5.], counts))
    with pytest.raises(ValueError, match='shuffle.*trace'):
        mix.propose(5, trace=True, shuffle=True)
    np.random.seed(7)
    shuffled = mix.propose(50)
    assert sorted(shuffled[:, 0]) == sorted(s[:, 0]) and not (shuffled == s).all()


@pytest.mark.parametrize("dummy", [0])
def case_mixture_golden(be, dummy=0):
    from pypmc_amd.density.mixture import (create_gaussian_mixture, create_t_mixture,
                                           recover_gaussian_mixture, recover_t_mixture)
    for tag in ("d2k3", "d5k4", "d20k16", "d1k2", "d7k1"):
        g = load_golden("logpdf_gauss_" + tag)
        mix = create_gaussian_mixture(g["mu"], g["sigma"], g["weights"])
        mix._backend = be
        np.testing.assert_allclose(np.array([c.inv_sigma for c in mix.components]), g["inv_sigma"], rtol=1e-12)
        np.testing.assert_allclose([c.log_normalization for c in mix.components], g["log_norm"], rtol=1e-13)
        N, K = len(g["x"]), len(mix)
        ind = np.zeros((N, K))
        out = mix.multi_evaluate(g["x"], individual=ind)
        assert_rel(out, g["out"], what="out " + tag)
        assert_rel(ind, g["individual"], what="individual " + tag)
        sub = np.zeros((N, K))
        assert mix.multi_evaluate(g["x"], individual=sub, components=list(g["subset"])) is None
        assert_rel(sub, g["individual_subset"], what="subset " + tag)
        assert_rel([mix.evaluate(x) for x in g["x"][:5]], g["evaluate_first5"], what="evaluate")
        res, single = mix.evaluate(g["x"][0], individual=True)
        assert_rel(single, g["individual"][0])
        mix.weights[:] = g["weights_zero0"]
        assert_rel(mix.multi_evaluate(g["x"]), g["out_zero0"], what="zero weight " + tag)
        m, c, w = recover_gaussian_mixture(mix)
        np.testing.assert_array_equal(m, g["mu"])
        np.testing.assert_array_equal(c, g["sigma"])
    for tag in ("d3k2", "d30k8", "d2k3"):
        g = load_golden("logpdf_student_" + tag)
        mix = create_t_mixture(g["mu"], g["sigma"], g["dof"], g["weights"])
        mix._backend = be
        ind = np.zeros((len(g["x"]), len(mix)))
        assert_rel(mix.multi_evaluate(g["x"], individual=ind), g["out"], what="t out " + tag)
        assert_rel(ind, g["individual"], what="t individual " + tag)
        assert recover_t_mixture(mix)[2].tolist() == g["dof"].tolist()
    with pytest.raises(AssertionError, match='Number of means'):
        create_gaussian_mixture(np.zeros((2, 2)), np.zeros((3, 2, 2)))


def case_propose_counts_bit_exact(be):
    from pypmc_amd.density.mixture import create_gaussian_mixture
    g = load_golden("propose_trace")
    mix = create_gaussian_mixture(g["mu"], g["sigma"], g["weights"])
    np.random.seed(int(g["seed"]))
    samples, origin = mix.propose(int(g["N"]), trace=True, shuffle=False)
    np.testing.assert_array_equal(origin, g["origin"])                 # bit-exact indices
    np.testing.assert_array_equal(np.bincount(origin, minlength=len(mix)), g["counts"])   # and counts
    np.testing.assert_allclose(samples.mean(axis=0), g["sample_mean"], rtol=1e-10, atol=1e-14)


class _Replay(object):
    """proposal wrapper whose ``propose`` replays stored samples (FixProposal of
    importance_sampling_test.py:37-45)"""


def _fix_proposal(mix, samples, origin):
    from pypmc_amd.density.mixture import MixtureDensity

    class FixProposal(MixtureDensity):
        def __init__(self, other):
            self.__dict__.update(other.__dict__)

        def propose(self, N=1, rng=None, trace=False, shuffle=True):
            assert N == len(samples)
            return (samples.copy(), origin.copy()) if trace else samples.copy()
    return FixProposal(mix)


def case_importance_sampler(be):
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    from pypmc_amd.sampler.importance_sampling import (ImportanceSampler, calculate_mean,
                                                       calculate_covariance, calculate_expectation)
    from pypmc_amd.tools.convergence import perp, ess, perp_from_sums, ess_from_sums
    from pypmc_amd.tools import indicator
    for tag, student in (("gauss_d2", False), ("student_d5", True)):
        g = load_golden("is_" + tag)
        if student:
            prop = create_t_mixture(g["prop_mu"], g["prop_sigma"], g["prop_dof"], g["prop_weights"])
        else:
            prop = create_gaussian_mixture(g["prop_mu"], g["prop_sigma"], g["prop_weights"])
        prop._backend = be
        target = create_gaussian_mixture(g["target_mu"], g["target_sigma"], g["target_weights"])
        target._backend = be
        N = len(g["samples"])
        for tgt in (target.evaluate, lambda x: target.evaluate(x)):      # batched and per-sample paths
            sampler = ImportanceSampler(tgt, _fix_proposal(prop, g["samples"], g["origin"]),
                                        save_target_values=True, backend=be)
            origin = sampler.run(N, trace_sort=True)
            np.testing.assert_array_equal(origin, g["origin"])
            np.testing.assert_array_equal(sampler.samples[-1], g["samples"])
            assert_rel(sampler.target_values[:][:, 0], g["target_values"], what="target values")
            w = sampler.weights[-1][:, 0]
            assert_rel(w, g["weights"], what="weights " + tag)
            assert abs(perp(w, backend=be) - float(g["perp"])) < 1e-12
            assert abs(ess(w, backend=be) - float(g["ess"])) < 1e-12
            S, L, Q = sampler.last_weight_sums
            assert abs(perp_from_sums(S, L, N) - float(g["perp"])) < 1e-12
            assert abs(ess_from_sums(S, Q, N) - float(g["ess"])) < 1e-12
        # a second run appends a second History run; N = 0 is a no-op returning 0
        assert sampler.run(0) == 0
        sampler.run(N)
        assert len(sampler.samples) == 2 and len(sampler.weights) == 2
        assert sampler.samples[:].shape == (2 * N, g["samples"].shape[1])
        sampler.clear()
        assert len(sampler.samples) == 0 and len(sampler.weights) == 0
        # moments helpers
        w = g["weights"]
        np.testing.assert_allclose(calculate_mean(g["samples"], w, backend=be),
                                   np.average(g["samples"], axis=0, weights=w), rtol=1e-12)
        mean = np.average(g["samples"], axis=0, weights=w)
        ref_cov = w.sum() ** 2 / (w.sum() ** 2 - (w ** 2).sum()) * \
            calculate_expectation(g["samples"], w, lambda x: np.outer(x - mean, x - mean))
        np.testing.assert_allclose(calculate_covariance(g["samples"], w, backend=be), ref_cov, rtol=1e-10, atol=1e-15)
    # indicator: points outside get zero weight and the target is not called there
    g = load_golden("is_gauss_d2")
    prop = create_gaussian_mixture(g["prop_mu"], g["prop_sigma"], g["prop_weights"])
    prop._backend = be
    calls = []

    def target(x):
        calls.append(1)
        return 0.0
    ind = indicator.hyperrectangle([-1e3, -1e3], [0., 1e3])
    sampler = ImportanceSampler(target, _fix_proposal(prop, g["samples"], g["origin"]), indicator=ind, backend=be)
    sampler.run(len(g["samples"]))
    inside = g["samples"][:, 0] <= 0
    assert len(calls) == inside.sum()
    w = sampler.weights[:][:, 0]
    assert (w[~inside] == 0).all() and (w[inside] > 0).all()
    # exp overflow is an OverflowError as with math.exp (importance_sampling.py:207)
    with pytest.raises(OverflowError):
        ImportanceSampler(lambda x: 1e4, _fix_proposal(prop, g["samples"], g["origin"]), backend=be).run(len(g["samples"]))


def case_combine_weights(be):
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import combine_weights
    g = load_golden("combine_weights")
    p1 = create_gaussian_mixture(g["p1_mu"], g["p1_sigma"], g["p1_weights"])
    p2 = create_gaussian_mixture(g["p2_mu"], g["p2_sigma"], g["p2_weights"])
    p1._backend = p2._backend = be
    hist = combine_weights([g["s1"], g["s2"]], [g["w1"], g["w2"]], [p1, p2], backend=be)
    assert len(hist) == 2 and len(hist[0]) == len(g["s1"]) and len(hist[1]) == len(g["s2"])
    assert_rel(hist[:][:, 0], g["combined_log"], what="log branch")
    hist = combine_weights([g["s1"], g["s2"]], [g["w1_zeros"], g["w2"]], [p1, p2], backend=be)
    assert_rel(hist[:][:, 0], g["combined_linear"], what="linear branch")
    with pytest.raises(AssertionError, match='importance-sampling runs but'):
        combine_weights([g["s1"]], [g["w1"], g["w2"]], [p1, p2], backend=be)


def case_history(be):
    from pypmc_amd.tools import History
    h = History(2)
    for i in range(2):
        a = h.append(i + 1)
        a[:] = i + 1
    np.testing.assert_array_equal(h[0], [[1., 1.]])
    np.testing.assert_array_equal(h[1], [[2., 2.], [2., 2.]])
    np.testing.assert_array_equal(h[:], [[1., 1.], [2., 2.], [2., 2.]])
    np.testing.assert_array_equal(h[-1], h[1])
    assert len(h) == 2
    h[0][0, 0] = 7.                                    # views, not copies
    assert h[:][0, 0] == 7.
    with pytest.raises(NotImplementedError):
        h[::2]
    with pytest.raises(AssertionError):
        h.append(0)
    p = History(1, prealloc=10)
    p.append(4)[:] = 1.
    assert p.memleft == 6 and len(p[:]) == 4
    p.clear()
    assert len(p) == 0 and p[:].size == 0


# ------------------------------------------------------------------------------------------------
# tests/test_gpu_devices_frontend.py sets this to a list of device ordinals: the VB cases then run through
# GaussianInference(..., devices=DEVICES) -- one process, the data sharded over those devices by the library
DEVICES = None


def _vb_from_golden(g, be, weighted):
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.variational import GaussianInference
    sw = g["sample_weights"] if weighted else None
    extra = dict(devices=DEVICES) if DEVICES is not None else {}
    if str(g["init_kind"]) == "mixture":
        guess = create_gaussian_mixture(g["init_mu"], g["init_sigma"], g["init_weights"])
        return GaussianInference(g["data"], initial_guess=guess, weights=sw, backend=be, **extra)
    return GaussianInference(g["data"], components=len(g["init_weights"]), initial_guess="first",
                             weights=sw, backend=be, **extra)


def _check_vb_stage(vb, g, stage):
    p = lambda k: g[stage + k]
    for name in ("alpha", "beta", "nu", "m", "W", "log_det_W", "expectation_det_ln_lambda", "expectation_ln_pi"):
        np.testing.assert_allclose(getattr(vb, name), p(name), rtol=1e-10, atol=1e-14, err_msg=stage + name)
    np.testing.assert_allclose(vb.N_comp, p("N_comp"), rtol=1e-10, err_msg=stage + "N_comp")
    np.testing.assert_allclose(vb.inv_N_comp, p("inv_N_comp"), rtol=1e-10)
    np.testing.assert_allclose(vb.x_mean_comp, p("x_mean_comp"), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(vb.S, p("S"), rtol=1e-10, atol=1e-12)
    assert_rel(vb.r, p("r"), what=stage + "r")
    assert_rel(vb.expectation_gauss_exponent, p("expectation_gauss_exponent"), what=stage + "exponent")
    lr, ref = vb.log_rho, p("log_rho")
    assert np.max(np.abs(lr - ref) / np.maximum(np.abs(ref), 1e-3)) < RTOL
    bound, ref_bound = vb.likelihood_bound(), float(g[stage + "bound"])
    assert abs(bound - ref_bound) <= 1e-10 * abs(ref_bound), (bound, ref_bound)


def case_vb_golden(be):
    for tag, weighted in (("d2k3", False), ("d5k4w", True), ("d20k8", False), ("d3k5first", True)):
        g = load_golden("vb_" + tag)
        vb = _vb_from_golden(g, be, weighted)
        assert vb.K == len(g["init_weights"]) and vb.N == len(g["data"]) and vb.dim == g["data"].shape[1]
        for name in ("alpha0", "beta0", "nu0", "m0", "W0"):
            np.testing.assert_allclose(getattr(vb, name), g[name], rtol=1e-14)
        _check_vb_stage(vb, g, "e0_")
        vb.update()
        _check_vb_stage(vb, g, "u1_")
        # run(): iteration count, surviving K and the final posterior
        vb2 = _vb_from_golden(g, be, weighted)
        nit = vb2.run(iterations=25, prune=1., rel_tol=1e-10, abs_tol=1e-5)
        ref_it = int(g["run_iterations"])
        if ref_it < 0:
            assert nit is None, tag
        else:
            # at the fixed point the bound only moves by rounding noise, so the step at which
            # "bound == old_bound or 0 < diff/bound < 1e-10" first holds may shift by a step or two
            assert nit is not None and abs(nit - ref_it) <= 2, (tag, nit, ref_it)
        assert vb2.K == int(g["run_K"])
        post = vb2.posterior2prior()
        for k in ("alpha0", "beta0", "nu0", "m0", "W0"):
            np.testing.assert_allclose(post[k], g["run_post_" + k], rtol=1e-10, atol=1e-11, err_msg=tag + k)
        assert abs(vb2.likelihood_bound() - float(g["run_bound"])) < 1e-10 * abs(float(g["run_bound"]))
        mm = vb2.make_mixture()
        np.testing.assert_allclose(mm.weights, g["run_mix_weights"], rtol=1e-10)
        np.testing.assert_allclose([c.mu for c in mm.components], g["run_mix_mu"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose([c.sigma for c in mm.components], g["run_mix_sigma"], rtol=1e-10, atol=1e-11)
        # the posterior can seed a new object (variational_test.py:365-389)
        vb3 = type(vb2)(g["data"], backend=be, **post)
        assert vb3.K == vb2.K


def case_vb_hand_computed(be):
    """variational_test.py:196-291 -- first E-step and M-step computed by hand."""
    from scipy.special import digamma
    from pypmc_amd.mix_adapt.variational import GaussianInference
    data = np.array([[-2., 3.], [2., 5.], [-1., 7.], [0., 4.], [1., 6.],
                     [2., -3.], [-1., -6.], [1., -4.], [-2., -7.]])
    means = [np.array([0., 5.]), np.array([0., -5.])]
    alpha0, beta0, nu0 = 1e-5, 1e-5, 3
    infer = GaussianInference(data, 2, m=np.vstack((means[0] - 2., means[1] + 2.)),
                              alpha0=alpha0, beta0=beta0, nu0=nu0, backend=be)
    np.testing.assert_allclose(infer.W0[0], np.eye(2))
    np.testing.assert_allclose(infer.inv_W0[0], np.eye(2))
    acc = 1e-15
    for (n, k), val in (((0, 1), 2e5 + 3. * 52), ((1, 0), 2e5 + 3. * 20), ((0, 0), 2e5)):
        assert abs(infer.expectation_gauss_exponent[n, k] - val) <= 10 * acc * val
    e = digamma(3. / 2.) + digamma(1.) + 2. * np.log(2)
    assert abs(infer.expectation_det_ln_lambda[0] - e) < 1e-14
    assert abs(infer.expectation_ln_pi[0] - (digamma(1e-5) - digamma(2e-5))) < 1e-9
    assert abs(infer.r[0, 1] - 1.3336148155022614e-34) < 1e-44
    assert abs(infer.r[0, 0] - 1) < 1e-15
    N_comp = np.array([5., 4.])
    np.testing.assert_allclose(infer.N_comp, N_comp)
    assert abs(infer.inv_N_comp[0] - 1. / 5.) < 1e-16
    x_mean = np.einsum('k,nk,ni->ki', 1. / N_comp, infer.r, data)
    np.testing.assert_allclose(infer.x_mean_comp, x_mean, atol=1e-14)
    S = sum(infer.r[n, 0] * np.outer(data[n] - x_mean[0], data[n] - x_mean[0]) for n in range(9)) / 5.
    np.testing.assert_allclose(infer.S[0], S, rtol=1e-12)
    infer.M_step()
    np.testing.assert_allclose(infer.nu, nu0 + N_comp)
    np.testing.assert_allclose(infer.beta, N_comp + beta0)
    np.testing.assert_allclose(infer.m, np.einsum('k,k,ki->ki', 1. / (N_comp + beta0), N_comp, x_mean), atol=1e-14)
    inv_W = np.eye(2) + N_comp[0] * S + (beta0 * N_comp[0]) / (beta0 + N_comp[0]) * np.outer(x_mean[0], x_mean[0])
    np.testing.assert_allclose(infer.W[0], np.linalg.inv(inv_W))


def case_vb_errors_and_prune(be):
    from pypmc_amd.mix_adapt.variational import (GaussianInference, Wishart_log_B, Wishart_H,
                                                 Wishart_expect_log_lambda, Dirichlet_log_C)
    from pypmc_amd.density.mixture import create_gaussian_mixture
    rs = np.random.RandomState(3)
    data = np.vstack((rs.normal(-3, 1, (60, 2)), rs.normal(3, 1, (60, 2))))
    K, D = 3, 2
    with pytest.raises(ValueError, match='Specify either `components`'):
        GaussianInference(data, backend=be)
    with pytest.raises(TypeError, match='unexpected keyword'):
        GaussianInference(data, K, foo=1, backend=be)
    with pytest.raises(ValueError, match='All elements of alpha0 must exceed'):
        GaussianInference(data, K, alpha0=-1., backend=be)
    with pytest.raises(ValueError, match='len\\(beta0\\)=2 does not match K=3'):
        GaussianInference(data, K, beta0=[1., 1.], backend=be)
    with pytest.raises(ValueError, match='All elements of nu0 must exceed 1'):
        GaussianInference(data, K, nu0=0.5, backend=be)
    with pytest.raises(ValueError, match='Shape of m'):
        GaussianInference(data, K, m=np.zeros((2, 2)), backend=be)
    with pytest.raises(ValueError, match='W0 is neither None'):
        GaussianInference(data, K, W0=np.eye(3), backend=be)
    with pytest.raises(np.linalg.LinAlgError):
        GaussianInference(data, K, W=np.zeros((K, D, D)), backend=be)
    guess = create_gaussian_mixture([[-3., -3.], [3., 3.]], [np.eye(2)] * 2)
    with pytest.raises(ValueError, match='EITHER ``m`` OR ``initial_guess``'):
        GaussianInference(data, initial_guess=guess, m=np.zeros((2, 2)), backend=be)
    with pytest.raises(AssertionError, match='does not match the number of weights'):
        GaussianInference(data, K, weights=np.ones(3), backend=be)
    with pytest.raises(ValueError, match="Can't auto-initialize"):
        GaussianInference(data[:2], 3, backend=be)
    # prune: a far-away third component dies; manual loop == run() (variational_test.py:391-425)
    m = np.array([[-3., -3.], [3., 3.], [40., 40.]])
    vb = GaussianInference(data, 3, m=m, backend=be)
    vb2 = GaussianInference(data, 3, m=m, backend=be)
    nit = vb.run(20, prune=1.)
    for _ in range(nit):
        vb2.update()
        vb2.prune(1.)
    assert vb.K == 2 and vb2.K == 2
    np.testing.assert_allclose(vb.m, vb2.m, rtol=1e-12)
    np.testing.assert_allclose(vb.W, vb2.W, rtol=1e-12)
    assert vb.r.shape == (len(data), 2)
    with pytest.raises(ValueError, match='would remove all components'):
        vb.prune(1e9)
    pp = vb.prior_posterior()
    assert set(pp) == {'alpha0', 'beta0', 'm0', 'nu0', 'W0', 'alpha', 'beta', 'm', 'nu', 'W', 'components'}
    # Wishart / Dirichlet helpers (values: Mathematica, variational_test.py:704-728)
    W = np.array([[1., 0.3], [0.3, 2.]])
    ld = np.log(np.linalg.det(W))
    from scipy.special import gammaln, digamma
    logB = -0.5 * 4 * ld - 4 * np.log(2) - 0.5 * np.log(np.pi) - gammaln(2.) - gammaln(1.5)
    assert abs(Wishart_log_B(2, 4., ld) - logB) < 1e-13
    el = digamma(2.) + digamma(1.5) + 2 * np.log(2.) + ld
    assert abs(Wishart_expect_log_lambda(2, 4., ld) - el) < 1e-13
    assert abs(Wishart_H(2, 4., ld) - (-logB - 0.5 * el + 4.)) < 1e-13
    assert abs(Dirichlet_log_C(np.array([1., 2., 3.])) - np.log(60.)) < 1e-13


def case_vbmerge_golden(be):
    """VBMerge: reduction of an 18-component mixture (reference: variational.pyx:1035-1218)."""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.variational import VBMerge
    g = load_golden("vbmerge")
    big = create_gaussian_mixture(g["in_mu"], g["in_sigma"], g["in_weights"])
    big._backend = be
    merge = VBMerge(big, N=int(g["N"]), components=int(g["components"]), initial_guess='first', backend=be)
    for stage in ("e0_", "u1_"):
        if stage == "u1_":
            merge.update()
        p = lambda k: g[stage + k]
        for name in ("alpha", "beta", "nu", "m", "W", "expectation_det_ln_lambda", "expectation_ln_pi"):
            np.testing.assert_allclose(getattr(merge, name), p(name), rtol=1e-10, atol=1e-14, err_msg=stage + name)
        assert_rel(merge.expectation_gauss_exponent, p("expectation_gauss_exponent"), what=stage + "exponent")
        assert_rel(merge.r, p("r"), rtol=1e-10, what=stage + "r")
        np.testing.assert_allclose(merge.N_comp, p("N_comp"), rtol=1e-10)
        np.testing.assert_allclose(merge.x_mean_comp, p("x_mean_comp"), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(merge.S, p("S"), rtol=1e-10, atol=1e-12)
        b, ref = merge.likelihood_bound(), float(g[stage + "bound"])
        assert abs(b - ref) <= 1e-10 * abs(ref), (stage, b, ref)
    merge2 = VBMerge(big, N=int(g["N"]), components=int(g["components"]), initial_guess='first', backend=be)
    nit = merge2.run(100, prune=1.)
    ref_it = int(g["run_iterations"])
    assert (nit is None) == (ref_it < 0) and (nit is None or abs(nit - ref_it) <= 2)
    assert merge2.K == int(g["run_K"])
    mm = merge2.make_mixture()
    np.testing.assert_allclose(mm.weights, g["run_mix_weights"], rtol=1e-10)
    np.testing.assert_allclose([c.mu for c in mm.components], g["run_mix_mu"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose([c.sigma for c in mm.components], g["run_mix_sigma"], rtol=1e-10, atol=1e-10)
    with pytest.raises(ValueError, match="more output components than input components"):
        VBMerge(big, N=100, components=50, backend=be)


# ------------------------------------------------------------------------------------------------
def _mix_from(g, prefix, student, be):
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    if student:
        mix = create_t_mixture(g[prefix + "mu"], g[prefix + "sigma"], g[prefix + "dof"], g[prefix + "weights"])
    else:
        mix = create_gaussian_mixture(g[prefix + "mu"], g[prefix + "sigma"], g[prefix + "weights"])
    mix._backend = be
    return mix


def _check_mix(res, g, prefix, student, what, live=None):
    K = len(res)
    idx = list(range(K)) if live is None else live
    np.testing.assert_allclose(res.weights, g[prefix + "weights"], rtol=1e-10, atol=1e-15, err_msg=what)
    np.testing.assert_allclose(np.array([c.mu for c in res.components])[idx], g[prefix + "mu"][idx],
                               rtol=1e-10, atol=1e-12, err_msg=what)
    np.testing.assert_allclose(np.array([c.sigma for c in res.components])[idx], g[prefix + "sigma"][idx],
                               rtol=1e-10, atol=1e-12, err_msg=what)
    if student:
        np.testing.assert_allclose(np.array([c.dof for c in res.components])[idx], g[prefix + "dof"][idx],
                                   rtol=1e-10, err_msg=what)


def case_gaussian_pmc_golden(be):
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc, PMC
    for tag in ("d2k3", "d5k4"):
        g = load_golden("pmc_gauss_" + tag)
        x, iw, latent = g["samples"], g["weights"], g["latent"]
        N = len(x)
        prop = _mix_from(g, "in_", False, be)
        cases = dict(rb_w=dict(weights=iw), rb_u=dict(),
                     rb_w_latent_min=dict(weights=iw, latent=latent, rb=True, mincount=int(0.25 * N)),
                     nrb_w=dict(weights=iw, latent=latent, rb=False),
                     nrb_u=dict(latent=latent, rb=False))
        for cname, kw in cases.items():
            before = prop.weights.copy()
            res = gaussian_pmc(x, prop, backend=be, **kw)
            np.testing.assert_array_equal(prop.weights, before)           # copy=True leaves the input alone
            _check_mix(res, g, cname + "_", False, tag + " " + cname)
        # dead component: weight stays zero, the others are adapted
        dead = _mix_from(g, "in_", False, be)
        dead.weights[1] = 0.
        dead.normalize()
        np.testing.assert_allclose(dead.weights, g["dead_in_weights"])
        res = gaussian_pmc(x, dead, weights=iw, backend=be)
        K = len(res)
        _check_mix(res, g, "dead_rb_w_", False, tag + " dead", live=[k for k in range(K) if k != 1])
        assert res.weights[1] == 0.
        # copy=False adapts in place
        inplace = _mix_from(g, "in_", False, be)
        assert gaussian_pmc(x, inplace, weights=iw, copy=False, backend=be) is inplace
        _check_mix(inplace, g, "rb_w_", False, tag + " inplace")
        # PMC driver
        pmc = PMC(x, prop, weights=iw, latent=latent, rb=True, backend=be)
        assert abs(pmc.log_likelihood() - float(g["pmcrun_ll0"])) < 1e-10 * abs(float(g["pmcrun_ll0"]))
        nit = pmc.run(iterations=5, prune=0.)
        assert (-1 if nit is None else nit) == int(g["pmcrun_iterations"])
        assert abs(pmc.log_likelihood() - float(g["pmcrun_ll"])) < 1e-10 * abs(float(g["pmcrun_ll"]))
        _check_mix(pmc.density, g, "pmcrun_", False, tag + " PMC.run")


def case_pmc_errors_and_fallback(be):
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc, student_t_pmc, PMC
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture, MixtureDensity
    from pypmc_amd.density.gauss import Gauss
    from pypmc_amd.density.student_t import StudentT
    g = load_golden("pmc_gauss_d2k3")
    x, iw, latent = g["samples"], g["weights"], g["latent"]
    prop = _mix_from(g, "in_", False, be)
    with pytest.raises(ValueError, match='`mincount` must be 0'):
        gaussian_pmc(x, prop, mincount=10, backend=be)
    with pytest.raises(ValueError, match='`rb` must be True'):
        gaussian_pmc(x, prop, rb=False, backend=be)
    with pytest.raises(AssertionError, match='Number of weights'):
        gaussian_pmc(x, prop, weights=iw[:-1], backend=be)
    with pytest.raises(AssertionError, match='one-dimensional'):
        gaussian_pmc(x, prop, weights=iw[:, None], backend=be)
    with pytest.raises(TypeError):
        PMC(x, "not a mixture", backend=be)
    mixed = MixtureDensity([Gauss([0., 0.], np.eye(2)), StudentT([0., 0.], np.eye(2), 3.)], backend=be)
    with pytest.raises(TypeError):
        PMC(x, mixed, backend=be)
    with pytest.raises(ValueError):
        PMC(x, prop, rb=False, backend=be)
    # invalid covariance: all weight on ONE sample for component 0 -> singular -> old parameters
    # are restored, weight zero, others renormalised (pmc_test.py "invalid cov")
    few = iw.copy()
    idx0 = np.where(latent == 0)[0]
    few[idx0] = 0.
    few[idx0[0]] = 1.
    res = gaussian_pmc(x, prop, weights=few, latent=latent, rb=False, backend=be)
    assert res.weights[0] == 0.
    np.testing.assert_array_equal(res.components[0].mu, prop.components[0].mu)
    np.testing.assert_array_equal(res.components[0].sigma, prop.components[0].sigma)
    assert res.normalized() and (res.weights[1:] > 0).all()
    # mincount prunes (the reference's iterate-while-removing quirk included): counts of
    # [big, small, small, big] with mincount above the two small ones removes only the first small
    mu = np.array([[0., 0.], [5., 5.], [-5., 5.], [5., -5.]])
    mix = create_gaussian_mixture(mu, [np.eye(2)] * 4)
    mix._backend = be
    rs = np.random.RandomState(0)
    lat = np.repeat([0, 1, 2, 3], [400, 5, 5, 400])
    xs = mu[lat] + rs.normal(size=(len(lat), 2))
    res = gaussian_pmc(xs, mix, latent=lat, rb=True, mincount=20, backend=be)
    assert res.weights[1] == 0. and res.weights[2] != 0.
    assert abs(res.weights.sum() - 1) < 1e-12


def case_student_t_pmc_golden(be):
    from pypmc_amd.mix_adapt.pmc import student_t_pmc, PMC
    for tag in ("d2k3", "d4k3"):
        g = load_golden("pmc_student_" + tag)
        x, iw, latent = g["samples"], g["weights"], g["latent"]
        prop = _mix_from(g, "in_", True, be)
        cases = dict(rb_w_dof=dict(weights=iw, dof_solver_steps=100),
                     rb_w_nodof=dict(weights=iw, dof_solver_steps=0),
                     rb_u_dof=dict(dof_solver_steps=100),
                     nrb_w_dof=dict(weights=iw, latent=latent, rb=False, dof_solver_steps=100),
                     nrb_u_nodof=dict(latent=latent, rb=False, dof_solver_steps=0),
                     rb_w_clamp=dict(weights=iw, dof_solver_steps=100, mindof=5., maxdof=5.5))
        for cname, kw in cases.items():
            res = student_t_pmc(x, prop, backend=be, **kw)
            _check_mix(res, g, cname + "_", True, tag + " " + cname)
        pmc = PMC(x, prop, weights=iw, backend=be, dof_solver_steps=0)
        assert pmc.pmc is student_t_pmc
        l0 = pmc.log_likelihood()
        pmc.run(2)
        assert pmc.log_likelihood() >= l0


def case_example_pmc(be):
    """BASELINE config 1: the reference's examples/pmc.py (:15-73), seeded, step by step."""
    from pypmc_amd.density.gauss import Gauss
    from pypmc_amd.density.mixture import MixtureDensity, create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import ImportanceSampler
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    g = load_golden("example_pmc")
    target = create_gaussian_mixture(g["target_means"], g["target_covs"], g["target_weights"])
    target._backend = be
    initial = MixtureDensity([Gauss(m, np.eye(2)) for m in g["prop_means"]], backend=be)
    np.random.seed(int(g["seed"]))
    sampler = ImportanceSampler(target.evaluate, initial, backend=be)
    for i in range(int(g["steps"])):
        origin = sampler.run(int(g["n_per_step"]), trace_sort=True)
        np.testing.assert_array_equal(origin, g["origin_%d" % i], err_msg="origin step %d" % i)   # bit-exact
        samples, weights = sampler.samples[-1], sampler.weights[-1][:, 0]
        if i == 0:
            np.testing.assert_allclose(samples, g["samples_0"], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(weights, g["weights_%d" % i], rtol=1e-10, atol=1e-300, err_msg="weights %d" % i)
        gaussian_pmc(samples, sampler.proposal, weights, origin, mincount=20, rb=True, copy=False, backend=be)
        np.testing.assert_allclose(sampler.proposal.weights, g["prop_weights_%d" % i], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose([c.mu for c in sampler.proposal.components], g["prop_mu_%d" % i],
                                   rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose([c.sigma for c in sampler.proposal.components], g["prop_sigma_%d" % i],
                                   rtol=1e-10, atol=1e-12)
    # the adapted proposal has found both modes
    w = sampler.proposal.weights
    assert abs(w[0] - 0.3) < 0.05 and abs(w[1] - 0.7) < 0.05 and w[2] < 0.05


def case_reference_known_answers(be):
    """Known-answer values held by the reference's own tests (Mathematica / MCMC numbers quoted in
    pypmc/mix_adapt/variational_test.py:672-728)."""
    from pypmc_amd.density.gauss import Gauss
    from pypmc_amd.density.mixture import MixtureDensity
    from pypmc_amd.mix_adapt.variational import VBMerge, Wishart_log_B, Wishart_H, Wishart_expect_log_lambda
    W = np.array([[1, 0.3], [0.3, 11.2]])
    nu = 8.3
    ld = np.log(np.linalg.det(W))
    assert abs(Wishart_log_B(3, 6, 0.0) - np.log(0.00013192862453429398)) < 1e-7        # :710-717
    assert abs(Wishart_log_B(2, nu, ld) - (-19.6714760251454)) < 1e-7                    # :719-720
    assert abs(Wishart_H(2, nu, ld) - 11.4262373965875) < 1e-7                           # :722-725
    assert abs(Wishart_expect_log_lambda(2, nu, ld) - 6.24348627492751) < 1e-7           # :727-728
    # weighted moments with hand-computed answers (pypmc/sampler/importance_sampling_test.py:176-208)
    from pypmc_amd.sampler.importance_sampling import calculate_mean, calculate_covariance, calculate_expectation
    samples = np.array([[0., 4.5], [4., 5.5], [2., 5.]])
    wts = np.array([1., 2., 5.])
    np.testing.assert_allclose(calculate_expectation(samples, wts, lambda x: x), [2.25, 5.0625], rtol=0, atol=1e-15)
    np.testing.assert_allclose(calculate_mean(samples, wts, backend=be), [2.25, 5.0625], rtol=0, atol=1e-14)
    np.testing.assert_allclose(calculate_covariance(samples, wts, backend=be),
                               8. / 34. * np.array([[11.5, 2.875], [2.875, 0.71875]]), rtol=0, atol=1e-14)
    for f in (calculate_mean, calculate_covariance):
        with pytest.raises(AssertionError, match="number of samples.*must.*equal.*number of weights"):
            f(samples, [1., 2., 3., 4.], backend=be)
    # VBMerge on two nearly identical components: merges them in two steps (:672-702)
    target_mean = np.array([4.3, 1.1])
    target_sigma = np.array([[0.01, 0.003], [0.003, 0.0025]])
    means = (np.array([4.30733653, 1.10121756]), np.array([4.29948, 1.09937727]))
    cov = (np.array([[0.01382637, 0.00361037], [0.00361037, 0.0043224]]),
           np.array([[0.00969403, 0.00292157], [0.00292157, 0.00247721]]))
    weights = np.array([0.12644431, 0.87355569])
    mix = MixtureDensity([Gauss(m, c) for m, c in zip(means, cov)], weights)
    vb = VBMerge(mix, N=1e4, components=2, m=np.linspace(-1., 1., 4).reshape((2, 2)), backend=be)
    S = np.array([[0.01022336, 0.00301026], [0.00301026, 0.00271089]])
    np.testing.assert_allclose(vb.S[0], S, rtol=1e-5)
    assert vb.run() == 2
    assert vb.K == 1
    res = vb.make_mixture()
    np.testing.assert_allclose(res.components[0].mu, target_mean, rtol=1e-3)
    np.testing.assert_allclose(res.components[0].sigma, target_sigma, rtol=0.15)


def case_device_history(be):
    """DeviceHistory: History's run structure (reference tools/_history.py:7-116) over backend
    storage; indexing gives read-only host copies, ``device()`` the stored views."""
    from pypmc_amd.tools import DeviceHistory
    h = DeviceHistory(2, prealloc=2, backend=be)
    assert len(h) == 0 and h[:].size == 0 and h.device() is None
    for i in range(3):                                 # third append outgrows the preallocation
        a = h.append(i + 1)
        assert tuple(a.shape) == (i + 1, 2)
        a[:] = be.asdevice(np.full((i + 1, 2), i + 1.))
    np.testing.assert_array_equal(h[0], [[1., 1.]])
    np.testing.assert_array_equal(h[1], [[2., 2.], [2., 2.]])
    np.testing.assert_array_equal(h[:], np.repeat([1., 2., 2., 3., 3., 3.], 2).reshape(6, 2))
    np.testing.assert_array_equal(h[1:], h[:][1:])
    np.testing.assert_array_equal(h[-1], be.tohost(h.device(-1)))
    assert len(h) == 3 and h[0] is h[0]                # the host copy is cached
    with pytest.raises(ValueError):
        h[0][0, 0] = 7.                                # host copies are read-only
    h.device(0)[0, 0] = 7.                             # the device view is the store
    h.append(1)[:] = be.asdevice(np.zeros((1, 2)))     # any change of the store drops the host cache
    assert h[0][0, 0] == 7. and len(h) == 4
    with pytest.raises(NotImplementedError):
        h[::2]
    with pytest.raises(AssertionError):
        h.append(0)
    h.clear()
    assert len(h) == 0 and h.memleft == 2


def case_combine_weights_device_inputs(be):
    """combine_weights on device-resident runs returns a DeviceHistory with the same numbers."""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.sampler.importance_sampling import combine_weights
    from pypmc_amd.tools import DeviceHistory, History
    g = load_golden("combine_weights")
    p1 = create_gaussian_mixture(g["p1_mu"], g["p1_sigma"], g["p1_weights"])
    p2 = create_gaussian_mixture(g["p2_mu"], g["p2_sigma"], g["p2_weights"])
    p1._backend = p2._backend = be
    hist = combine_weights([g["s1"], g["s2"]], [g["w1"], g["w2"]], [p1, p2], backend=be)
    assert isinstance(hist, History)
    if type(be).__name__ != "HipBackend":
        return                                         # numpy "device" arrays are host inputs
    dev = [be.asdevice(g[k]) for k in ("s1", "s2", "w1", "w2", "w1_zeros")]
    hist = combine_weights(dev[:2], dev[2:4], [p1, p2], backend=be)
    assert isinstance(hist, DeviceHistory) and len(hist) == 2
    assert_rel(hist[:][:, 0], g["combined_log"], what="log branch, device inputs")
    assert_rel(be.tohost(hist.device(1))[:, 0], g["combined_log"][len(g["s1"]):], what="device view")
    hist = combine_weights(dev[:2], [dev[4], g["w2"]], [p1, p2], backend=be)      # mixed host / device
    assert_rel(hist[:][:, 0], g["combined_linear"], what="linear branch, device inputs")


def case_local_densities(be):
    """LocalGauss / LocalStudentT (gauss.pyx:12-67, student_t.pyx:13-55): known answers of the reference's tests
    (gauss_test.py:43-55, student_t_test.py:60-79), the LinAlgError contract and the generator call order."""
    from pypmc_amd.density.gauss import LocalGauss
    from pypmc_amd.density.student_t import LocalStudentT
    sigma = np.array([[0.01, 0.003], [0.003, 0.0025]])
    x, y = np.array([4.3, 1.1]), np.array([4.35, 1.2])
    g = LocalGauss(sigma, backend=be)
    assert g.symmetric and g.dim == 2
    assert abs(g.evaluate(x, y) - 1.30077135) < 1e-8 and abs(g.evaluate(y, x) - 1.30077135) < 1e-8
    d = x - y
    ref = -np.log(2 * np.pi) - 0.5 * np.log(np.linalg.det(sigma)) - 0.5 * d.dot(np.linalg.inv(sigma)).dot(d)
    assert abs(g.evaluate(x, y) - ref) < 1e-12 * abs(ref)
    for bad in (np.array([[0.0, 0.0], [0.0, 1.0]]), np.array([[0.01, 0.003], [0.001, 0.0025]])):
        with pytest.raises(np.linalg.LinAlgError):
            g.update(bad)
        with pytest.raises(np.linalg.LinAlgError):
            LocalGauss(bad, backend=be)
    np.testing.assert_array_equal(g.sigma, sigma)                    # untouched by the failed updates
    assert abs(g.evaluate(x, y) - ref) < 1e-12 * abs(ref)

    class Rng(object):                                                # records the calls, returns ones
        def __init__(self):
            self.calls = []

        def normal(self, a, b, n):
            self.calls.append(("normal", n))
            return np.ones(n)

        def chisquare(self, dof):
            self.calls.append(("chisquare", dof))
            return dof

    rng = Rng()
    np.testing.assert_allclose(g.propose(y, rng), y + np.linalg.cholesky(sigma).dot(np.ones(2)))
    assert rng.calls == [("normal", 2)]
    t = LocalStudentT(sigma, 5.0, backend=be)
    from scipy.special import gammaln
    maha = d.dot(np.linalg.inv(sigma)).dot(d)
    reft = gammaln(3.5) - gammaln(2.5) - np.log(5. * np.pi) - 0.5 * np.log(np.linalg.det(sigma)) - 3.5 * np.log(1. + maha / 5.)
    assert abs(t.evaluate(x, y) - reft) < 1e-12 * abs(reft) and abs(t.evaluate(y, x) - reft) < 1e-12 * abs(reft)
    rng = Rng()
    np.testing.assert_allclose(t.propose(y, rng), y + np.linalg.cholesky(sigma).dot(np.ones(2)))   # sqrt(5 / 5) = 1
    assert rng.calls == [("normal", 2), ("chisquare", 5.0)]
    with pytest.raises(AssertionError, match="must be greater than zero"):
        LocalStudentT(sigma, -1.0, backend=be)
    # Gauss.propose asks a foreign generator once per sample for `dim` normals (gauss.pyx:159-163)
    from pypmc_amd.density.gauss import Gauss
    rng = Rng()
    out = Gauss(y, sigma, backend=be).propose(3, rng)
    assert out.shape == (3, 2) and rng.calls == [("normal", 2)] * 3


def case_far_start_values(be):
    """Moments are taken in ONE pass about the component's current mean; when the weighted mean turns out far from
    it (start values, a badly placed proposal) mix_adapt repeats the statistics about the mean just found
    (_stats.shift_is_far), so the covariance keeps the accuracy of the reference's two passes
    (variational.pyx:806-932, pmc.pyx:188-222) instead of cancelling (d/sigma)^2 leading parts."""
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    from pypmc_amd.mix_adapt.variational import GaussianInference
    from pypmc_amd.mix_adapt._stats import shift_is_far
    rs = np.random.RandomState(4)
    cov = np.array([[[0.01, 0.003], [0.003, 0.0025]], [[0.1, 0.], [0., 0.02]]])
    mean = np.array([[1., -4.], [-5., 2.]])
    x = np.vstack([rs.multivariate_normal(mean[k], cov[k], size=n) for k, n in ((0, 300), (1, 900))])
    lab = np.repeat([0, 1], [300, 900])

    def two_pass(k, w=None):
        xs = x[lab == k]
        ws = np.ones(len(xs)) if w is None else w[lab == k]
        m = (ws[:, None] * xs).sum(axis=0) / ws.sum()
        d = xs - m
        return m, np.einsum('n,ni,nj->ij', ws, d, d) / ws.sum()

    # VB: start means 300 standard deviations away from the clusters (variational_test.py:391-406 uses +2 / +10)
    vb = GaussianInference(x, 2, m=mean + np.array([[30., -30.], [-30., 30.]]), backend=be)
    for k in range(2):
        m, S = two_pass(k)
        np.testing.assert_allclose(vb.N_comp[k], (300, 900)[k], rtol=1e-12)
        np.testing.assert_allclose(vb.x_mean_comp[k], m, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(vb.S[k], S, rtol=2e-12, atol=1e-16)     # one pass about m_k: 1e-8 at best
    # PMC: a proposal whose components sit 100 sigma off the samples they are responsible for
    iw = rs.uniform(0.5, 1.5, len(x))
    prop = create_gaussian_mixture(mean + np.array([[1., 1.], [-3., 3.]]), 400. * cov, [.5, .5])
    prop._backend = be
    res = gaussian_pmc(x, prop, weights=iw, latent=lab, rb=False, backend=be)
    for k in range(2):
        m, S = two_pass(k, iw)
        np.testing.assert_allclose(res.components[k].mu, m, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(res.components[k].sigma, S, rtol=2e-12, atol=1e-16)
    # the criterion itself: a mean 3 sigma off its shift is not "far", 30 sigma is
    S0 = np.array([10.])
    for off, far in ((3., False), (30., True)):
        assert shift_is_far(S0, np.array([[off * 10., 0.]]), np.array([[[10. * (1. + off * off), 0.], [0., 10.]]])) is far


def case_big_dimension(be):
    """Sample dimensions beyond the per-dimension kernel units (D > 64: the run-time-dimension unit) through the
    public front-end, against closed-form numpy; the limit (1024) is checked where a density is built."""
    from pypmc_amd.density.gauss import Gauss
    from pypmc_amd.density.mixture import create_gaussian_mixture
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    from pypmc_amd.mix_adapt.variational import GaussianInference
    rs = np.random.RandomState(72)
    D, K, N = 72, 3, 700
    mu = rs.normal(0, 2, (K, D))
    A = rs.normal(size=(K, D, D))
    cov = np.einsum('kij,klj->kil', A, A) / D + 0.5 * np.eye(D)
    w = np.array([.5, .3, .2])
    mix = create_gaussian_mixture(mu, cov, w)
    mix._backend = be
    for c in mix.components:
        c._backend = be
    comp = rs.choice(K, N, p=w)
    x = mu[comp] + np.einsum('nij,nj->ni', np.linalg.cholesky(cov)[comp], rs.normal(size=(N, D)))
    d = x[:, None, :] - mu[None]
    maha = np.einsum('nki,kij,nkj->nk', d, np.linalg.inv(cov), d)
    logq_k = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1] - 0.5 * maha
    mx = logq_k.max(axis=1)
    logq = mx + np.log((w * np.exp(logq_k - mx[:, None])).sum(axis=1))
    ind = np.empty((N, K))
    out = mix.multi_evaluate(x, individual=ind)
    assert_rel(ind, logq_k, rtol=1e-10, what="component log-densities, D = 72")
    assert_rel(out, logq, rtol=1e-10, what="mixture log-density, D = 72")
    assert abs(mix.evaluate(x[3]) - logq[3]) < 1e-10 * abs(logq[3])
    # Rao-Blackwellised Gaussian PMC update (pmc.pyx:120-246) in closed form
    iw = rs.uniform(0.2, 2.0, N)
    rho = w * np.exp(logq_k - logq[:, None])
    res = gaussian_pmc(x, mix, weights=iw, backend=be)
    wr = iw[:, None] * rho
    np.testing.assert_allclose(res.weights, wr.sum(axis=0) / iw.sum(), rtol=1e-10)
    for k in range(K):
        m = (wr[:, k, None] * x).sum(axis=0) / wr[:, k].sum()
        dk = x - m
        S = np.einsum('n,ni,nj->ij', wr[:, k], dk, dk) / wr[:, k].sum()
        np.testing.assert_allclose(res.components[k].mu, m, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(res.components[k].sigma, S, rtol=1e-10, atol=1e-11)
    # VB E-step on the same data: N_comp = sum_n r_nk, rows of r sum to one
    vb = GaussianInference(x, initial_guess=mix, backend=be)
    vb.E_step()
    np.testing.assert_allclose(vb.r.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(vb.N_comp, vb.r.sum(axis=0), rtol=1e-10)
    np.testing.assert_allclose(vb.x_mean_comp, (vb.r.T @ x) / vb.N_comp[:, None], rtol=1e-10, atol=1e-12)
    with pytest.raises(ValueError, match="up to 1024"):
        Gauss(np.zeros(1025), np.eye(1025), backend=be)


def case_copies_and_pickles(be):
    """advice r5: the front-end objects hold device state (resident data, library handles) that is neither picklable nor
    to be freed twice -- copy.deepcopy and pickle go through the host state, the copy finds its device state again on
    first use and computes the same numbers"""
    import copy
    import pickle
    import pypmc_amd as pypmc
    from pypmc_amd import backend as backend_module
    old_default = backend_module._default
    backend_module.set_default_backend(be)             # (the objects below name no backend: the default travels with a pickle)
    try:
        rs = np.random.RandomState(3)
        data = np.concatenate([rs.normal(-2, 1, (150, 3)), rs.normal(3, 1, (130, 3))])
        w = rs.uniform(0.5, 1.5, len(data))
        vb = pypmc.mix_adapt.variational.GaussianInference(data, components=3, weights=w)
        vb.update()
        for clone in (copy.deepcopy(vb), pickle.loads(pickle.dumps(vb))):
            assert getattr(clone, "_vb_samples", None) is None and clone._data_dev is None       # nothing of the device travelled
            np.testing.assert_array_equal(clone.N_comp, vb.N_comp)
            a, b = copy.deepcopy(vb), clone
            a.update()
            b.update()
            np.testing.assert_array_equal(a.N_comp, b.N_comp)
            np.testing.assert_array_equal(a.S, b.S)
            assert a.likelihood_bound() == b.likelihood_bound()
            np.testing.assert_array_equal(a.r, b.r)
        # a sampler after a run
        target = pypmc.density.mixture.create_gaussian_mixture(np.array([[0., 0.], [3., 3.]]), np.array([np.eye(2)] * 2))
        prop = pypmc.density.mixture.create_gaussian_mixture(np.array([[1., 1.]]), np.array([4. * np.eye(2)]))
        smp = pypmc.sampler.importance_sampling.ImportanceSampler(target.evaluate, prop, rng=np.random.RandomState(5))
        smp.run(500)
        c1 = copy.deepcopy(smp)
        c2 = pickle.loads(pickle.dumps(smp))
        for c in (c1, c2):
            np.testing.assert_array_equal(c.samples[:], smp.samples[:])
            np.testing.assert_array_equal(c.weights[:], smp.weights[:])
        c2.run(100)
        assert len(c2.samples[:]) == 600 and len(smp.samples[:]) == 500
    finally:
        backend_module.set_default_backend(old_default)


ALL_CASES = [case_copies_and_pickles, case_local_densities, case_far_start_values, case_big_dimension, case_example_pmc, case_vbmerge_golden, case_tools_kat, case_gauss_student_components, case_mixture_api, case_mixture_golden,
             case_propose_counts_bit_exact, case_importance_sampler, case_combine_weights, case_history,
             case_device_history, case_combine_weights_device_inputs, case_reference_known_answers,
             case_vb_golden, case_vb_hand_computed, case_vb_errors_and_prune, case_gaussian_pmc_golden,
             case_pmc_errors_and_fallback, case_student_t_pmc_golden]
