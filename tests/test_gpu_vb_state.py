"""The K-sized state of a variational-Bayes fit on the device (round 6: pmc_vbstate.hip, pmc_vb_state in pmc_ctx.hip,
GaussianInference's lazy fields) against the host path -- the numpy / LAPACK / scipy restatement of
pypmc/mix_adapt/variational.pyx:129-136 (M-step), :759-772 / :800-804 (expectations), :194-209 / :948-1034 (bound) that
every earlier round ran and the golden ``vb_*`` fixtures pin.  Tolerances: the E-step's kernels and constants are the same
on both sides (same bits for the same parameters); the M-step's inversions use LAPACK's algorithm, not LAPACK: 1e-11
relative on these well-conditioned matrices; the bound's ln Gamma / psi are the device's (1e-12 of the bound)."""
import copy
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import get_backend
    return get_backend(None)


def _data(N, D, K, seed, spread=6.0):
    rng = np.random.RandomState(seed)
    centres = rng.normal(size=(K, D)) * spread
    scales = 0.5 + rng.uniform(size=(K, D))
    which = rng.randint(K, size=N)
    return centres[which] + rng.normal(size=(N, D)) * scales[which]


def _fit(x, K, device, **kw):
    from pypmc_amd.mix_adapt.variational import GaussianInference
    vb = GaussianInference.__new__(GaussianInference)
    vb.device_update = device
    vb.__init__(x, K, **kw)
    return vb


FIELDS = ("alpha", "beta", "nu", "m", "W", "log_det_W", "N_comp", "x_mean_comp", "S", "expectation_det_ln_lambda",
          "expectation_ln_pi")


def _close(a, b, rtol, what):
    for name in FIELDS:
        x, y = np.asarray(getattr(a, name)), np.asarray(getattr(b, name))
        scale = np.abs(y).max() if y.size else 1.0
        np.testing.assert_allclose(x, y, rtol=rtol, atol=rtol * max(scale, 1e-300), err_msg="%s: %s" % (what, name))


@pytest.mark.parametrize("K,D,N", [(3, 2, 2000), (8, 5, 20000), (16, 20, 30000), (5, 33, 5000), (4, 64, 3000), (40, 10, 50000),
                                   (2, 1, 500)])
def test_one_update_on_the_device_is_the_host_update(be, K, D, N):
    x = _data(N, D, K, 11 * K + D)
    dev, host = _fit(x, K, True), _fit(x, K, False)
    assert dev._state_active() and not host._state_active()
    # the constructor's E-step: the same kernels from the same constants (the psi parts of the expectations are the
    # host's, the pack's c3 uses the host's log(2 pi)): the same bits
    _close(dev, host, 0.0, "constructor")
    assert dev._expectation_log_q_Z == host._expectation_log_q_Z
    assert abs(dev.likelihood_bound() - host.likelihood_bound()) <= 1e-12 * abs(host.likelihood_bound())
    for it in range(3):
        dev.update()
        host.update()
        bd, bh = dev.likelihood_bound(), host.likelihood_bound()
        assert abs(bd - bh) <= 1e-11 * abs(bh), (it, bd, bh)
        _close(dev, host, 1e-10, "update %d" % it)
    W = dev.W
    np.testing.assert_array_equal(W, W.transpose(0, 2, 1))              # symmetric bit for bit
    for a, b in (("_expectation_log_p_X", 1), ("_expectation_log_p_mu_lambda", 4), ("_expectation_log_q_mu_lambda", 7)):
        dev.likelihood_bound()
        host.likelihood_bound()
        assert abs(getattr(dev, a) - getattr(host, a)) <= 1e-10 * max(1.0, abs(getattr(host, a)))


def test_m_step_alone_is_queued_and_read_back(be):
    K, D = 6, 7
    x = _data(8000, D, K, 5)
    dev, host = _fit(x, K, True), _fit(x, K, False)
    dev.M_step()
    host.M_step()
    for name in ("alpha", "beta", "nu", "m", "W", "log_det_W"):
        ref = getattr(host, name)
        np.testing.assert_allclose(getattr(dev, name), ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max(), err_msg=name)
    # r / log_rho still belong to the parameters of the latest E-step (variational.pyx:636-638 keeps them resident)
    np.testing.assert_allclose(dev.r, host.r, rtol=1e-9, atol=1e-300)
    dev.E_step()
    host.E_step()
    _close(dev, host, 1e-10, "after E-step")


def test_a_whole_run_converges_to_the_same_fit(be):
    K, D = 12, 8
    x = _data(40000, D, 5, 77)
    dev, host = _fit(x, K, True), _fit(x, K, False)
    nd, nh = dev.run(200, verbose=False), host.run(200, verbose=False)
    assert dev.K == host.K
    assert (nd is None) == (nh is None) and (nd is None or abs(nd - nh) <= 3), (nd, nh)
    bd, bh = dev.likelihood_bound(), host.likelihood_bound()
    assert abs(bd - bh) <= 1e-8 * abs(bh)
    md, mh = dev.make_mixture(), host.make_mixture()
    assert len(md) == len(mh)
    np.testing.assert_allclose(md.weights, mh.weights, rtol=1e-6)
    for cd, ch in zip(md.components, mh.components):
        np.testing.assert_allclose(cd.mu, ch.mu, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(cd.sigma, ch.sigma, rtol=1e-5, atol=1e-8)


def test_weighted_data_and_a_mixture_as_initial_guess(be):
    from pypmc_amd.density.mixture import create_gaussian_mixture
    K, D, N = 4, 3, 6000
    x = _data(N, D, K, 21)
    rng = np.random.RandomState(2)
    w = rng.uniform(0.1, 2.0, size=N)
    guess = create_gaussian_mixture(rng.normal(size=(K, D)) * 4, np.array([np.eye(D) * 2.0] * K))
    dev, host = _fit(x, 0, True, weights=w, initial_guess=guess), _fit(x, 0, False, weights=w, initial_guess=guess)
    for _ in range(4):
        dev.update()
        host.update()
    _close(dev, host, 1e-10, "weighted")
    assert abs(dev.likelihood_bound() - host.likelihood_bound()) <= 1e-11 * abs(host.likelihood_bound())


def test_assigned_and_edited_fields_reach_the_device(be):
    K, D = 5, 4
    x = _data(5000, D, K, 9)
    dev, host = _fit(x, K, True), _fit(x, K, False)
    dev.update()
    host.update()
    for vb in (dev, host):
        vb.m = vb.m + 0.25                                        # assigned
        vb.beta[1] *= 2.0                                        # edited in place through the attribute
        W = vb.W
        W[2] = W[2] * 1.5                                        # edited in place through a reference the caller keeps
        vb.log_det_W[2] += D * np.log(1.5)
        vb.E_step()
    _close(dev, host, 1e-10, "after edits")
    assert abs(dev.likelihood_bound() - host.likelihood_bound()) <= 1e-11 * abs(host.likelihood_bound())
    # a prior edited in place between two updates
    for vb in (dev, host):
        vb.alpha0[:] = 0.5
        m0 = vb.m0
        vb.update()
        m0[0] += 1.0
        vb.update()
    _close(dev, host, 1e-10, "after prior edits")


def test_prune_and_run_with_the_state(be):
    K, D = 10, 3
    x = _data(20000, D, 3, 4)
    dev, host = _fit(x, K, True), _fit(x, K, False)
    for _ in range(25):
        dev.update()
        host.update()
        dev.prune(300.)
        host.prune(300.)
        assert dev.K == host.K
    assert dev.K < K                                              # (something was pruned on the way)
    _close(dev, host, 1e-8, "after pruning")
    assert dev._state.K == dev.K
    with pytest.raises(ValueError):
        dev.prune(1e300)


def test_copies_and_pickles_carry_the_fields(be):
    K, D = 4, 3
    x = _data(3000, D, K, 8)
    dev = _fit(x, K, True)
    dev.update()
    twin = copy.deepcopy(dev)
    back = pickle.loads(pickle.dumps(dev))
    for other in (twin, back):
        assert other.__dict__.get("_state") is None
        _close(other, dev, 0.0, "copy")
        other.update()
    dev.update()
    _close(twin, dev, 0.0, "copy, one update later")                # (the shifts of the moments travel with the copy)
    _close(back, dev, 0.0, "pickle, one update later")


def test_a_matrix_that_does_not_factorise_is_reported(be):
    K, D = 3, 3
    x = _data(2000, D, K, 3)
    dev = _fit(x, K, True)
    S = dev.S
    S[1] = -np.eye(D) * 1e6                                       # W^-1 of component 1 becomes indefinite
    with pytest.raises(np.linalg.LinAlgError) as e:
        dev.update()
    assert "component 1" in str(e.value)
    dev2 = _fit(x, K, True)
    S = dev2.S
    S[2] = -np.eye(D) * 1e6
    dev2.M_step()                                                 # queued ...
    with pytest.raises(np.linalg.LinAlgError):
        dev2.likelihood_bound()                                   # ... reported by the next call that reads a block


def test_host_switches(be, monkeypatch):
    x = _data(1000, 2, 2, 1)
    monkeypatch.setenv("PMC_VB_DEVICE_STATE", "0")
    vb = _fit(x, 2, True)
    assert not vb._state_active()
    monkeypatch.delenv("PMC_VB_DEVICE_STATE")
    assert _fit(x, 2, True)._state_active()


def test_kernel_level_calls_refuse_bad_arguments(be):
    import ctypes as C
    lib = be.lib
    assert lib.pmc_vb_max_dim() == 64
    assert lib.pmc_vb_small_len(5) == 28 and lib.pmc_vb_small_len(0) < 0
    assert lib.pmc_vb_bound_scratch_len(5) == 58                   # (10 terms per component + the ticket)
    assert lib.pmc_vb_mstep_device(0, 3, None, None, None) < 0
    assert lib.pmc_vb_bound_device(2, 65, None, None, None, None, None) < 0
    st = np.zeros(6)
    assert lib.pmc_vb_mstep_status(3, st.ctypes.data_as(C.POINTER(C.c_double))) == 0
    st[1], st[4] = 3.0, -0.5
    assert lib.pmc_vb_mstep_status(3, st.ctypes.data_as(C.POINTER(C.c_double))) == -2
    from pypmc_amd import _lib
    assert "component 1" in _lib.last_error() and "pivot 2" in _lib.last_error()


def test_the_devices_own_psi_is_an_option(be):
    """``device_psi = True``: the expectations' psi on the device as well -- accurate to 2e-15 (1 + |psi|), not scipy's bits"""
    from pypmc_amd.mix_adapt.variational import GaussianInference
    K, D = 6, 5
    x = _data(6000, D, K, 31)
    own = GaussianInference.__new__(GaussianInference)
    own.device_psi = True
    own.__init__(x, K)
    ref = _fit(x, K, True)
    for _ in range(3):
        own.update()
        ref.update()
    _close(own, ref, 1e-9, "device psi")                           # (three updates apart: the trajectories, not the function)


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_against_the_host_path(be, seed):
    """random K, D, N, weights, priors: two updates and the bound on both paths"""
    rng = np.random.RandomState(1000 + seed)
    D = int(rng.choice([1, 2, 3, 7, 8, 13, 20, 31, 40, 64]))
    K = int(rng.randint(1, 24))
    N = int(rng.randint(max(K, 50), 40000))
    x = _data(N, D, max(1, K // 2), seed, spread=float(rng.uniform(2, 10)))
    kw = dict(alpha0=float(rng.uniform(1e-3, 2)), beta0=float(rng.uniform(1e-3, 2)), nu0=D - 1 + float(rng.uniform(1e-2, 5)),
              W0=np.eye(D) * float(rng.uniform(0.1, 10)), m0=rng.normal(size=D))
    if seed % 3 == 0:
        kw["weights"] = rng.uniform(0.05, 3.0, size=N)
    dev, host = _fit(x, K, True, **kw), _fit(x, K, False, **kw)
    _close(dev, host, 0.0, "constructor")
    for it in range(2):
        dev.update()
        host.update()
        _close(dev, host, 1e-9, "update %d (K=%d D=%d N=%d)" % (it, K, D, N))
        bd, bh = dev.likelihood_bound(), host.likelihood_bound()
        assert abs(bd - bh) <= 1e-10 * abs(bh), (bd, bh)


@pytest.mark.parametrize("K,D,N,clusters,prune", [(10, 3, 20000, 3, 300.), (12, 8, 40000, 5, 1.), (4, 2, 3000, 4, 1.), (6, 20, 30000, 2, 50.),
                                                  (5, 5, 5000, 5, 0.)])
def test_run_inside_the_library_is_the_loop_of_this_file(be, K, D, N, clusters, prune):
    """run() as pmc_vb_state_run (the iterations between two prunings in one call) against the same steps driven from
    variational.py: the same kernels in the same order -- the same iteration count, the same survivors, the same bits"""
    x = _data(N, D, clusters, 17 * K + D)
    lib, loop = _fit(x, K, True), _fit(x, K, True)
    loop.run_in_library = False
    assert lib._run_in_library_ok() and not loop._run_in_library_ok()
    n_lib, n_loop = lib.run(80, prune=prune), loop.run(80, prune=prune)
    assert n_lib == n_loop and lib.K == loop.K
    _close(lib, loop, 0.0, "library loop")
    assert lib.likelihood_bound() == loop.likelihood_bound()
    assert lib._shift_valid == loop._shift_valid
    # and the object goes on as usual afterwards
    lib.update()
    loop.update()
    _close(lib, loop, 0.0, "one more update")


def test_run_inside_the_library_reports_what_the_loop_reports(be):
    K, D = 3, 3
    x = _data(2000, D, K, 3)
    vb = _fit(x, K, True)
    S = vb.S
    S[1] = -np.eye(D) * 1e6                                       # W^-1 of component 1 becomes indefinite in the first M-step
    with pytest.raises(np.linalg.LinAlgError) as e:
        vb.run(5)
    assert "component 1" in str(e.value)
    # an iteration cap of zero, and a cap that is reached
    vb = _fit(x, K, True)
    assert vb.run(0) is None
    assert vb.run(1, rel_tol=0., abs_tol=0.) is None


@pytest.mark.parametrize("K,D", [(64, 20), (1000, 3), (7, 64)])
def test_the_one_launch_bound_has_the_same_bits_every_time(be, K, D):
    """k_vb_bound_terms: the workgroup that draws the last ticket adds the terms up (they travel as agent-scope atomics across
    the eight L2s); the same state must give the same eight numbers call after call"""
    x = _data(max(4 * K, 2000), D, 4, K + D)
    vb = _fit(x, K, True)
    vb.update()
    st = vb._state
    first = st.step(None, bound=True)["bound"].copy()
    assert np.isfinite(first).all()
    for _ in range(400):
        np.testing.assert_array_equal(st.step(None, bound=True)["bound"], first)
    host = _fit(x, K, False)
    host.update()
    assert abs(first[0] - host.likelihood_bound()) <= 1e-11 * abs(first[0])


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
def test_the_state_over_several_devices(be, devices):
    """GaussianInference(devices=[...]): the K-sized state on the group's first device, every device its shard (virtual shards
    of the one GPU here) -- against the same group with the K-sized work on the host (pmc_vb_estep per E-step), and against the
    one-device fit (another partition of the sums: to rounding)"""
    from pypmc_amd.devices import DeviceGroup
    K, D, N = 9, 6, 30011
    x = _data(N, D, 4, 12)
    w = np.random.RandomState(1).uniform(0.2, 2.0, size=N)
    with DeviceGroup(devices) as group:
        dev = _fit(x, K, True, devices=group, weights=w)
        host = _fit(x, K, False, devices=group, weights=w)
        assert dev._state_active() and not host._state_active()
        _close(dev, host, 0.0, "constructor over %d devices" % len(devices))
        for it in range(3):
            dev.update()
            host.update()
            _close(dev, host, 1e-10, "update %d over %d devices" % (it, len(devices)))
            assert abs(dev.likelihood_bound() - host.likelihood_bound()) <= 1e-11 * abs(host.likelihood_bound())
        one = _fit(x, K, True, weights=w)
        for it in range(3):
            one.update()
        _close(dev, one, 1e-9, "several devices against one")
        # run(), pruning included, and the N x K attributes
        n_dev, n_host = dev.run(30, prune=200.), host.run(30, prune=200.)
        assert n_dev == n_host and dev.K == host.K
        _close(dev, host, 1e-8, "run over %d devices" % len(devices))
        np.testing.assert_allclose(dev.r, host.r, rtol=1e-7, atol=1e-300)
        del dev, host
