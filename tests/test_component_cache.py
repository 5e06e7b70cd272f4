"""density.mixture.component_set keeps the last few ComponentSets (and with them the parameter pack the backend
uploaded): the key must follow every way the reference lets a caller change a mixture."""
import copy

import numpy as np

from pypmc_amd.density.mixture import component_set, create_gaussian_mixture, create_t_mixture


def _mix(K=4, D=3, seed=0, t=False):
    rs = np.random.RandomState(seed)
    mu = rs.normal(size=(K, D))
    cov = np.array([np.eye(D) * (1. + k) for k in range(K)])
    return create_t_mixture(mu, cov, np.full(K, 5.), None) if t else create_gaussian_mixture(mu, cov)


def test_same_parameters_same_set():
    m = _mix()
    a = component_set(m.components, m.weights)
    assert component_set(m.components, m.weights) is a
    assert component_set(copy.deepcopy(m).components, m.weights) is a          # a copy holds the same parameters
    sub = component_set(m.components, m.weights, [1, 2], 4)
    assert sub is not a and sub.K == 2 and sub.ld == 4                         # a subset is another set ...
    assert component_set(m.components, m.weights, [1, 2], 4) is not sub        # ... and is not kept


def test_a_stamp_never_leaves_its_process():
    """advice r2: a component unpickled from another process / from disk must not hit a local component's pack"""
    import pickle
    from pypmc_amd.density.mixture import clear_component_cache
    m = _mix()
    a = component_set(m.components, m.weights)
    blob = pickle.dumps(m.components)
    assert b'_stamp' not in blob
    back = pickle.loads(blob)
    assert all(x._stamp != y._stamp for x, y in zip(back, m.components))
    assert component_set(back, m.weights) is not a                              # same values, but a new state
    clear_component_cache()
    assert component_set(m.components, m.weights) is not a


def test_every_change_makes_a_new_set():
    m = _mix()
    a = component_set(m.components, m.weights)
    m.components[2].update(m.components[2].mu, 2. * m.components[2].sigma)
    b = component_set(m.components, m.weights)
    assert b is not a and np.array_equal(b.precision[2], m.components[2].inv_sigma)
    m.components[1].mu[0] += 1.                                               # in place, as the reference allows
    c = component_set(m.components, m.weights)
    assert c is not b and c.mu[1, 0] == m.components[1].mu[0]
    m.weights[0] = 0.
    m.normalize()
    d = component_set(m.components, m.weights)
    assert d is not c and d.weight[0] == 0.
    removed = m.prune()
    assert len(removed) == 1
    e = component_set(m.components, m.weights)
    assert e is not d and e.K == 3


def test_student_t_dof_is_part_of_the_state():
    m = _mix(t=True)
    a = component_set(m.components, m.weights)
    c = m.components[0]
    c.update(c.mu, c.sigma, 7.)
    b = component_set(m.components, m.weights)
    assert b is not a and b.c3[0] == 7.


def test_batched_update_renews_the_stamps():
    from pypmc_amd.tools._linalg import chol_inv_det_batch
    m = _mix()
    a = component_set(m.components, m.weights)
    sig = np.array([c.sigma * 3. for c in m.components])
    chol, inv, logdet = chol_inv_det_batch(sig)
    for k, c in enumerate(m.components):
        c._assign(c.mu.copy(), sig[k], chol[k], inv[k], float(logdet[k]))
    b = component_set(m.components, m.weights)
    assert b is not a and np.allclose(b.precision, inv)


def test_batched_factorisation_in_one_dimension():
    """D = 1: the transposed copy potri works in must be a copy (a (K, 1, 1) transpose is already contiguous, and
    potri overwrote the factor the determinant is taken from -- found in round 6 by the device-resident M-step)."""
    from pypmc_amd.tools._linalg import chol_inv_det_batch, chol_inv_det
    ms = np.array([[[4.0]], [[0.25]], [[9.0]]])
    lower, inverse, log_det = chol_inv_det_batch(ms, check_symmetric=False)
    for k in range(3):
        l, i, d = chol_inv_det(ms[k])
        assert np.array_equal(lower[k], l) and np.array_equal(inverse[k], i) and log_det[k] == d
    np.testing.assert_allclose(log_det, np.log(ms[:, 0, 0]))


def test_history_append_keeps_the_reference_bookkeeping_without_its_quadratic_copy():
    """tools.History (pypmc/tools/_history.py:60-110): runs, slices and ``memleft`` as the reference's -- whose append copies the
    whole store every time the preallocation is used up; here the store keeps spare rows (2000 runs stay cheap)."""
    import time
    from pypmc_amd.tools._history import History
    h = History(3, prealloc=5)
    rs = np.random.RandomState(0)
    runs = []
    first = h.append(2)
    first[:] = 1.0
    assert h.memleft == 3
    h.append(3)[:] = 2.0
    assert h.memleft == 0
    runs = [np.full((2, 3), 1.0), np.full((3, 3), 2.0)]
    t0 = time.perf_counter()
    for i in range(2000):
        n = int(rs.randint(1, 40))
        a = rs.normal(size=(n, 3))
        h.append(n)[:] = a
        runs.append(a)
        assert h.memleft == 0
    assert time.perf_counter() - t0 < 2.0
    assert len(h) == 2002
    np.testing.assert_array_equal(h[:], np.vstack(runs))
    np.testing.assert_array_equal(h[17], runs[17])
    np.testing.assert_array_equal(h[5:9], np.vstack(runs[5:9]))
    np.testing.assert_array_equal(h[-1], runs[-1])
    h.clear()
    assert len(h) == 0 and h.memleft == 5 and h[:].size == 0


def test_all_degree_of_freedom_conditions_at_once_find_brentqs_roots():
    """mix_adapt.pmc._solve_dofs (student_t_pmc with many components) against the reference's loop -- scipy's brentq on the
    condition of pmc.pyx:478-497 per component, its ValueError handled as pmc.pyx:700-710 does: the same clamps exactly, the
    roots to brentq's own tolerance (xtol = 2e-12) and the conditioning of a root where the condition is flat."""
    from scipy.optimize import brentq
    from scipy.special import digamma
    from pypmc_amd.mix_adapt.pmc import _solve_dofs, _dof_condition, _trigamma
    from scipy.special import polygamma
    x = 10 ** np.random.RandomState(1).uniform(-6, 4, 500)
    assert np.max(np.abs(_trigamma(x) - polygamma(1, x)) / polygamma(1, x)) < 1e-9
    rs = np.random.RandomState(0)
    nu_true = 10 ** rs.uniform(-5.5, 3.5, 300)               # some roots outside [mindof, maxdof]
    const = -(np.log(.5 * nu_true) - digamma(.5 * nu_true))
    mindof, maxdof = 1e-5, 1e3
    for start in (None, nu_true * 1.5, np.full(300, 7.)):
        got = _solve_dofs(const, mindof, maxdof, start)
        ref = np.empty_like(got)
        for i, c in enumerate(const):
            cond = _dof_condition(c)
            try:
                ref[i] = brentq(cond, mindof, maxdof, maxiter=100)
            except ValueError:
                ref[i] = mindof if cond(mindof) < 0. else maxdof
        clamped = (ref == mindof) | (ref == maxdof)
        assert clamped.sum() > 10 and np.array_equal(got[clamped], ref[clamped])
        assert (np.abs(got - ref) <= 4e-12 + 1e-11 * ref).all()
        inside = (nu_true > mindof) & (nu_true < maxdof)
        assert (np.abs(got[inside] - nu_true[inside]) <= 1e-11 * nu_true[inside]).all()
    assert _solve_dofs(np.array([1.0, np.nan]), mindof, maxdof) is None
    # clamp bounds as the golden case uses them
    got = _solve_dofs(const[:20], 5., 5.5)
    assert ((got >= 5.) & (got <= 5.5)).all()


def test_stacked_arrays_of_a_batched_update_are_used_while_they_are_the_components_rows():
    """density.mixture._stacked: the factor / inverse arrays a batched update registered stand in for the gather out of the
    components only while every component's array still IS a row of them and the stamps are those of the registration"""
    from pypmc_amd.density.mixture import register_stacked, _stacked, clear_component_cache
    from pypmc_amd.tools._linalg import chol_inv_det_batch
    m = _mix()
    sig = np.array([c.sigma * 2. for c in m.components])
    chol, inv, logdet = chol_inv_det_batch(sig)
    for k, c in enumerate(m.components):
        c._assign(c.mu.copy(), sig[k].copy(), chol[k], inv[k], float(logdet[k]))
    assert _stacked(m.components, 'inv_sigma') is None                       # (nothing registered)
    register_stacked(m.components, chol, inv)
    assert _stacked(m.components, 'inv_sigma') is inv and _stacked(m.components, 'cholesky_sigma') is chol
    cs = component_set(m.components, m.weights)
    assert cs.precision is inv or np.shares_memory(cs.precision, inv)
    m.components[1].inv_sigma[0, 0] *= 1.5                                    # edited in place: still the row, still the truth
    assert _stacked(m.components, 'inv_sigma')[1, 0, 0] == m.components[1].inv_sigma[0, 0]
    m.components[0].inv_sigma = m.components[0].inv_sigma.copy()             # replaced: no longer the row
    assert _stacked(m.components, 'inv_sigma') is None
    c = m.components[2]
    c.update(c.mu, c.sigma)                                                  # a new stamp
    assert _stacked(m.components, 'cholesky_sigma') is None
    clear_component_cache()
