"""ctypes/numpy face of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module; nothing under ``pypmc_amd/`` does.  See ``pmc_oracle.c`` for the
reference file:line each function restates, and ``tests/test_oracle_golden.py`` for the
golden vectors (generated from the reference itself) that pin it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpmc_oracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_int64)
_sz = C.c_size_t


def build(force=False):
    """Compile libpmc_oracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "pmc_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libpmc_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_bilinear_sym.restype = C.c_double
        _lib.orc_logsumexp.restype = C.c_double
        _lib.orc_perp.restype = C.c_double
        _lib.orc_ess.restype = C.c_double
        _lib.orc_vb_expectation_log_q_Z.restype = C.c_double
        _lib.orc_pmc_log_likelihood.restype = C.c_double
        _lib.orc_is_weights.restype = C.c_size_t
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _d(a):
    """contiguous float64 view/copy + pointer"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _opt(a):
    if a is None:
        return None, None
    return _d(a)


def num_threads():
    return int(lib().orc_num_threads())


# ----------------------------------------------------------------------------- L1 kernels
def bilinear_sym(matrix, vector):
    m, mp = _d(matrix)
    v, vp = _d(vector)
    return lib().orc_bilinear_sym(mp, vp, _sz(len(v)))


def logsumexp(a, weights):
    a, ap = _d(a)
    w, wp = _d(weights)
    return lib().orc_logsumexp(ap, wp, _sz(len(a)))


def logsumexp2D(a, weights):
    a, ap = _d(a)
    w, wp = _d(weights)
    assert (w >= 0.).all(), 'Found negative weight'
    res = np.zeros(len(a))
    lib().orc_logsumexp2D(ap, wp, _sz(a.shape[0]), _sz(a.shape[1]), res.ctypes.data_as(_dp))
    return res


# ----------------------------------------------------------------------------- densities
def gauss_multi_evaluate(x, mu, inv_sigma, log_normalization):
    x, xp = _d(x)
    mu, mup = _d(mu)
    s, sp = _d(inv_sigma)
    out = np.empty(len(x))
    lib().orc_gauss_multi_evaluate(xp, _sz(x.shape[0]), _sz(x.shape[1]), mup, sp,
                                   C.c_double(log_normalization), out.ctypes.data_as(_dp), _sz(1))
    return out


def student_t_multi_evaluate(x, mu, inv_sigma, log_norm, prefactor, inv_dof):
    x, xp = _d(x)
    mu, mup = _d(mu)
    s, sp = _d(inv_sigma)
    out = np.empty(len(x))
    lib().orc_student_t_multi_evaluate(xp, _sz(x.shape[0]), _sz(x.shape[1]), mup, sp,
                                       C.c_double(log_norm), C.c_double(prefactor),
                                       C.c_double(inv_dof), out.ctypes.data_as(_dp), _sz(1))
    return out


def mixture_multi_evaluate(family, x, weights, mu, inv_sigma, log_norm, prefactor=None,
                           inv_dof=None, components=None, individual=None, mt=False):
    """family 0 = Gauss, 1 = StudentT.  Returns (out or None, individual)."""
    x, xp = _d(x)
    N, D = x.shape
    w, wp = _d(weights)
    K = len(w)
    mu, mup = _d(np.reshape(mu, (K, D)))
    s, sp = _d(np.reshape(inv_sigma, (K, D, D)))
    ln, lnp = _d(log_norm)
    pf, pfp = _opt(prefactor)
    idf, idfp = _opt(inv_dof)
    if individual is None:
        individual = np.empty((N, K))
    assert individual.flags.c_contiguous and individual.shape == (N, K)
    indp = individual.ctypes.data_as(_dp)
    if components is None:
        out = np.empty(N)
        outp = out.ctypes.data_as(_dp)
        if mt:
            lib().orc_mixture_multi_evaluate_mt(C.c_int(family), xp, _sz(N), _sz(K), _sz(D), wp,
                                                mup, sp, lnp, pfp, idfp, indp, outp)
        else:
            lib().orc_mixture_multi_evaluate(C.c_int(family), xp, _sz(N), _sz(K), _sz(D), wp, mup,
                                             sp, lnp, pfp, idfp, indp, outp, None, _sz(0))
        return out, individual
    comps, cp = _i(list(components))
    lib().orc_mixture_multi_evaluate(C.c_int(family), xp, _sz(N), _sz(K), _sz(D), wp, mup, sp, lnp,
                                     pfp, idfp, indp, None, cp, _sz(len(comps)))
    return None, individual


# ----------------------------------------------------------------------------- importance sampling
def is_weights(log_target, log_proposal):
    t, tp = _d(log_target)
    q, qp = _d(log_proposal)
    w = np.empty(len(t))
    overflow = lib().orc_is_weights(tp, qp, _sz(len(t)), w.ctypes.data_as(_dp))
    if overflow:
        raise OverflowError('math range error')
    return w


def combine_weights_run(q, counts, t, omega, n_total, log_scale):
    """Deterministic-mixture weights of run ``t``; q[n, l] = log q_l(x^t_n)."""
    q, qp = _d(q)
    c, cp = _d(counts)
    o, op = _d(omega)
    out = np.empty(len(q))
    lib().orc_combine_weights(qp, _sz(q.shape[0]), _sz(q.shape[1]), cp, _sz(int(t)), op,
                              C.c_double(float(n_total)), C.c_int(int(bool(log_scale))),
                              out.ctypes.data_as(_dp))
    return out


def perp(weights):
    w, wp = _d(weights)
    return lib().orc_perp(wp, _sz(len(w)))


def ess(weights):
    w, wp = _d(weights)
    return lib().orc_ess(wp, _sz(len(w)))


# ----------------------------------------------------------------------------- variational Bayes
def vb_estep(data, weights, m, W, beta, nu, ln_pi, det_ln_lambda, mt=False):
    """N-sized part of GaussianInference.E_step.  Returns a dict with the reference's
    attribute names."""
    x, xp = _d(data)
    N, D = x.shape
    m, mp = _d(m)
    K = m.shape[0]
    W, Wp = _d(W)
    beta, bp = _d(beta)
    nu, nup = _d(nu)
    lp, lpp = _d(ln_pi)
    ll, llp = _d(det_ln_lambda)
    sw, swp = _opt(weights)
    E = np.empty((N, K))
    log_rho = np.empty((N, K))
    r = np.empty((N, K))
    N_comp = np.empty(K)
    inv_N_comp = np.empty(K)
    x_mean = np.empty((K, D))
    S = np.empty((K, D, D))
    elogqz = C.c_double(0.0)
    fn = lib().orc_vb_estep_mt if mt else lib().orc_vb_estep
    fn(xp, _sz(N), _sz(D), _sz(K), swp, mp, Wp, bp, nup, lpp, llp,
       E.ctypes.data_as(_dp), log_rho.ctypes.data_as(_dp), r.ctypes.data_as(_dp),
       N_comp.ctypes.data_as(_dp), inv_N_comp.ctypes.data_as(_dp),
       x_mean.ctypes.data_as(_dp), S.ctypes.data_as(_dp), C.byref(elogqz))
    return dict(expectation_gauss_exponent=E, log_rho=log_rho, r=r, N_comp=N_comp,
                inv_N_comp=inv_N_comp, x_mean_comp=x_mean, S=S,
                expectation_log_q_Z=elogqz.value)


# ----------------------------------------------------------------------------- PMC
def rho_rb(family, samples, weights, mu, inv_sigma, log_norm, prefactor, inv_dof, live, mt=False):
    """calculate_rho_rb (pmc.pyx:23-43).  ``mt`` (every component live only): the component densities on all host
    cores, sample chunks in parallel, same per-pair arithmetic."""
    x = np.ascontiguousarray(samples, dtype=np.float64)
    K = len(weights)
    rho = np.zeros((len(x), K))
    if mt and list(live) == list(range(K)):
        mixture_multi_evaluate(family, x, weights, mu, inv_sigma, log_norm, prefactor, inv_dof, individual=rho, mt=True)
    else:
        mixture_multi_evaluate(family, x, weights, mu, inv_sigma, log_norm, prefactor, inv_dof,
                               components=live, individual=rho)
    w, wp = _d(weights)
    lv, lvp = _i(list(live))
    lib().orc_rho_rb_finish(rho.ctypes.data_as(_dp), _sz(len(x)), _sz(K), wp, lvp, _sz(len(lv)))
    return rho


def rho_non_rb(N, K, latent, live):
    rho = np.zeros((N, K))
    lat = np.ascontiguousarray(latent, dtype=np.int64)
    lv, lvp = _i(list(live))
    lib().orc_rho_non_rb(rho.ctypes.data_as(_dp), _sz(N), _sz(K), lat.ctypes.data_as(_lp), lvp,
                         _sz(len(lv)))
    return rho


def student_t_gamma(samples, mu, inv_sigma, dof, live):
    x, xp = _d(samples)
    N, D = x.shape
    mu, mup = _d(mu)
    K = mu.shape[0]
    s, sp = _d(inv_sigma)
    dof, dp = _d(dof)
    lv, lvp = _i(list(live))
    gamma = np.zeros((N, K))
    lib().orc_student_t_gamma(xp, _sz(N), _sz(D), _sz(K), mup, sp, dp, lvp, _sz(len(lv)),
                              gamma.ctypes.data_as(_dp))
    return gamma


def pmc_reductions(samples, rho, gamma, weights, live, mt=False):
    """alpha (unnormalised), mu, cov of gaussian_pmc / student_t_pmc.  ``mt``: the components spread over the host's
    cores (bit-identical results: every component's sums keep their order)."""
    x, xp = _d(samples)
    N, D = x.shape
    rho, rp = _d(rho)
    K = rho.shape[1]
    g, gp = _opt(gamma)
    w, wp = _opt(weights)
    lv, lvp = _i(list(live))
    alpha = np.empty(K)
    mu = np.empty((K, D))
    cov = np.zeros((K, D, D))
    (lib().orc_pmc_reductions_mt if mt else lib().orc_pmc_reductions)(xp, _sz(N), _sz(D), _sz(K), rp, gp, wp, lvp, _sz(len(lv)),
                             alpha.ctypes.data_as(_dp), mu.ctypes.data_as(_dp),
                             cov.ctypes.data_as(_dp))
    return alpha, mu, cov


def student_t_dof_const(samples, rho, weights, weight_normalization, mu, inv_sigma, dof,
                        psi_half_dim_nu, psi_half_nu, live):
    x, xp = _d(samples)
    N, D = x.shape
    rho, rp = _d(rho)
    K = rho.shape[1]
    w, wp = _opt(weights)
    mu, mup = _d(mu)
    s, sp = _d(inv_sigma)
    dof, dp = _d(dof)
    p1, p1p = _d(psi_half_dim_nu)
    p2, p2p = _d(psi_half_nu)
    lv, lvp = _i(list(live))
    out = np.full(K, np.nan)
    lib().orc_student_t_dof_const(xp, _sz(N), _sz(D), _sz(K), rp, wp,
                                  C.c_double(weight_normalization), mup, sp, dp, p1p, p2p, lvp,
                                  _sz(len(lv)), out.ctypes.data_as(_dp))
    return out


def pmc_log_likelihood(log_q, normalized_weights=None):
    q, qp = _d(log_q)
    w, wp = _opt(normalized_weights)
    return lib().orc_pmc_log_likelihood(qp, wp, _sz(len(q)))
