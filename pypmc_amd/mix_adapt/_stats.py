"""K-sized host conversions of the statistics vector produced by pmc_sufficient_stats
(include/pmc_hip.h) into the reference's centred conventions."""
import ctypes as _C

import numpy as np

from .. import _lib
from .._lib import NSCALARS

TINY = np.finfo('d').tiny


def regularize(x):
    """pypmc/tools/_regularize.pyx:6-17: zeros -> smallest positive float (in place)."""
    x[x == 0] = TINY
    return x


def split_stats(flat, K, D):
    """flat = [scalars(NSCALARS) | K x (1 + D + D(D+1)/2) | K x 2] (host array) ->
    scalars, S0 (K), M1 (K,D), M2 (K,D,D symmetric), V1 (K), V2 (K)."""
    flat = np.asarray(flat, dtype=np.float64)
    T = D * (D + 1) // 2
    ps = 1 + D + T
    scalars = flat[:NSCALARS].copy()
    body = flat[NSCALARS:NSCALARS + K * ps].reshape(K, ps)
    S0 = body[:, 0].copy()
    M1 = body[:, 1:1 + D].copy()
    M2 = np.zeros((K, D, D))
    il, jl = np.tril_indices(D)          # row-major (i, j<=i): the kernel's packing order
    M2[:, il, jl] = body[:, 1 + D:1 + D + T]
    M2[:, jl, il] = body[:, 1 + D:1 + D + T]
    vs = flat[NSCALARS + K * ps:NSCALARS + K * ps + 2 * K].reshape(K, 2)
    return scalars, S0, M1, M2, vs[:, 0].copy(), vs[:, 1].copy()


def centred_moments(S0_mean, M1, M2, shift, S0_cov=None):
    """mean_k = shift_k + M1_k / reg(S0_mean_k);
    cov_k = (M2_k - S0_mean_k dbar dbar^T) / reg(S0_cov_k)   with dbar = M1_k / reg(S0_mean_k).

    With S0_cov = S0_mean this is  sum u (x - mean)(x - mean)^T / sum u  (variational.pyx:855-932,
    pmc.pyx:198-204); the Student-t update normalises the covariance by a different sum
    (pmc.pyx:629-630)."""
    n_mean = regularize(np.array(S0_mean, dtype=np.float64))
    n_cov = n_mean if S0_cov is None else regularize(np.array(S0_cov, dtype=np.float64))
    dbar = M1 / n_mean[:, None]
    mean = shift + dbar
    cov = (M2 - n_mean[:, None, None] * np.einsum('ki,kj->kij', dbar, dbar)) / n_cov[:, None, None]
    return mean, cov


def shift_is_far(S0, M1, M2, limit=100.):
    """True if some component's weighted mean lies more than sqrt(limit) of its own standard deviations (in some
    coordinate) away from the shift its one-pass moments were taken about.  cov = M2/S0 - dbar dbar^T then cancels
    ~limit leading parts: the relative error of the covariance grows to limit * 1e-16, against the 1e-16 of the
    reference's two passes (mean first, then moments about it: variational.pyx:806-932, pmc.pyx:188-222).  The
    callers answer with a second pass of the statistics about the mean just found.  Components that hold less than
    a millionth of the total weight (dying ones, whose few far samples would ask for a second pass in every
    iteration until they are pruned) or non-finite sums do not count."""
    S0 = np.asarray(S0, dtype=np.float64)
    fin = np.isfinite(S0)
    ok = fin & (S0 > 1e-200) & (S0 > 1e-6 * S0[fin].sum() if fin.any() else False)
    if not ok.any():
        return False
    n = S0[ok][:, None]
    dbar2 = (M1[ok] / n) ** 2
    raw = np.einsum('kii->ki', M2[ok]) / n                   # E[d_i^2] about the shift
    var = np.maximum(raw - dbar2, 1e-14 * raw)
    with np.errstate(invalid='ignore'):
        return bool(np.any(dbar2 > limit * var))



def convert_stats(flat, K, D, shift, n_cov=None):
    """``split_stats`` + ``shift_is_far`` + ``centred_moments`` in one call of the library's host-side
    ``pmc_host_convert_stats`` (include/pmc_ctx.h; the same operations in the same order: bit-identical to the numpy
    functions above, a tenth of their time at K = 64, D = 20).  ``flat`` as ``split_stats`` takes it; ``n_cov`` (K): the
    covariance's normalisation when it is not the first sum (``'vsum0'`` = the first Student-t sum of ``flat``).
    Returns scalars, S0 (raw), M1, mean, cov, far, V1, V2."""
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    T = D * (D + 1) // 2
    ps = 1 + D + T
    scalars = flat[:NSCALARS].copy()
    vs = flat[NSCALARS + K * ps:NSCALARS + K * ps + 2 * K].reshape(K, 2)
    V1, V2 = vs[:, 0].copy(), vs[:, 1].copy()
    if isinstance(n_cov, str):
        n_cov = V1
    shift = np.ascontiguousarray(shift, dtype=np.float64).reshape(K, D)
    S0, M1 = np.empty(K), np.empty((K, D))
    mean, cov = np.empty((K, D)), np.empty((K, D, D))
    far = _C.c_int(0)
    dp = lambda a: a.ctypes.data_as(_C.POINTER(_C.c_double))
    body = flat[NSCALARS:NSCALARS + K * ps]
    nc = None if n_cov is None else np.ascontiguousarray(n_cov, dtype=np.float64)
    _lib.check(_lib.load().pmc_host_convert_stats(K, D, dp(body), dp(shift), None if nc is None else dp(nc), dp(S0), dp(M1),
                                                  dp(mean), dp(cov), _C.byref(far)), "pmc_host_convert_stats")
    return scalars, S0, M1, mean, cov, bool(far.value), V1, V2
