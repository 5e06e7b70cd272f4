"""Singularity guards and log-sum-exp (reference: pypmc/tools/_regularize.pyx)."""
import numpy as np

TINY = np.finfo('d').tiny


def regularize(x):
    """Zeros become the smallest positive double, in place (reference: _regularize.pyx:6-17)."""
    x[x == 0] = TINY
    return x


def logsumexp2D(a, weights, backend=None):
    """Row-wise log sum_k w_k exp(a_nk) on the GPU (reference: _regularize.pyx:57-84)."""
    from ..backend import get_backend
    assert a is not None
    assert weights is not None
    weights = np.asarray(weights, dtype=np.float64)
    assert (weights >= 0.).all(), 'Found negative weight'
    be = get_backend(backend)
    return be.tohost(be.logsumexp2d(np.asarray(a, dtype=np.float64), weights))


def logsumexp(a, weights, backend=None):
    """log sum_i w_i exp(a_i) of one vector (reference: _regularize.pyx:19-55)."""
    assert a is not None
    assert weights is not None
    a = np.asarray(a, dtype=np.float64)
    return float(logsumexp2D(a.reshape(1, -1), weights, backend)[0])
