"""Host-side K-sized linear algebra of the path (reference: pypmc/tools/_linalg.pyx).

``chol_inv_det`` stays on the host exactly as in the reference (a handful of D x D LAPACK
factorisations per update); ``bilinear_sym`` is the quadratic form the kernels evaluate per
(sample, component) and is offered here for single vectors through the same HIP path."""
import numpy as np
from scipy.linalg import cholesky
from scipy.linalg.lapack import get_lapack_funcs


_POTRI = get_lapack_funcs('potri', (np.empty((1, 1)),))     # dpotri, looked up once
_TRIL = {}


def _strict_lower(dim):
    if dim not in _TRIL:
        _TRIL[dim] = np.tril_indices(dim, -1)
    return _TRIL[dim]


class single_threaded_blas(object):
    """Context for loops over many small factorisations: the BLAS thread pool only costs there
    (D = 40: 93 -> 60 us per chol_inv_det).  No-op without threadpoolctl."""

    def __enter__(self):
        try:
            from threadpoolctl import threadpool_limits
            self._ctx = threadpool_limits(limits=1, user_api='blas')
            self._ctx.__enter__()
        except Exception:                                    # pragma: no cover
            self._ctx = None
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def chol_inv_det(m):
    """Lower Cholesky factor L (m = L L^T), the symmetrised inverse and log(det m).
    Raises ``numpy.linalg.LinAlgError`` for asymmetric, non positive definite or non-finite
    input (reference: _linalg.pyx:41-95 -- same LAPACK calls: potrf via scipy, potri).

    K of these run per proposal update; at K = 128, D = 40 they were a quarter of a
    device-resident PMC iteration, hence the care about Python overhead here."""
    m = np.asarray_chkfinite(m)
    # numpy.allclose(m, m.T) with its default tolerances, minus its generality
    if m.ndim != 2 or m.shape[0] != m.shape[1] or \
            not (np.abs(m - m.T) <= 1e-8 + 1e-5 * np.abs(m.T)).all():
        raise np.linalg.LinAlgError('matrix not symmetric:\n' + repr(m))
    lower = cholesky(m, lower=True, check_finite=False)      # LinAlgError if not positive definite
    inverse = _POTRI(lower, True)[0]                          # only the lower triangle is meaningful
    il, jl = _strict_lower(len(m))
    inverse[jl, il] = inverse[il, jl]                         # mirror it
    log_det = 0.0
    for v in np.log(np.diag(lower)).tolist():                 # left-to-right, as the reference sums
        log_det += v
    log_det *= 2.0
    if not np.isfinite(log_det):
        raise np.linalg.LinAlgError('Nonpositive eigenvalues lead to invalid determinant ' + repr(log_det))
    return lower, inverse, log_det


def bilinear_sym(matrix, vector, backend=None):
    """x^T M x for a symmetric M (reference: _linalg.pyx:10-39) through the Mahalanobis kernel.

    On the hot path M is always positive definite (``inv_sigma`` / ``W`` come out of
    ``chol_inv_det``) and the kernel evaluates |R x|^2 with M = R^T R.  This stand-alone helper
    also accepts an indefinite M by shifting it:  x^T M x = x^T (M + c I) x - c x^T x."""
    from ..backend import ComponentSet, get_backend
    be = get_backend(backend)
    v = np.asarray(vector, dtype=np.float64).reshape(1, -1)
    M = np.asarray(matrix, dtype=np.float64)
    D = v.shape[1]
    c = float(np.linalg.norm(M)) + 1.0            # > |lambda_min|
    prec = np.array([M + c * np.eye(D), np.eye(D)])
    ind = be.tohost(be.logpdf(v, ComponentSet(0, np.zeros((2, D)), prec), want_out=False,
                              want_individual=True)["individual"])
    return float(-2.0 * ind[0, 0] - c * (-2.0 * ind[0, 1]))
