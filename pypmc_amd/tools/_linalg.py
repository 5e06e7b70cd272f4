"""Host-side K-sized linear algebra of the path (reference: pypmc/tools/_linalg.pyx).

``chol_inv_det`` stays on the host exactly as in the reference (a handful of D x D LAPACK
factorisations per update); ``bilinear_sym`` is the quadratic form the kernels evaluate per
(sample, component) and is offered here for single vectors through the same HIP path."""
import numpy as np
from scipy.linalg import cholesky
from scipy.linalg.lapack import get_lapack_funcs


def chol_inv_det(m):
    """Lower Cholesky factor L (m = L L^T), the symmetrised inverse and log(det m).
    Raises ``numpy.linalg.LinAlgError`` for asymmetric, non positive definite or non-finite
    input (reference: _linalg.pyx:41-95 -- same LAPACK calls: potrf via scipy, potri)."""
    m = np.asarray_chkfinite(m)
    if not np.allclose(m, m.T):
        raise np.linalg.LinAlgError('matrix not symmetric:\n' + repr(m))
    lower = cholesky(m, lower=True)              # LinAlgError if not positive definite
    potri = get_lapack_funcs('potri', (m,))
    inverse = potri(lower, True)[0]              # only the lower triangle is meaningful
    il, jl = np.tril_indices(len(m), -1)
    inverse[jl, il] = inverse[il, jl]
    log_det = 0.0
    for i in range(len(m)):
        log_det += np.log(lower[i, i])
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
