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


_CONTROLLER = []                 # the process's threadpoolctl.ThreadpoolController, looked up once


class single_threaded_blas(object):
    """Context for loops over many small factorisations: the BLAS thread pool only costs there
    (D = 40 on a 128-core host: 45 -> 8 us per potri).  The library scan behind
    ``threadpoolctl.threadpool_limits`` takes a millisecond, so the controller is created once and only its
    ``limit`` is entered per use.  No-op without threadpoolctl."""

    def __enter__(self):
        try:
            if not _CONTROLLER:
                from threadpoolctl import ThreadpoolController
                _CONTROLLER.append(ThreadpoolController())
            self._ctx = _CONTROLLER[0].limit(limits=1, user_api='blas')
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


def chol_inv_det_batch(ms, check_symmetric=True):
    """``chol_inv_det`` of a stack of K matrices in a handful of array operations: the K-sized host update of
    a PMC / VB iteration is a Python loop over components otherwise, and at N <= 1e6 that loop -- not the
    kernels -- is the iteration's wall time (K = 128, D = 40: 14 ms against 10 ms on the device).

    Same checks and the same LAPACK factorisation as ``chol_inv_det`` (finite, symmetric to numpy.allclose's
    tolerances, potrf, potri of the lower factor, mirrored inverse, log det summed left to right); raises
    ``numpy.linalg.LinAlgError`` if ANY matrix fails, without saying which -- callers that need the
    reference's per-component behaviour (object left unchanged, weight zeroed) fall back to the loop then.
    ``check_symmetric=False`` is for callers whose matrices are symmetric by construction (mirrored by the
    finishing kernel, or sums of x x^T): the comparison is a third of the whole call.
    Returns lower (K, D, D), inverse (K, D, D), log_det (K)."""
    ms = np.asarray(ms, dtype=np.float64)
    if ms.ndim != 3 or ms.shape[1] != ms.shape[2]:
        raise np.linalg.LinAlgError('expected a stack of square matrices, got shape %s' % (ms.shape,))
    if not np.isfinite(ms).all():
        raise np.linalg.LinAlgError('array must not contain infs or NaNs')
    if check_symmetric:
        mt = np.ascontiguousarray(ms.transpose(0, 2, 1))         # strided operands cost numpy 4x here
        if not (np.abs(ms - mt) <= 1e-8 + 1e-5 * np.abs(mt)).all():
            raise np.linalg.LinAlgError('matrix not symmetric')
    K, D = ms.shape[0], ms.shape[1]
    native = _native_batch(np.ascontiguousarray(ms), K, D) if K * D * D >= 20000 else None
    if native is not None:
        return native
    with single_threaded_blas():
        lower = np.linalg.cholesky(ms)                           # batched potrf; LinAlgError if one is not PD
        # potri has no batched form.  Each call gets its factor as a Fortran-ordered view of a transposed copy
        # and works in place there (f2py copies and transposes a C-ordered argument: 27 -> 10 us per call)
        work = np.array(lower.transpose(0, 2, 1), order='C', copy=True)   # (a copy also at D = 1, where the transpose is contiguous)
        for k in range(K):
            res = _POTRI(work[k].T, True, overwrite_c=True)[0]
            if res.__array_interface__['data'][0] != work[k].__array_interface__['data'][0]:
                work[k] = res.T                                  # f2py chose to copy after all
    # potri filled the lower triangle of each work[k].T, i.e. the upper triangle of work[k]; mirror it.  Adding
    # the zeros of the other triangle is exact
    inverse = np.triu(work)
    inverse += np.ascontiguousarray(np.triu(work, 1).transpose(0, 2, 1))
    diag = np.log(np.diagonal(lower, axis1=1, axis2=2))
    log_det = 2.0 * np.cumsum(diag, axis=1)[:, -1] if D else np.zeros(K)    # left-to-right, as the reference sums
    if not np.isfinite(log_det).all():
        raise np.linalg.LinAlgError('Nonpositive eigenvalues lead to invalid determinant')
    return lower, inverse, log_det


_LAPACK_PTRS = []


def _native_batch(ms, K, D):
    """The whole batch in one call of the library's host code (pmc_host_chol_inv_det_batch, include/pmc_ctx.h): scipy's
    own dpotrf / dpotri by address -- the same routines, hence the same bits, as the loop below, without its per-matrix
    interpreter overhead, transposed copies and triangle masks (K = 128, D = 40: 6.4 -> 2.7 ms on the build container).  None if the library or scipy's
    LAPACK capsules are not available (the loop runs then); LinAlgError if a matrix does not factorise."""
    import ctypes as C
    try:
        from .. import _lib
        lib = _lib.load()
        if not _LAPACK_PTRS:
            import scipy.linalg.cython_lapack as cl
            get = C.pythonapi.PyCapsule_GetPointer
            get.restype, get.argtypes = C.c_void_p, [C.py_object, C.c_char_p]
            name = C.pythonapi.PyCapsule_GetName
            name.restype, name.argtypes = C.c_char_p, [C.py_object]
            _LAPACK_PTRS.extend(get(cl.__pyx_capi__[f], name(cl.__pyx_capi__[f])) for f in ("dpotrf", "dpotri"))
    except Exception:                                        # pragma: no cover
        return None
    lower, inverse, log_det = np.empty((K, D, D)), np.empty((K, D, D)), np.empty(K)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    with single_threaded_blas():
        rc = lib.pmc_host_chol_inv_det_batch(K, D, dp(ms), C.c_void_p(_LAPACK_PTRS[0]), C.c_void_p(_LAPACK_PTRS[1]), dp(lower),
                                             dp(inverse), dp(log_det), None)
    if rc != 0:
        raise np.linalg.LinAlgError('a matrix of the batch is not positive definite (or its determinant is not finite)')
    # (the determinant once more with numpy's log, so that it is bit for bit what chol_inv_det gives)
    log_det = 2.0 * np.cumsum(np.log(np.diagonal(lower, axis1=1, axis2=2)), axis=1)[:, -1]
    if not np.isfinite(log_det).all():
        raise np.linalg.LinAlgError('Nonpositive eigenvalues lead to invalid determinant')
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
