"""Gaussian density with its log-pdf on the GPU (reference: pypmc/density/gauss.pyx)."""
import itertools

import numpy as np

from .base import ProbabilityDensity, LocalDensity
from ..tools._linalg import chol_inv_det
from ..backend import ComponentSet, get_backend
from .._lib import PMC_KIND_GAUSS, check_dim


# every parameter state of a component gets a process-unique stamp (set by update / _assign, shared by in-process
# copies, which hold the same parameters): density.mixture.component_set keys its cache of uploaded parameter packs
# on it.  A stamp never leaves its process: pickling drops it and unpickling draws a new one (__getstate__ /
# __setstate__), so a component that arrives from another process or from disk cannot collide with a local one.
_STAMPS = itertools.count(1)


def _as_matrix(sigma):
    """scalar / nested list / array -> fresh 2-D float array (1 x 1 for a scalar)."""
    s = np.array(sigma, dtype=float)
    if s.ndim == 0:
        s = s.reshape(1, 1)
    elif s.ndim == 1:
        s = s.reshape(1, -1) if s.size == 1 else np.atleast_2d(s)
    return s


class Gauss(ProbabilityDensity):
    r"""N(mu, sigma).  Usable as a MixtureDensity component.

    Host state (authoritative, picklable): ``mu, sigma, inv_sigma, cholesky_sigma,
    log_det_sigma, log_normalization, dim``.  ``evaluate``/``multi_evaluate`` run the HIP
    mixture kernel with a single component."""

    def __init__(self, mu, sigma, backend=None):
        self._backend = backend
        self.update(mu, sigma)

    def update(self, mu, sigma):
        """Replace mean and covariance.  A ``LinAlgError`` (asymmetric / not positive definite /
        non-finite covariance) leaves the object untouched (reference: gauss.pyx:40-48, :86-116)."""
        sigma = _as_matrix(sigma)
        cholesky_sigma, inv_sigma, log_det_sigma = chol_inv_det(sigma)   # may raise LinAlgError
        mu = np.array(mu, dtype=float).reshape(-1)
        check_dim(len(mu))
        assert len(mu) == sigma.shape[0], \
            "Dimensions of mean (%d) and covariance matrix (%d) do not match!" % (len(mu), sigma.shape[0])
        self.mu, self.sigma, self.dim = mu, sigma, len(mu)
        self.cholesky_sigma, self.inv_sigma, self.log_det_sigma = cholesky_sigma, inv_sigma, log_det_sigma
        # gauss.pyx:56
        self.log_normalization = -0.5 * self.dim * np.log(2 * np.pi) - 0.5 * self.log_det_sigma
        self._stamp = next(_STAMPS)

    def __deepcopy__(self, memo):
        """Components are value objects made of a few arrays and scalars; copying them attribute by attribute
        is what ``copy.deepcopy`` does too, minus its bookkeeping (K = 128: 3.4 -> 0.4 ms per mixture; every
        ``gaussian_pmc(..., copy=True)`` and every ImportanceSampler copies its proposal)."""
        new = self.__class__.__new__(self.__class__)
        for key, value in self.__dict__.items():
            new.__dict__[key] = value.copy() if isinstance(value, np.ndarray) else value
        return new

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_stamp', None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._stamp = next(_STAMPS)

    def _assign(self, mu, sigma, cholesky_sigma, inv_sigma, log_det_sigma):
        """``update`` with the factorisation already done (mix_adapt's batched K-sized updates)."""
        self.mu, self.sigma, self.dim = mu, sigma, len(mu)
        self.cholesky_sigma, self.inv_sigma, self.log_det_sigma = cholesky_sigma, inv_sigma, log_det_sigma
        self.log_normalization = -0.5 * self.dim * np.log(2 * np.pi) - 0.5 * self.log_det_sigma
        self._stamp = next(_STAMPS)

    # -- kernel description ------------------------------------------------------------------
    kind = PMC_KIND_GAUSS

    def _kernel_constants(self):
        """(c0, c1, c2, c3) of ``enum pmc_kind`` for this component."""
        return self.log_normalization, 0., 0., 0.

    def _component_set(self):
        return ComponentSet(self.kind, self.mu[None], self.inv_sigma[None],
                            *[np.array([c]) for c in self._kernel_constants()])

    # -- evaluation ---------------------------------------------------------------------------
    def evaluate(self, x):
        x = np.asarray(x, dtype=np.float64).reshape(1, -1)
        return float(self.multi_evaluate(x)[0])

    def multi_evaluate(self, x, out=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.ndim == 2 and x.shape[1] == self.dim
        if out is None:
            out = np.empty(len(x))
        else:
            assert len(out) == len(x)
        be = get_backend(self._backend)
        out[:] = be.tohost(be.logpdf(x, self._component_set())["out"])
        return out

    # -- sampling (host; the generator stream is consumed exactly as the reference does) ---------
    def propose(self, N=1, rng=np.random.mtrand):
        """mu + L z with z ~ N(0, 1)^D drawn sample by sample (reference: gauss.pyx:50-52,
        :159-163).  With numpy's own generators one (N, D) draw consumes the stream identically; any other
        ``rng`` object is asked for ``rng.normal(0, 1, dim)`` once per sample, exactly as the reference does."""
        N = int(N)
        if not N:
            return np.empty((0, self.dim))
        if _is_numpy_rng(rng):
            z = np.asarray(rng.normal(0, 1, (N, self.dim)), dtype=float).reshape(N, self.dim)
        else:
            z = np.array([np.asarray(rng.normal(0, 1, self.dim), dtype=float).reshape(self.dim) for _ in range(N)])
        return self.mu + z.dot(self.cholesky_sigma.T)


def _is_numpy_rng(rng):
    return rng is np.random or rng is np.random.mtrand or isinstance(rng, (np.random.RandomState, np.random.Generator))


class LocalGauss(LocalDensity):
    """Local Gaussian N(x | y, sigma) with redefinable covariance (reference: gauss.pyx:12-67).  ``evaluate`` is
    the zero-mean Gaussian's log-pdf at x - y on the HIP path; ``propose`` consumes ``rng`` as the reference does."""
    symmetric = True

    def __init__(self, sigma, backend=None):
        self._backend = backend
        self.update(sigma)

    def update(self, sigma):
        """New covariance; a ``LinAlgError`` leaves the object untouched (gauss.pyx:22-48)."""
        sigma = _as_matrix(sigma)
        self._set(Gauss(np.zeros(sigma.shape[0]), sigma, backend=self._backend))       # may raise LinAlgError

    def _set(self, centred):
        self._centred = centred
        self.sigma, self.dim = centred.sigma, centred.dim
        self.cholesky_sigma, self.inv_sigma, self.log_det_sigma = centred.cholesky_sigma, centred.inv_sigma, centred.log_det_sigma
        self._compute_norm()

    def _compute_norm(self):
        self.log_normalization = self._centred.log_normalization                      # gauss.pyx:54-56

    def _get_gauss_sample(self, rng):
        return np.dot(self.cholesky_sigma, rng.normal(0, 1, self.dim))                # gauss.pyx:50-52

    def evaluate(self, x, y):
        return self._centred.evaluate(np.asarray(x, dtype=float) - np.asarray(y, dtype=float))   # gauss.pyx:59-60

    def propose(self, y, rng=np.random.mtrand):
        return y + self._get_gauss_sample(rng)                                        # gauss.pyx:66-67
