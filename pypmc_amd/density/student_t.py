"""Student's t density with its log-pdf on the GPU (reference: pypmc/density/student_t.pyx)."""
import numpy as np
from scipy.special import gammaln

from .gauss import Gauss, LocalGauss, _STAMPS, _as_matrix
from .._lib import PMC_KIND_STUDENT_T


class StudentT(Gauss):
    r"""Multivariate t_nu(mu, sigma).  Usable as a MixtureDensity component."""

    kind = PMC_KIND_STUDENT_T

    def __init__(self, mu, sigma, dof, backend=None):
        self._backend = backend
        self.update(mu, sigma, dof)

    def update(self, mu, sigma, dof):
        """Replace mean, covariance and degrees of freedom; a ``LinAlgError`` leaves the object
        untouched (reference: student_t.pyx:78-117)."""
        dof = float(dof)
        assert dof > 0., "Degree of freedom (``dof``) must be greater than zero (got %g)." % dof
        Gauss.update(self, mu, sigma)                    # raises before anything is modified
        self.dof = dof
        # student_t.pyx:33-34
        self.log_normalization = gammaln(.5 * (self.dof + self.dim)) - gammaln(.5 * self.dof) \
            - 0.5 * self.dim * np.log(self.dof * np.pi) - 0.5 * self.log_det_sigma
        self._eval_prefactor = -.5 * (self.dof + self.dim)      # student_t.pyx:116
        self._inv_dof = 1. / self.dof                           # :117
        self._stamp = next(_STAMPS)                             # (dof is part of the parameter state)

    def _assign(self, mu, sigma, cholesky_sigma, inv_sigma, log_det_sigma, dof):
        Gauss._assign(self, mu, sigma, cholesky_sigma, inv_sigma, log_det_sigma)
        self.dof = float(dof)
        self.log_normalization = gammaln(.5 * (self.dof + self.dim)) - gammaln(.5 * self.dof) \
            - 0.5 * self.dim * np.log(self.dof * np.pi) - 0.5 * self.log_det_sigma
        self._eval_prefactor = -.5 * (self.dof + self.dim)
        self._inv_dof = 1. / self.dof
        self._stamp = next(_STAMPS)

    def _kernel_constants(self):
        return self.log_normalization, self._eval_prefactor, self._inv_dof, self.dof

    def propose(self, N=1, rng=np.random.mtrand):
        """mu + L z sqrt(nu / chi2_nu), one normal vector and one chi-square per sample in the
        reference's call order (student_t.pyx:49-55, :172-176)."""
        out = np.empty((int(N), self.dim))
        for n in range(int(N)):
            g = self.cholesky_sigma.dot(rng.normal(0, 1, self.dim))
            out[n] = self.mu + g * np.sqrt(self.dof / rng.chisquare(self.dof))
        return out


class LocalStudentT(LocalGauss):
    """Local Student's t density t_nu(x | y, sigma) with redefinable covariance (reference: student_t.pyx:13-55)."""

    def __init__(self, sigma, dof, backend=None):
        self.symmetric = True
        dof = float(dof)
        assert dof > 0., "Degree of freedom (``dof``) must be greater than zero (got %g)." % dof
        self.dof = dof
        self._backend = backend
        self.update(sigma)

    def update(self, sigma):
        sigma = _as_matrix(sigma)
        self._set(StudentT(np.zeros(sigma.shape[0]), sigma, self.dof, backend=self._backend))   # may raise LinAlgError

    def propose(self, y, rng=np.random.mtrand):
        # one normal vector, then one chi-square per sample (student_t.pyx:49-55)
        return y + self._get_gauss_sample(rng) * np.sqrt(self.dof / rng.chisquare(self.dof))

