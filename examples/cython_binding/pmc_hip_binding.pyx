# cython: language_level=3, boundscheck=False, wraparound=False
"""What a pypmc maintainer adds to bind the MI355X path INSIDE the reference's own Cython sources: the handle layer
of libpmc_hip.so (include/pmc_ctx.h) declared `cdef extern`, and the bodies of the three hot methods replaced by one
call each, on the typed memoryviews the .pyx files already hold.

    GaussianInference.E_step                    pypmc/mix_adapt/variational.pyx:116-127
    MixtureDensity.multi_evaluate               pypmc/density/mixture.pyx:112-156 (and its `components=` mode, :153-156)
    gaussian_pmc (its N-sized part)             pypmc/mix_adapt/pmc.pyx:188-222

Built and run by tests/test_gpu_cython_binding.py (cythonize + gcc, no HIP headers, no torch):
    cythonize -i examples/cython_binding/pmc_hip_binding.pyx   with  -I include -L pypmc_amd/lib -lpmc_hip
"""
import numpy as np

cdef extern from "pmc_ctx.h":
    ctypedef struct pmc_ctx
    ctypedef struct pmc_mix
    ctypedef struct pmc_samples
    const char *pmc_last_error()
    int pmc_init(int device, pmc_ctx **out)
    int pmc_shutdown(pmc_ctx *ctx)
    int pmc_mixture_create(pmc_ctx *ctx, int family, int K, int D, const double *w, const double *mu,
                           const double *inv_sigma, const double *log_norm, const double *dof, pmc_mix **out)
    int pmc_mixture_destroy(pmc_mix *mix)
    int pmc_samples_upload(pmc_ctx *ctx, const double *x, long long N, int D, pmc_samples **out)
    int pmc_samples_free(pmc_samples *s)
    int pmc_mix_logpdf(const pmc_mix *mix, const pmc_samples *s, double *out, double *individual)
    int pmc_mix_logpdf_components(const pmc_mix *mix, const pmc_samples *s, const int *components, int ncomponents,
                                  double *individual)
    int pmc_vb_estep(pmc_ctx *ctx, const pmc_samples *s, const double *sample_w, int K, const double *m, const double *W,
                     const double *nu, const double *beta, const double *ln_pi, const double *ln_lambda,
                     const double *shift, double *N_k, double *xbar, double *S, double *elogqz, double *r, double *log_rho)
    int pmc_pmc_update_stats(pmc_ctx *ctx, const pmc_mix *mix, const pmc_samples *s, const double *w, int weights_on_device,
                             const long long *latent, int rb, double *alpha, double *mu, double *sigma, double *dof_const,
                             double *loglik, double *norm)


cdef pmc_ctx *_ctx = NULL


cdef int _check(int rc) except -1:
    if rc != 0:
        raise RuntimeError(pmc_last_error().decode())
    return 0


cdef pmc_ctx *_context() except NULL:
    global _ctx
    if _ctx == NULL:
        _check(pmc_init(0, &_ctx))
    return _ctx


def shutdown():
    global _ctx
    if _ctx != NULL:
        pmc_shutdown(_ctx)
        _ctx = NULL


cdef class DeviceSamples:
    """the data of a GaussianInference / the samples of an update, resident on the GPU (made once, in __init__)"""
    cdef pmc_samples *handle
    cdef readonly long long N
    cdef readonly int dim

    def __cinit__(self, double[:, ::1] x):
        self.handle = NULL
        self.N, self.dim = x.shape[0], x.shape[1]
        _check(pmc_samples_upload(_context(), &x[0, 0], self.N, self.dim, &self.handle))

    def __dealloc__(self):
        if self.handle != NULL:
            pmc_samples_free(self.handle)


def E_step(DeviceSamples data, double[:, ::1] m, double[:, :, ::1] W, double[::1] nu, double[::1] beta,
           double[::1] expectation_ln_pi, double[::1] expectation_det_ln_lambda, weights=None):
    """the body of GaussianInference.E_step after its two K-sized updates (variational.pyx:118-119):
    returns N_comp, x_mean_comp, S, E[log q(Z)] -- what :675-757, :806-932, :1003-1013 compute"""
    cdef int K = m.shape[0], D = m.shape[1]
    cdef double[::1] N_comp = np.empty(K)
    cdef double[:, ::1] x_mean = np.empty((K, D))
    cdef double[:, :, ::1] S = np.empty((K, D, D))
    cdef double elogqz = 0.
    cdef double[::1] sw
    cdef const double *swp = NULL
    if weights is not None:
        sw = np.ascontiguousarray(weights, dtype=np.float64)
        swp = &sw[0]
    _check(pmc_vb_estep(_context(), data.handle, swp, K, &m[0, 0], &W[0, 0, 0], &nu[0], &beta[0], &expectation_ln_pi[0],
                        &expectation_det_ln_lambda[0], NULL, &N_comp[0], &x_mean[0, 0], &S[0, 0, 0], &elogqz, NULL, NULL))
    return np.asarray(N_comp), np.asarray(x_mean), np.asarray(S), elogqz


def multi_evaluate(DeviceSamples x, double[::1] weights, double[:, ::1] mu, double[:, :, ::1] inv_sigma,
                   double[::1] log_norm):
    """MixtureDensity.multi_evaluate of a Gaussian mixture (mixture.pyx:112-156): out, individual"""
    cdef int K = mu.shape[0], D = mu.shape[1]
    cdef pmc_mix *mix = NULL
    cdef double[::1] out = np.empty(x.N)
    cdef double[:, ::1] individual = np.empty((x.N, K))
    _check(pmc_mixture_create(_context(), 0, K, D, &weights[0], &mu[0, 0], &inv_sigma[0, 0, 0], &log_norm[0], NULL, &mix))
    try:
        _check(pmc_mix_logpdf(mix, x.handle, &out[0], &individual[0, 0]))
    finally:
        pmc_mixture_destroy(mix)
    return np.asarray(out), np.asarray(individual)


def multi_evaluate_components(DeviceSamples x, double[::1] weights, double[:, ::1] mu, double[:, :, ::1] inv_sigma,
                              double[::1] log_norm, double[:, ::1] individual, int[::1] components):
    """multi_evaluate(x, individual=individual, components=[...]) (mixture.pyx:153-156): only the listed components'
    columns of ``individual`` are written, in place; returns None as the reference does in this mode"""
    cdef int K = mu.shape[0], D = mu.shape[1]
    cdef pmc_mix *mix = NULL
    if individual.shape[0] != x.N or individual.shape[1] != K:
        raise ValueError("individual must be N x K")
    _check(pmc_mixture_create(_context(), 0, K, D, &weights[0], &mu[0, 0], &inv_sigma[0, 0, 0], &log_norm[0], NULL, &mix))
    try:
        _check(pmc_mix_logpdf_components(mix, x.handle, &components[0] if components.shape[0] else NULL,
                                         <int>components.shape[0], &individual[0, 0]))
    finally:
        pmc_mixture_destroy(mix)


def gaussian_pmc_sums(DeviceSamples x, double[::1] weights, double[:, ::1] mu, double[:, :, ::1] inv_sigma,
                      double[::1] log_norm, double[::1] importance_weights):
    """the N-sized part of gaussian_pmc, Rao-Blackwellised (pmc.pyx:23-43, :188-222): alpha, new means, new covariances"""
    cdef int K = mu.shape[0], D = mu.shape[1]
    cdef pmc_mix *mix = NULL
    cdef double[::1] alpha = np.zeros(K)
    cdef double[:, ::1] new_mu = np.zeros((K, D))
    cdef double[:, :, ::1] new_sigma = np.zeros((K, D, D))
    _check(pmc_mixture_create(_context(), 0, K, D, &weights[0], &mu[0, 0], &inv_sigma[0, 0, 0], &log_norm[0], NULL, &mix))
    try:
        _check(pmc_pmc_update_stats(_context(), mix, x.handle, &importance_weights[0], 0, NULL, 1, &alpha[0], &new_mu[0, 0],
                                    &new_sigma[0, 0, 0], NULL, NULL, NULL))
    finally:
        pmc_mixture_destroy(mix)
    return np.asarray(alpha), np.asarray(new_mu), np.asarray(new_sigma)
