/*
 * pmc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C (IEEE fp64 throughout) restatement of the
 * pypmc Cython/numpy hot loops that the HIP kernels of this repository replace.
 * It exists to CHECK the GPU path (tests/, __graft_entry__.smoke()) and to be
 * TIMED as the CPU baseline (bench.py `cpu_baseline`).  Nothing under
 * pypmc_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  Every function below is compared in
 * tests/test_oracle_golden.py against golden vectors produced by importing the
 * reference itself (tests/golden/make_golden.py, run in the build container
 * where /root/reference is available) and against the known-answer values of
 * the reference's own unit tests.
 *
 * Each function cites the reference file:line (relative to /root/reference)
 * whose loop order it follows.  Build with -O2/-O3 and -ffp-contract=off so
 * that the operation order (no FMA contraction) equals the reference's x86-64
 * Cython build.
 *
 * The *_mt variants are the same loops with the outer sample loop split over
 * OpenMP threads (the reference itself is single threaded); they are only used
 * for the "all host cores" CPU baseline.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_TINY 2.2250738585072014e-308 /* numpy.finfo('d').tiny */

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* pypmc/tools/_linalg.pyx:10-39  bilinear_sym */
double orc_bilinear_sym(const double *matrix, const double *vector, size_t dim)
{
    double res = 0.0;
    size_t i, j;
    for (i = 0; i < dim; ++i) {
        /* diagonal contribution */
        res += vector[i] * vector[i] * matrix[i * dim + i];
        for (j = 0; j < i; ++j)
            /* off-diagonal elements come twice */
            res += 2. * vector[i] * vector[j] * matrix[i * dim + j];
    }
    return res;
}

/* pypmc/density/gauss.pyx:132-153  Gauss.multi_evaluate
 * `out` may be a strided column of an N x K matrix (mixture.pyx:144). */
void orc_gauss_multi_evaluate(const double *x, size_t N, size_t dim, const double *mu,
                              const double *inv_sigma, double log_normalization, double *out,
                              size_t out_stride)
{
    double *diff = (double *)malloc(sizeof(double) * (dim ? dim : 1));
    size_t n, i;
    for (n = 0; n < N; ++n) {
        for (i = 0; i < dim; ++i)
            diff[i] = x[n * dim + i] - mu[i];
        out[n * out_stride] = log_normalization - 0.5 * orc_bilinear_sym(inv_sigma, diff, dim);
    }
    free(diff);
}

/* pypmc/density/student_t.pyx:135-166  StudentT.multi_evaluate (operation order :159-164) */
void orc_student_t_multi_evaluate(const double *x, size_t N, size_t dim, const double *mu,
                                  const double *inv_sigma, double log_norm, double prefactor,
                                  double inv_dof, double *out, size_t out_stride)
{
    double *diff = (double *)malloc(sizeof(double) * (dim ? dim : 1));
    size_t n, i;
    for (n = 0; n < N; ++n) {
        double r;
        for (i = 0; i < dim; ++i)
            diff[i] = x[n * dim + i] - mu[i];
        r = orc_bilinear_sym(inv_sigma, diff, dim);
        r *= inv_dof;
        r += 1.;
        r = log(r);
        r *= prefactor;
        r += log_norm;
        out[n * out_stride] = r;
    }
    free(diff);
}

/* pypmc/tools/_regularize.pyx:19-55  logsumexp */
double orc_logsumexp(const double *a, const double *weights, size_t len)
{
    size_t i;
    double max_val = -DBL_MAX;
    double res = 0.0;
    for (i = 0; i < len; ++i)
        if (a[i] > max_val)
            max_val = a[i];
    for (i = 0; i < len; ++i)
        res += weights[i] * exp(a[i] - max_val);
    return log(res) + max_val;
}

/* pypmc/tools/_regularize.pyx:57-84  logsumexp2D (row-major N x K) */
void orc_logsumexp2D(const double *a, const double *weights, size_t N, size_t K, double *res)
{
    size_t n, k;
    for (n = 0; n < N; ++n) {
        double max_val = -DBL_MAX;
        double r = 0.0;
        for (k = 0; k < K; ++k)
            if (a[n * K + k] > max_val)
                max_val = a[n * K + k];
        for (k = 0; k < K; ++k)
            r += weights[k] * exp(a[n * K + k] - max_val);
        res[n] = log(r) + max_val;
    }
}

/*
 * pypmc/density/mixture.pyx:112-156  MixtureDensity.multi_evaluate
 *
 * family: 0 = Gauss components, 1 = StudentT components.
 * mu K x D, inv_sigma K x D x D, log_norm K, prefactor/inv_dof K (Student-t only).
 * individual: N x K row-major (required).  components == NULL: all K columns are
 * filled (k-outer loop, mixture.pyx:144-145) and out = logsumexp2D(individual, weights)
 * (:147-150).  components != NULL: only those columns are filled, out untouched (:153-156).
 */
void orc_mixture_multi_evaluate(int family, const double *x, size_t N, size_t K, size_t dim,
                                const double *weights, const double *mu, const double *inv_sigma,
                                const double *log_norm, const double *prefactor,
                                const double *inv_dof, double *individual, double *out,
                                const int *components, size_t ncomponents)
{
    size_t kk, ncol = components ? ncomponents : K;
    for (kk = 0; kk < ncol; ++kk) {
        size_t k = components ? (size_t)components[kk] : kk;
        if (family == 0)
            orc_gauss_multi_evaluate(x, N, dim, mu + k * dim, inv_sigma + k * dim * dim, log_norm[k],
                                     individual + k, K);
        else
            orc_student_t_multi_evaluate(x, N, dim, mu + k * dim, inv_sigma + k * dim * dim,
                                         log_norm[k], prefactor[k], inv_dof[k], individual + k, K);
    }
    if (!components && out)
        orc_logsumexp2D(individual, weights, N, K, out);
}

/* Sample-parallel variant of the above for the all-cores CPU baseline (n-outer, same
 * per-(n,k) arithmetic).  Not a statement about the reference, which has no threading. */
void orc_mixture_multi_evaluate_mt(int family, const double *x, size_t N, size_t K, size_t dim,
                                   const double *weights, const double *mu,
                                   const double *inv_sigma, const double *log_norm,
                                   const double *prefactor, const double *inv_dof,
                                   double *individual, double *out)
{
    const size_t chunk = 4096;
    long c, nchunks = (long)((N + chunk - 1) / chunk);
#pragma omp parallel for schedule(dynamic)
    for (c = 0; c < nchunks; ++c) {
        size_t n0 = (size_t)c * chunk, n1 = n0 + chunk < N ? n0 + chunk : N;
        orc_mixture_multi_evaluate(family, x + n0 * dim, n1 - n0, K, dim, weights, mu, inv_sigma,
                                   log_norm, prefactor, inv_dof, individual + n0 * K, out + n0, NULL,
                                   0);
    }
}

/*
 * pypmc/sampler/importance_sampling.py:197-215  ImportanceSampler._calculate_weights
 * with the target values and proposal values already evaluated:
 * w_i = exp(target_i - proposal_i)   (math.exp; the overflow -> OverflowError case is
 * reported through the return value: number of non-finite results produced from finite
 * exponents, which the caller turns into the same exception).
 */
size_t orc_is_weights(const double *log_target, const double *log_proposal, size_t N, double *w)
{
    size_t i, overflow = 0;
    for (i = 0; i < N; ++i) {
        double tmp = log_target[i] - log_proposal[i];
        w[i] = exp(tmp);
        if (isinf(w[i]) && !isinf(tmp))
            ++overflow;
    }
    return overflow;
}

/*
 * pypmc/sampler/importance_sampling.py:313-365  deterministic-mixture weights of run t
 * (_combine_weights_linear :313-333, _combine_weights_log :335-371).  q: N x T row-major,
 * q[n,l] = log q_l(x^t_n) (what proposals[l].multi_evaluate(samples[t]) returns); counts[l] = N_l.
 */
void orc_combine_weights(const double *q, size_t N, size_t T, const double *counts, size_t t,
                         const double *omega, double n_total, int log_scale, double *out)
{
    size_t n, l;
    if (log_scale) {
        double *lse = (double *)malloc((N ? N : 1) * sizeof(double));
        orc_logsumexp2D(q, counts, N, T, lse);                 /* :362 */
        for (n = 0; n < N; ++n) {
            double lw = log(omega[n]);                           /* :349 */
            lw += q[n * T + t];                                  /* :350 */
            lw += log(n_total);                                  /* :351 */
            lw -= lse[n];                                        /* :362 */
            out[n] = exp(lw);                                    /* :365 */
        }
        free(lse);
    } else {
        for (n = 0; n < N; ++n) {
            double den = 0.0;
            for (l = 0; l < T; ++l)
                den += counts[l] * exp(q[n * T + l]);            /* :321-322 */
            den /= n_total;                                      /* :323 */
            out[n] = exp(q[n * T + t]) * omega[n] / den;         /* :325-327 */
        }
    }
}

/* pypmc/tools/convergence.py:31-39  perp */
double orc_perp(const double *weights, size_t N)
{
    size_t i;
    double sum = 0.0, entr = 0.0;
    for (i = 0; i < N; ++i)
        sum += weights[i];
    for (i = 0; i < N; ++i) {
        double w = weights[i] / sum;
        if (w != 0.0) /* masked zeros, log(1)=0 */
            entr += w * log(w);
    }
    entr = -entr;
    return exp(entr) / (double)N;
}

/* pypmc/tools/convergence.py:67-72  ess */
double orc_ess(const double *weights, size_t N)
{
    size_t i;
    double sum = 0.0, cv = 0.0;
    for (i = 0; i < N; ++i)
        sum += weights[i];
    for (i = 0; i < N; ++i) {
        double t = (double)N * (weights[i] / sum) - 1.0;
        cv += t * t;
    }
    cv /= (double)N;
    return 1.0 / (1.0 + cv);
}

/* ------------------------------------------------------------------------------------------
 * Variational Bayes E-step (pypmc/mix_adapt/variational.pyx)
 * ---------------------------------------------------------------------------------------- */

/* variational.pyx:774-798  _update_expectation_gauss_exponent (k outer, n inner) */
void orc_vb_gauss_exponent(const double *data, size_t N, size_t dim, size_t K, const double *m,
                           const double *W, const double *beta, const double *nu, double *E)
{
    double *tmp = (double *)malloc(sizeof(double) * (dim ? dim : 1));
    size_t k, n, i;
    for (k = 0; k < K; ++k) {
        const double *Wk = W + k * dim * dim;
        for (n = 0; n < N; ++n) {
            for (i = 0; i < dim; ++i)
                tmp[i] = data[n * dim + i] - m[k * dim + i];
            E[n * K + k] = (double)dim / beta[k] + nu[k] * orc_bilinear_sym(Wk, tmp, dim);
        }
    }
    free(tmp);
}

/* variational.pyx:675-691 _update_log_rho  +  :711-755 _update_r */
void orc_vb_log_rho_r(const double *E, size_t N, size_t K, size_t dim, const double *ln_pi,
                      const double *det_ln_lambda, double *log_rho, double *r)
{
    const double dlog = (double)dim * log(2. * M_PI);
    const double tiny = ORC_TINY;
    size_t n, k;
    for (n = 0; n < N; ++n)
        for (k = 0; k < K; ++k)
            log_rho[n * K + k] = ln_pi[k] + 0.5 * (det_ln_lambda[k] - dlog - E[n * K + k]);

    for (n = 0; n < N; ++n) {
        double max = log_rho[n * K], norm, norm_inv, log_norm_inv;
        for (k = 1; k < K; ++k)
            if (log_rho[n * K + k] > max)
                max = log_rho[n * K + k];
        norm = 0.0;
        for (k = 0; k < K; ++k) {
            log_rho[n * K + k] -= max;
            r[n * K + k] = exp(log_rho[n * K + k]);
            norm += r[n * K + k];
        }
        norm_inv = 1. / norm;
        log_norm_inv = log(norm_inv);
        for (k = 0; k < K; ++k) {
            r[n * K + k] *= norm_inv;
            if (r[n * K + k] == 0.0)
                r[n * K + k] = tiny;
            log_rho[n * K + k] += log_norm_inv;
        }
    }
}

/* variational.pyx:699-709  _update_N_comp[_weighted]  (einsum 'n,nk->k' / 'nk->k';
 * numpy's einsum summation order is not part of its contract -> compare with tolerance)
 * followed by inv_N_comp = 1/regularize(N_comp) (_regularize.pyx:16). */
void orc_vb_N_comp(const double *r, const double *weights /* or NULL */, size_t N, size_t K,
                   double *N_comp, double *inv_N_comp)
{
    size_t n, k;
    for (k = 0; k < K; ++k)
        N_comp[k] = 0.0;
    for (n = 0; n < N; ++n)
        for (k = 0; k < K; ++k)
            N_comp[k] += weights ? weights[n] * r[n * K + k] : r[n * K + k];
    for (k = 0; k < K; ++k) {
        if (N_comp[k] == 0.0)
            N_comp[k] = ORC_TINY;
        inv_N_comp[k] = 1. / N_comp[k];
    }
}

/* variational.pyx:806-853  _update_x_mean_comp[_weighted] */
void orc_vb_x_mean_comp(const double *data, const double *r, const double *weights /* or NULL */,
                        size_t N, size_t K, size_t dim, const double *inv_N_comp,
                        double *x_mean_comp)
{
    size_t k, n, i;
    for (k = 0; k < K * dim; ++k)
        x_mean_comp[k] = 0.0;
    for (k = 0; k < K; ++k) {
        for (n = 0; n < N; ++n) {
            if (weights) {
                double w = weights[n] * r[n * K + k];
                for (i = 0; i < dim; ++i)
                    x_mean_comp[k * dim + i] += w * data[n * dim + i];
            } else {
                for (i = 0; i < dim; ++i)
                    x_mean_comp[k * dim + i] += r[n * K + k] * data[n * dim + i];
            }
        }
        for (i = 0; i < dim; ++i)
            x_mean_comp[k * dim + i] *= inv_N_comp[k];
    }
}

/* variational.pyx:855-932  _update_S[_weighted] */
void orc_vb_S(const double *data, const double *r, const double *weights /* or NULL */, size_t N,
              size_t K, size_t dim, const double *inv_N_comp, const double *x_mean_comp, double *S)
{
    double *tmpv = (double *)malloc(sizeof(double) * (dim ? dim : 1));
    size_t k, n, i, j;
    for (k = 0; k < K * dim * dim; ++k)
        S[k] = 0.0;
    for (k = 0; k < K; ++k) {
        double *Sk = S + k * dim * dim;
        for (n = 0; n < N; ++n) {
            for (i = 0; i < dim; ++i)
                tmpv[i] = data[n * dim + i] - x_mean_comp[k * dim + i];
            if (weights) {
                double w = weights[n] * r[n * K + k];
                for (i = 0; i < dim; ++i)
                    for (j = 0; j < i + 1; ++j)
                        Sk[i * dim + j] += w * tmpv[i] * tmpv[j];
            } else {
                for (i = 0; i < dim; ++i)
                    for (j = 0; j < i + 1; ++j)
                        Sk[i * dim + j] += r[n * K + k] * tmpv[i] * tmpv[j];
            }
        }
        for (i = 0; i < dim; ++i)
            for (j = 0; j < i + 1; ++j) {
                Sk[i * dim + j] *= inv_N_comp[k];
                Sk[j * dim + i] = Sk[i * dim + j];
            }
    }
    free(tmpv);
}

/* variational.pyx:1003-1013  _update_expectation_log_q_Z[_weighted]  (einsum '[n,]nk,nk') */
double orc_vb_expectation_log_q_Z(const double *r, const double *log_rho,
                                  const double *weights /* or NULL */, size_t N, size_t K)
{
    double res = 0.0;
    size_t n, k;
    for (n = 0; n < N; ++n) {
        double row = 0.0;
        for (k = 0; k < K; ++k)
            row += r[n * K + k] * log_rho[n * K + k];
        res += weights ? weights[n] * row : row;
    }
    return res;
}

/*
 * The N-sized part of GaussianInference.E_step (variational.pyx:116-127) in one call:
 * E, log_rho, r (N x K each, caller buffers), N_comp, inv_N_comp, x_mean_comp, S and the
 * bound term E[log q(Z)].  ln_pi / det_ln_lambda are the K-sized expectations computed on
 * the host (:759-772, :800-804).
 */
void orc_vb_estep(const double *data, size_t N, size_t dim, size_t K, const double *weights,
                  const double *m, const double *W, const double *beta, const double *nu,
                  const double *ln_pi, const double *det_ln_lambda, double *E, double *log_rho,
                  double *r, double *N_comp, double *inv_N_comp, double *x_mean_comp, double *S,
                  double *elogqz)
{
    orc_vb_gauss_exponent(data, N, dim, K, m, W, beta, nu, E);
    orc_vb_log_rho_r(E, N, K, dim, ln_pi, det_ln_lambda, log_rho, r);
    orc_vb_N_comp(r, weights, N, K, N_comp, inv_N_comp);
    orc_vb_x_mean_comp(data, r, weights, N, K, dim, inv_N_comp, x_mean_comp);
    orc_vb_S(data, r, weights, N, K, dim, inv_N_comp, x_mean_comp, S);
    if (elogqz)
        *elogqz = orc_vb_expectation_log_q_Z(r, log_rho, weights, N, K);
}

/*
 * All-cores variant for the CPU baseline: the sample range is split into chunks, each
 * thread runs the reference's loops on its chunk producing *unnormalised* chunk sums of
 * N_k, sum r x and (in a second sweep, once x_mean is known) sum r (x-xbar)(x-xbar)^T, which
 * are added in chunk order.  E/log_rho/r are still materialised N x K as the reference does.
 */
void orc_vb_estep_mt(const double *data, size_t N, size_t dim, size_t K, const double *weights,
                     const double *m, const double *W, const double *beta, const double *nu,
                     const double *ln_pi, const double *det_ln_lambda, double *E,
                     double *log_rho, double *r, double *N_comp, double *inv_N_comp,
                     double *x_mean_comp, double *S, double *elogqz)
{
    const size_t chunk = 2048;
    long c, nchunks = (long)((N + chunk - 1) / chunk);
    size_t P1 = K + K * dim, P2 = K * dim * dim;
    double *part1 = (double *)calloc((size_t)nchunks * (P1 + 1), sizeof(double));
    double *part2 = (double *)calloc((size_t)nchunks * P2, sizeof(double));
    double *ones = (double *)malloc(sizeof(double) * K);
    size_t k, i, j;
    for (k = 0; k < K; ++k)
        ones[k] = 1.0;

#pragma omp parallel for schedule(dynamic)
    for (c = 0; c < nchunks; ++c) {
        size_t n0 = (size_t)c * chunk, n1 = n0 + chunk < N ? n0 + chunk : N, nn = n1 - n0;
        double *p = part1 + (size_t)c * (P1 + 1);
        double *dummy_inv = (double *)malloc(sizeof(double) * K);
        const double *wc = weights ? weights + n0 : NULL;
        orc_vb_gauss_exponent(data + n0 * dim, nn, dim, K, m, W, beta, nu, E + n0 * K);
        orc_vb_log_rho_r(E + n0 * K, nn, K, dim, ln_pi, det_ln_lambda, log_rho + n0 * K, r + n0 * K);
        {
            size_t n, kk;
            for (n = 0; n < nn; ++n)
                for (kk = 0; kk < K; ++kk)
                    p[kk] += wc ? wc[n] * r[(n0 + n) * K + kk] : r[(n0 + n) * K + kk];
        }
        orc_vb_x_mean_comp(data + n0 * dim, r + n0 * K, wc, nn, K, dim, ones, p + K);
        p[P1] = orc_vb_expectation_log_q_Z(r + n0 * K, log_rho + n0 * K, wc, nn, K);
        free(dummy_inv);
    }
    for (k = 0; k < K; ++k)
        N_comp[k] = 0.0;
    for (k = 0; k < K * dim; ++k)
        x_mean_comp[k] = 0.0;
    if (elogqz)
        *elogqz = 0.0;
    for (c = 0; c < nchunks; ++c) {
        double *p = part1 + (size_t)c * (P1 + 1);
        for (k = 0; k < K; ++k)
            N_comp[k] += p[k];
        for (k = 0; k < K * dim; ++k)
            x_mean_comp[k] += p[K + k];
        if (elogqz)
            *elogqz += p[P1];
    }
    for (k = 0; k < K; ++k) {
        if (N_comp[k] == 0.0)
            N_comp[k] = ORC_TINY;
        inv_N_comp[k] = 1. / N_comp[k];
        for (i = 0; i < dim; ++i)
            x_mean_comp[k * dim + i] *= inv_N_comp[k];
    }
#pragma omp parallel for schedule(dynamic)
    for (c = 0; c < nchunks; ++c) {
        size_t n0 = (size_t)c * chunk, n1 = n0 + chunk < N ? n0 + chunk : N, nn = n1 - n0;
        orc_vb_S(data + n0 * dim, r + n0 * K, weights ? weights + n0 : NULL, nn, K, dim, ones,
                 x_mean_comp, part2 + (size_t)c * P2);
    }
    for (k = 0; k < P2; ++k)
        S[k] = 0.0;
    for (c = 0; c < nchunks; ++c)
        for (k = 0; k < P2; ++k)
            S[k] += part2[(size_t)c * P2 + k];
    for (k = 0; k < K; ++k)
        for (i = 0; i < dim; ++i)
            for (j = 0; j < dim; ++j)
                S[k * dim * dim + i * dim + j] *= inv_N_comp[k];
    free(part1);
    free(part2);
    free(ones);
}

/* ------------------------------------------------------------------------------------------
 * PMC (pypmc/mix_adapt/pmc.pyx)
 * ---------------------------------------------------------------------------------------- */

/*
 * pmc.pyx:23-43  calculate_rho_rb.  rho is N x K, zero on entry in the columns of dead
 * components (they take part in the row maximum of logsumexp2D with value 0 and weight 0).
 * On entry the live columns hold log q_k(x_n) (multi_evaluate(individual=rho, components=live)).
 */
void orc_rho_rb_finish(double *rho, size_t N, size_t K, const double *component_weights,
                       const int *live, size_t nlive)
{
    const double tiny = ORC_TINY;
    double *log_denominator = (double *)malloc(sizeof(double) * (N ? N : 1));
    size_t n, kk;
    orc_logsumexp2D(rho, component_weights, N, K, log_denominator);
    for (kk = 0; kk < nlive; ++kk) {
        size_t k = (size_t)live[kk];
        for (n = 0; n < N; ++n) {
            rho[n * K + k] = exp(rho[n * K + k]) * component_weights[k];
            rho[n * K + k] /= exp(log_denominator[n]) + tiny;
        }
    }
    free(log_denominator);
}

/* pmc.pyx:45-51  calculate_rho_non_rb */
void orc_rho_non_rb(double *rho, size_t N, size_t K, const int64_t *latent, const int *live,
                    size_t nlive)
{
    size_t n, kk;
    memset(rho, 0, sizeof(double) * N * K);
    for (kk = 0; kk < nlive; ++kk) {
        size_t k = (size_t)live[kk];
        for (n = 0; n < N; ++n)
            if (latent[n] == (int64_t)k)
                rho[n * K + k] = 1.;
    }
}

/* pmc.pyx:602-610  gamma_nk = (nu_k + D) / (nu_k + bilinear_sym(inv_sigma_k, x_n - mu_k)),
 * live components only; other columns untouched (np.empty in the reference). */
void orc_student_t_gamma(const double *samples, size_t N, size_t dim, size_t K, const double *mu,
                         const double *inv_sigma, const double *dof, const int *live, size_t nlive,
                         double *gamma)
{
    double *d = (double *)malloc(sizeof(double) * (dim ? dim : 1));
    size_t kk, n, i;
    for (kk = 0; kk < nlive; ++kk) {
        size_t k = (size_t)live[kk];
        for (n = 0; n < N; ++n) {
            for (i = 0; i < dim; ++i) {
                d[i] = samples[n * dim + i];
                d[i] -= mu[k * dim + i];
            }
            gamma[n * K + k] =
                (dof[k] + (double)dim) / (dof[k] + orc_bilinear_sym(inv_sigma + k * dim * dim, d, dim));
        }
    }
    free(d);
}

/*
 * The reductions of gaussian_pmc (pmc.pyx:188-222) and student_t_pmc (:612-652):
 *   alpha_k (unnormalised) = sum_n [w_n] rho_nk                       ('n,nk->k')
 *   mu_k = sum_n [w_n] rho_nk [gamma_nk] x_n / regularize(norm_k)      ('n,nk,[nk,]ni->ki')
 *          norm_k = alpha_k (Gauss) or sum_n [w_n] rho_nk gamma_nk (Student-t, :621-623)
 *   cov_k = sum_n [w_n] rho_nk [gamma_nk] (x_n-mu_k)(x_n-mu_k)^T / regularize(alpha_k)
 *          (live components only; normalised by alpha_k also in the Student-t case, :629-630)
 * einsum's summation order is unspecified -> tolerance compare.  gamma == NULL: Gaussian.
 * Outputs: alpha_unnorm K, mu K x D (all K as in the reference), cov K x D x D (live only).
 */
void orc_pmc_reductions(const double *samples, size_t N, size_t dim, size_t K, const double *rho,
                        const double *gamma /* or NULL */, const double *weights /* or NULL */,
                        const int *live, size_t nlive, double *alpha_unnorm, double *mu, double *cov)
{
    size_t n, k, i, j, kk;
    double *norm = (double *)calloc(K ? K : 1, sizeof(double));
    double *d = (double *)malloc(sizeof(double) * (dim ? dim : 1));
    for (k = 0; k < K; ++k)
        alpha_unnorm[k] = 0.0;
    for (k = 0; k < K * dim; ++k)
        mu[k] = 0.0;
    for (n = 0; n < N; ++n)
        for (k = 0; k < K; ++k) {
            double w = weights ? weights[n] * rho[n * K + k] : rho[n * K + k];
            double wg;
            alpha_unnorm[k] += w;
            if (gamma) {
                /* dead columns of gamma are uninitialised in the reference; rho is 0 there */
                wg = (rho[n * K + k] != 0.0) ? w * gamma[n * K + k] : 0.0;
            } else
                wg = w;
            norm[k] += wg;
            for (i = 0; i < dim; ++i)
                mu[k * dim + i] += wg * samples[n * dim + i];
        }
    for (k = 0; k < K; ++k) {
        double nk = norm[k] == 0.0 ? ORC_TINY : norm[k];
        for (i = 0; i < dim; ++i)
            mu[k * dim + i] *= 1. / nk;
    }
    for (kk = 0; kk < nlive; ++kk) {
        double inv_alpha;
        double *ck;
        k = (size_t)live[kk];
        ck = cov + k * dim * dim;
        for (i = 0; i < dim * dim; ++i)
            ck[i] = 0.0;
        for (n = 0; n < N; ++n) {
            double w = weights ? weights[n] * rho[n * K + k] : rho[n * K + k];
            if (gamma)
                w *= gamma[n * K + k];
            for (i = 0; i < dim; ++i)
                d[i] = samples[n * dim + i] - mu[k * dim + i];
            for (i = 0; i < dim; ++i)
                for (j = 0; j < dim; ++j)
                    ck[i * dim + j] += w * d[i] * d[j];
        }
        inv_alpha = 1. / (alpha_unnorm[k] == 0.0 ? ORC_TINY : alpha_unnorm[k]);
        for (i = 0; i < dim * dim; ++i)
            ck[i] *= inv_alpha;
    }
    free(norm);
    free(d);
}

/*
 * orc_pmc_reductions with the COMPONENTS spread over the host's cores (a checker for full-size GPU runs; not a
 * statement about the reference, which has no threading).  Every component's sums are taken over the samples in the
 * same order as above, so the results are bit-identical to orc_pmc_reductions.
 */
void orc_pmc_reductions_mt(const double *samples, size_t N, size_t dim, size_t K, const double *rho,
                           const double *gamma /* or NULL */, const double *weights /* or NULL */,
                           const int *live, size_t nlive, double *alpha_unnorm, double *mu, double *cov)
{
    long kq;
#pragma omp parallel for schedule(dynamic)
    for (kq = 0; kq < (long)K; ++kq) {
        size_t k = (size_t)kq, n, i, j, kk;
        double norm = 0.0, nk;
        double *d = (double *)malloc(sizeof(double) * (dim ? dim : 1));
        int is_live = 0;
        alpha_unnorm[k] = 0.0;
        for (i = 0; i < dim; ++i)
            mu[k * dim + i] = 0.0;
        for (n = 0; n < N; ++n) {
            double w = weights ? weights[n] * rho[n * K + k] : rho[n * K + k];
            double wg;
            alpha_unnorm[k] += w;
            if (gamma)
                wg = (rho[n * K + k] != 0.0) ? w * gamma[n * K + k] : 0.0;
            else
                wg = w;
            norm += wg;
            for (i = 0; i < dim; ++i)
                mu[k * dim + i] += wg * samples[n * dim + i];
        }
        nk = norm == 0.0 ? ORC_TINY : norm;
        for (i = 0; i < dim; ++i)
            mu[k * dim + i] *= 1. / nk;
        for (kk = 0; kk < nlive; ++kk)
            if ((size_t)live[kk] == k)
                is_live = 1;
        if (is_live) {
            double inv_alpha;
            double *ck = cov + k * dim * dim;
            for (i = 0; i < dim * dim; ++i)
                ck[i] = 0.0;
            for (n = 0; n < N; ++n) {
                double w = weights ? weights[n] * rho[n * K + k] : rho[n * K + k];
                if (gamma)
                    w *= gamma[n * K + k];
                for (i = 0; i < dim; ++i)
                    d[i] = samples[n * dim + i] - mu[k * dim + i];
                for (i = 0; i < dim; ++i)
                    for (j = 0; j < dim; ++j)
                        ck[i * dim + j] += w * d[i] * d[j];
            }
            inv_alpha = 1. / (alpha_unnorm[k] == 0.0 ? ORC_TINY : alpha_unnorm[k]);
            for (i = 0; i < dim * dim; ++i)
                ck[i] *= inv_alpha;
        }
        free(d);
    }
}

/*
 * pmc.pyx:654-691  the N-sized part of the degree-of-freedom condition:
 * for live k:  c_k = 1 - (sum_n [w_n] (xi+delta)_nk) / weight_normalization   with
 * (xi+delta)_nk = rho (log(.5 (b+nu)) - psi(.5 (D+nu))) + (1-rho)(log(.5 nu) - psi(.5 nu))
 *                 + rho (D+nu)/(b+nu) + (1-rho),   b = bilinear_sym(inv_sigma_k, x_n - mu_k).
 * The two digamma values depend on k only and are passed in (psi_half_dim_nu, psi_half_nu);
 * the reference evaluates scipy's digamma per (n,k) with these same arguments.
 */
void orc_student_t_dof_const(const double *samples, size_t N, size_t dim, size_t K,
                             const double *rho, const double *weights /* or NULL */,
                             double weight_normalization, const double *mu,
                             const double *inv_sigma, const double *dof,
                             const double *psi_half_dim_nu, const double *psi_half_nu,
                             const int *live, size_t nlive, double *nu_condition_const)
{
    double *d = (double *)malloc(sizeof(double) * (dim ? dim : 1));
    size_t kk, n, i;
    for (kk = 0; kk < nlive; ++kk) {
        size_t k = (size_t)live[kk];
        double nu = dof[k], acc = 0.0;
        for (n = 0; n < N; ++n) {
            double bil, g, rh = rho[n * K + k];
            for (i = 0; i < dim; ++i) {
                d[i] = samples[n * dim + i];
                d[i] -= mu[k * dim + i];
            }
            bil = orc_bilinear_sym(inv_sigma + k * dim * dim, d, dim);
            g = log(.5 * (bil + nu));
            g -= psi_half_dim_nu[k];
            g *= rh;
            g += (1. - rh) * (log(.5 * nu) - psi_half_nu[k]);
            g += rh * ((double)dim + nu) / (bil + nu);
            g += (1. - rh);
            acc += weights ? g * weights[n] : g;
        }
        acc /= weight_normalization;
        nu_condition_const[k] = 1. - acc;
    }
    free(d);
}

/* pmc.pyx:388-391  PMC.log_likelihood = sum_n wbar_n log q(x_n)  (wbar = 1/N if unweighted) */
double orc_pmc_log_likelihood(const double *log_q, const double *normalized_weights /* or NULL */,
                              size_t N)
{
    double s = 0.0;
    size_t n;
    if (normalized_weights) {
        for (n = 0; n < N; ++n)
            s += log_q[n] * normalized_weights[n];
        return s;
    }
    for (n = 0; n < N; ++n)
        s += log_q[n];
    return s / (double)N;
}
