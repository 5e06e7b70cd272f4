/* The handle layer of libpmc_hip.so (include/pmc_ctx.h) from plain C: compiled with gcc, no HIP headers, no device
 * pointers -- what a Cython binding of pypmc's .pyx files sees.
 *
 *   gcc -O2 -std=c99 -I include examples/ctx_demo.c -L pypmc_amd/lib -lpmc_hip -Wl,-rpath,$PWD/pypmc_amd/lib -lm -o ctx_demo
 *   ./ctx_demo [N]
 *
 * One PMC iteration of a 3-component Gaussian proposal in D = 4 against a target mixture, the way pypmc's
 * examples/pmc.py:61-65 does it -- propose, importance weights, perplexity, Rao-Blackwellised update -- and the VB
 * E-step of the same samples.  The same fixed parameters and lattice samples as examples/cabi_demo.cpp;
 * tests/test_gpu_cabi_demo.py compares the printed numbers with the oracle.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pmc_ctx.h"

#define CHECK(x)                                                        \
    do {                                                                \
        if ((x) != 0) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, pmc_last_error());          \
            return 3;                                                   \
        }                                                               \
    } while (0)

#define D 4
#define K 3

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 1000;
    const double pi = 3.14159265358979323846;
    const double mu[K * D] = {0, 0, 0, 0, 2, -1, 0.5, 1, -3, 2, 1, -1};
    const double var[K * D] = {1, 2, 0.5, 1, 0.3, 0.7, 1.1, 2.0, 1.5, 0.4, 0.9, 1.2};
    const double weight[K] = {0.5, 0.3, 0.2};
    double prec[K * D * D] = {0}, tprec[K * D * D] = {0}, log_norm[K], tln[K];
    for (int k = 0; k < K; ++k) {
        double ld = 0.0;
        for (int i = 0; i < D; ++i) {
            prec[(k * D + i) * D + i] = 1.0 / var[k * D + i];
            tprec[(k * D + i) * D + i] = 1.0;
            ld += log(var[k * D + i]);
        }
        log_norm[k] = -0.5 * D * log(2.0 * pi) - 0.5 * ld;
        tln[k] = -0.5 * D * log(2.0 * pi);
    }
    double *x = (double *)malloc(sizeof(double) * (size_t)N * D);
    for (int64_t n = 0; n < N; ++n)
        for (int i = 0; i < D; ++i) x[n * D + i] = sin(0.37 * (double)n + 1.3 * i) * 3.0;

    pmc_ctx *ctx;
    pmc_mix *proposal, *target;
    pmc_samples *samples;
    CHECK(pmc_init(0, &ctx));
    CHECK(pmc_mixture_create(ctx, PMC_KIND_GAUSS, K, D, weight, mu, prec, log_norm, NULL, &proposal));
    CHECK(pmc_mixture_create(ctx, PMC_KIND_GAUSS, K, D, weight, mu, tprec, tln, NULL, &target));
    CHECK(pmc_samples_upload(ctx, x, N, D, &samples));

    /* MixtureDensity.multi_evaluate */
    double *logq = (double *)malloc(sizeof(double) * (size_t)N);
    CHECK(pmc_mix_logpdf(proposal, samples, logq, NULL));
    for (int n = 0; n < 5 && n < N; ++n) printf("logq %d %.17g\n", n, logq[n]);

    /* ImportanceSampler._calculate_weights + perp */
    double sums[3];
    CHECK(pmc_is_weights(proposal, samples, NULL, target, NULL, NULL, sums));
    printf("sums %.17g %.17g %.17g\n", sums[0], sums[1], sums[2]);
    printf("perplexity %.17g\n", exp(-(sums[1] / sums[0] - log(sums[0]))) / (double)N);

    /* gaussian_pmc's N-sized part, with the importance weights that stayed on the device */
    double alpha[K], nmu[K * D], nsig[K * D * D], loglik, norm;
    CHECK(pmc_pmc_update_stats(ctx, proposal, samples, NULL, 1, NULL, 1, alpha, nmu, nsig, NULL, &loglik, &norm));
    for (int k = 0; k < K; ++k) printf("alpha %d %.17g\n", k, alpha[k]);
    for (int k = 0; k < K; ++k) printf("mu0 %d %.17g\n", k, nmu[k * D]);
    for (int k = 0; k < K; ++k) printf("sigma00 %d %.17g\n", k, nsig[k * D * D]);
    printf("loglik %.17g norm %.17g\n", loglik, norm);

    /* GaussianInference.E_step: W = precision / nu, the other expectations as the object would hold them */
    double W[K * D * D], nu[K], beta[K], ln_pi[K], ln_lambda[K], N_k[K], xbar[K * D], S[K * D * D], elogqz;
    for (int k = 0; k < K; ++k) {
        nu[k] = D + 2.0 + k;
        beta[k] = 1.0 + k;
        ln_pi[k] = log(weight[k]);
        ln_lambda[k] = 0.25 * k;
        for (int i = 0; i < D * D; ++i) W[k * D * D + i] = prec[k * D * D + i] / nu[k];
    }
    CHECK(pmc_vb_estep(ctx, samples, NULL, K, mu, W, nu, beta, ln_pi, ln_lambda, NULL, N_k, xbar, S, &elogqz, NULL, NULL));
    for (int k = 0; k < K; ++k) printf("N_comp %d %.17g\n", k, N_k[k]);
    for (int k = 0; k < K; ++k) printf("S00 %d %.17g\n", k, S[k * D * D]);
    printf("elogqz %.17g\n", elogqz);

    CHECK(pmc_samples_free(samples));
    CHECK(pmc_mixture_destroy(proposal));
    CHECK(pmc_mixture_destroy(target));
    CHECK(pmc_shutdown(ctx));
    free(logq);
    free(x);
    printf("done\n");
    return 0;
}
