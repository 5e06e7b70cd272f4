/*
 * pmc_hip.h -- C ABI of libpmc_hip.so, the MI355X (gfx950) adaptive-importance-sampling core.
 *
 * The reference (pypmc 1.2.6) has no FFI layer: its hot loops are Cython functions and methods
 * called from Python.  This header declares, one entry point per reference loop nest, what a
 * binding for that path has to call.  Citations are file:line under the reference tree.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer prefixed d_ is a DEVICE pointer (HIP), every
 *     pointer prefixed h_ is a HOST pointer; `stream` is a hipStream_t passed as void*.
 *   - all arithmetic is IEEE fp64 (the reference uses `double` memoryviews throughout).
 *   - every function returns 0 on success, a negative PMC_E* code on failure and never throws or
 *     aborts; pmc_last_error() returns a thread-local message for the last failure.
 *   - launches are asynchronous on `stream`; outputs are valid after the stream is synchronised.
 *   - results are deterministic run-to-run (fixed reduction trees, no floating-point atomics).
 *   - all N-sized and K-sized device memory is the caller's (pmc_workspace_bytes); the library keeps only
 *     20 KB of device scratch per (device, stream) for its finishing reduction and block tickets and a pool of HIP events
 *     while timing is enabled.  Calls on different streams may run concurrently if they are given
 *     different workspaces.
 *
 * Component parameter pack ("pack")
 *   The kernels read mixture parameters through the scalar cache from one flat fp64 array built on
 *   the host by pmc_pack_components().  Per component (stride pmc_pack_stride(Dp) doubles):
 *     [0,Dp)            mean / shift  mu_k
 *     [Dp,Dp+T)         R_k, upper triangular, row-major (i, j>=i), T=Dp(Dp+1)/2, with
 *                       precision_k = R_k^T R_k  so that  (x-mu)^T precision (x-mu) = |R (x-mu)|^2
 *                       (replaces bilinear_sym(inv_sigma, x-mu), pypmc/tools/_linalg.pyx:10-39)
 *     [Dp+T+0..3]       c0..c3, kind specific (see pmc_kind)
 *     [Dp+T+4]          linear component weight w_k (mixture.pyx:147, _regularize.pyx:72-81)
 *     [Dp+T+5]          output column of this component in N x ld row-major outputs
 *   Dp = pmc_padded_dim(D) >= D is the compiled kernel dimension used for D.
 */
#ifndef PMC_HIP_H
#define PMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMC_ABI_VERSION 2

enum pmc_status {
    PMC_OK = 0,
    PMC_EINVAL = -1,       /* bad argument (shape, NULL, unsupported dimension) */
    PMC_ENOTPOSDEF = -2,   /* a precision matrix is not positive definite */
    PMC_EHIP = -3,         /* HIP runtime error (message has hipGetErrorString) */
    PMC_ENODEVICE = -4     /* no gfx950 device / kernels not loadable */
};

/* How the per-(sample, component) value a_nk is formed from maha_nk = |R_k (x_n - mu_k)|^2 */
enum pmc_kind {
    /* Gauss.multi_evaluate, pypmc/density/gauss.pyx:146-151:  a = c0 - 0.5*maha,  c0 = log_normalization */
    PMC_KIND_GAUSS = 0,
    /* StudentT.multi_evaluate, pypmc/density/student_t.pyx:159-164 (same operation order):
     *   a = c0 + c1*log(1 + maha*c2),  c0 = log_norm, c1 = -(dof+D)/2, c2 = 1/dof, c3 = dof */
    PMC_KIND_STUDENT_T = 1,
    /* GaussianInference, pypmc/mix_adapt/variational.pyx:798 and :691 (same operation order):
     *   E = c0 + c1*maha,  a = c2 + 0.5*(c3 - E);
     *   c0 = D/beta_k, c1 = nu_k, c2 = E[ln pi_k], c3 = E[ln|Lambda_k|] - D*log(2 pi) */
    PMC_KIND_VB = 2
};

/* What pmc_responsibilities() turns a row of a_nk into */
enum pmc_resp_mode {
    /* variational.pyx:728-755 _update_r: r = softmax_k(a), zeros -> tiny, log_rho normalised */
    PMC_RESP_VB = 0,
    /* pmc.pyx:23-43 calculate_rho_rb: rho = exp(a) w_k / (exp(logsumexp) + tiny) */
    PMC_RESP_PMC_RB = 1,
    /* pmc.pyx:45-51 calculate_rho_non_rb: rho = [latent_n == column_k] */
    PMC_RESP_PMC_LATENT = 2
};

/* ---- library / device ------------------------------------------------------------------- */
int pmc_abi_version(void);
const char *pmc_last_error(void);
/* number of visible HIP devices, or a negative status */
int pmc_device_count(void);
/* writes the gcnArchName of `device` (e.g. "gfx950:sramecc+:xnack-") into buf */
int pmc_device_arch(int device, char *buf, size_t buflen);

/* ---- dimensions ---------------------------------------------------------------------------- */
/* largest supported sample dimension (1024).  Up to pmc_max_compiled_dim() (64) the kernels are compiled per
   dimension, fully unrolled, with a sample's coordinates in registers; beyond it the run-time-dimension unit
   takes over (the reference's loops take any length, pypmc/tools/_linalg.pyx:32-37): the Mahalanobis forms and
   the second moments as 16 x 16 x 4 fp64 MFMA tiles, everything else the same kernels.  That unit keeps one
   scratch of its own: pmc_mixture_logpdf / pmc_importance_weights hold the forms the caller does not keep
   (d_maha_tiles) in a stream-ordered allocation of at most 256 MiB (hipMallocAsync / hipFreeAsync on the caller's
   stream) and walk the samples in chunks that fit it (pmc_configure("big_dim_scratch_bytes", ...) moves the bound). */
int pmc_max_dim(void);
int pmc_max_compiled_dim(void);
/* compiled kernel dimension used for D (>= D; D itself beyond pmc_max_compiled_dim()), or PMC_EINVAL if D is
   unsupported */
int pmc_padded_dim(int D);
/* doubles per component in a pack for sample dimension D */
int64_t pmc_pack_stride(int D);
/* samples per tile of the internal tile-major N x K layout (64 = one wavefront) */
int pmc_tile(void);

/* ---- host-side parameter preparation ------------------------------------------------------- */
/*
 * Build the pack for K components on the HOST (Cholesky factor of each precision matrix,
 * padding to the compiled dimension).  precision is K x D x D row-major symmetric positive
 * definite: Gauss/StudentT `inv_sigma` (gauss.pyx:112, student_t.pyx:111) or the VB `W`
 * (variational.pyx:783).  c0..c3 and weight are K vectors (c* may be NULL = 0, weight NULL = 1),
 * column is a K vector of output columns (NULL = 0..K-1).  h_pack receives K*pmc_pack_stride(D)
 * doubles.  Returns PMC_ENOTPOSDEF (and names the component) if a factorisation fails.
 */
int pmc_pack_components(int K, int D, const double *h_mu, const double *h_precision,
                        const double *h_c0, const double *h_c1, const double *h_c2,
                        const double *h_c3, const double *h_weight, const int32_t *h_column,
                        double *h_pack);

/*
 * A pack that carries only means / shifts: what pmc_sufficient_stats reads of its pack (the moments are taken
 * about d = x - mu_k).  No matrices are built or factorised -- the second pass of an update whose means turned
 * out far from the shift (pypmc_amd.mix_adapt._stats.shift_is_far; the reference takes its moments about the
 * mean it has just computed, variational.pyx:806-932, pmc.pyx:188-222) only needs K x D numbers.  The triangular
 * factor and c0..c3 are zero, weight 1, column k: such a pack must not be handed to the log-pdf / responsibility
 * kernels.  h_pack receives K*pmc_pack_stride(D) doubles.
 */
int pmc_pack_means(int K, int D, const double *h_mu, double *h_pack);

/* ---- streams ---------------------------------------------------------------------------------- */
/*
 * The library keeps 20 KB of device scratch per (device, stream) it has launched a finishing reduction on -- at most
 * 256 such slots.  A caller that creates and destroys streams over its lifetime calls this (current device = the
 * stream's) once the stream is idle, before destroying it: the slot is handed to the next new stream.  Needed for
 * correctness too: a new stream may receive the handle value of a destroyed one.  Callers with a fixed set of streams
 * (PyTorch's stream pool) never need it.
 */
int pmc_stream_release(void *stream);

/* ---- workspace ------------------------------------------------------------------------------ */
/* bytes of device scratch the calls below need for N samples, K components, dimension D */
int64_t pmc_workspace_bytes(int64_t N, int K, int D);
/* doubles in one tile-major N x K buffer (d_u, d_scratch): ceil(N/64)*K*64 */
int64_t pmc_tile_buffer_len(int64_t N, int K);
/* doubles per component in the statistics vector: 1 + D + D(D+1)/2 */
int64_t pmc_stats_stride(int D);

/* ---- mixture log-pdf and importance weights ---------------------------------------------- */
/*
 * MixtureDensity.multi_evaluate (pypmc/density/mixture.pyx:112-156) with the component loops
 * (gauss.pyx:146-151 / student_t.pyx:154-164) and logsumexp2D (_regularize.pyx:57-84) fused
 * into one pass over the samples.
 *
 *   d_x           N x D row-major samples
 *   d_pack        K components (kind PMC_KIND_GAUSS or PMC_KIND_STUDENT_T)
 *   max_init_zero 0: row maximum starts at -DBL_MAX (_regularize.pyx:73);
 *                 1: it starts at 0.0 -- the mixture has dead (zero weight, all-zero column)
 *                    components that take part in the maximum (pmc.pyx:24-34)
 *   d_out         N, log q(x_n) = log sum_k w_k exp(a_nk)           (NULL: not wanted)
 *   d_individual  N x ld row-major, a_nk written to column column_k (NULL: not wanted);
 *                 with d_out == NULL this is the `components=` subset mode (mixture.pyx:153-156)
 *   d_log_target  N, log P(x_n) (NULL: no importance weights).  If given:
 *   d_weights     N, w_n = exp(log_target_n - log q_n) (importance_sampling.py:197-215)
 *   d_sample_w    N, optional weights for the log-likelihood sum (NULL = 1)
 *   d_scalars     8 doubles (NULL: not wanted):
 *                 [0] sum w  [1] sum w*log w (zeros masked, convergence.py:31-39)  [2] sum w^2
 *                 [3] sum sample_w_n * log q_n (pmc.pyx:388-391)  [4] number of non-finite w_n
 *                 produced from a finite exponent (math.exp overflow, importance_sampling.py:207)
 *                 [5..7] reserved (0)
 *   d_workspace   pmc_workspace_bytes(N,K,D) bytes
 */
int pmc_mixture_logpdf(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                       int max_init_zero, double *d_out, double *d_individual, int64_t ld,
                       const double *d_log_target, double *d_weights, const double *d_sample_w,
                       double *d_scalars, void *d_workspace, void *stream);

/*
 * ImportanceSampler._calculate_weights (pypmc/sampler/importance_sampling.py:197-215) when the target
 * is itself a mixture density (`target = mixture.evaluate`, pypmc/examples/pmc.py:32-53): log P(x_n) from
 * d_target_pack, log q(x_n) from d_pack, w_n = exp(log P - log q) and the sums of pmc_mixture_logpdf in
 * ONE pass over the samples, for any combination of proposal and target families (kind / target_kind =
 * PMC_KIND_GAUSS or PMC_KIND_STUDENT_T).  Bitwise the same numbers as pmc_mixture_logpdf(target)
 * followed by pmc_mixture_logpdf(proposal, d_log_target).
 *   d_out             N, log q(x_n)   (NULL: not wanted)
 *   d_log_target_out  N, log P(x_n)   (NULL: not wanted; ImportanceSampler's target_values)
 *   d_weights         N
 *   d_scalars, d_workspace, d_sample_w as for pmc_mixture_logpdf (workspace for max(K, K_target))
 */
int pmc_importance_weights(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                           const double *d_target_pack, int K_target, int target_kind, double *d_out,
                           double *d_log_target_out, double *d_weights, const double *d_sample_w,
                           double *d_scalars, void *d_workspace, void *stream);

/*
 * perp / ess sums over a weight vector (pypmc/tools/convergence.py:31-39, :67-72):
 * d_scalars[0..2] = sum w, sum w log w (zeros masked), sum w^2.
 */
int pmc_weight_sums(const double *d_w, int64_t N, double *d_scalars, void *d_workspace,
                    void *stream);

/*
 * logsumexp2D of an existing N x K row-major matrix (pypmc/tools/_regularize.pyx:57-84), used by
 * combine_weights (importance_sampling.py:362) and by mixtures of foreign component types:
 * d_out[n] = max_k a[n,k] + log sum_k w_k exp(a[n,k] - max).
 */
int pmc_logsumexp2d(const double *d_a, const double *d_w, int64_t N, int K, double *d_out,
                    void *stream);

/*
 * Deterministic-mixture weights of run t out of T importance-sampling runs [Cor+12]
 * (pypmc/sampler/importance_sampling.py:313-365).  d_q: T x N row-major, d_q[l*N + n] = log q_l(x^t_n)
 * (row l is the d_out of pmc_mixture_logpdf for proposal l); d_counts[l] = N_l as double;
 * d_omega[n] = ordinary importance weight of x^t_n; n_total = sum_l N_l.
 *   log_scale != 0 (_combine_weights_log, :337-365; needs omega > 0):
 *       d_out[n] = exp(log omega_n + q[t,n] + log n_total - logsumexp2D(q[:,n], counts))
 *   log_scale == 0 (_combine_weights_linear, :313-333):
 *       d_out[n] = exp(q[t,n]) * omega_n / ((sum_l N_l exp(q[l,n])) / n_total)
 * d_flag (1 double, may be NULL) counts the non-finite results (the reference's final assert, :310).
 */
int pmc_combine_weights(const double *d_q, int64_t N, int T, const double *d_counts, int t,
                        const double *d_omega, double n_total, int log_scale, double *d_out,
                        double *d_flag, void *stream);

/* ---- proposing ------------------------------------------------------------------------------- */
/*
 * Device side of MixtureDensity.propose(N, rng, trace=True, shuffle=False)
 * (pypmc/density/mixture.pyx:159-212; Gauss.propose gauss.pyx:159-163; StudentT.propose
 * student_t.pyx:172-176).  The component counts come from the caller's generator on the host
 * (rng.multinomial, mixture.pyx:192 -- counts and origin indices stay bit-exact); d_offsets holds
 * their K+1 exclusive prefix sums.  Sample n (ordered by component) becomes
 * mu_k + L_k z [* sqrt(nu_k / chi2)], L_k = lower Cholesky factor of sigma_k (row-major D x D),
 * with Philox4x32-10 random numbers keyed by `seed` and counted by the GLOBAL sample index
 * first_sample + n, so shards of one logical run on several GPUs draw disjoint streams.
 * d_dof == NULL: Gaussian components.  d_origin (N int64, may be NULL) receives k per sample.
 */
int pmc_propose(const double *d_mu, const double *d_chol, const double *d_dof, const int64_t *d_offsets,
                int K, int D, int64_t N, int64_t first_sample, uint64_t seed, double *d_x,
                int64_t *d_origin, void *stream);

/* ---- responsibilities ------------------------------------------------------------------------ */
/*
 * The N x K responsibility matrix of the VB E-step (variational.pyx:774-798 exponent,
 * :675-691 log_rho, :711-757 r) or of the PMC update (pmc.pyx:23-51, and for Student-t
 * gamma_nk = (nu_k+D)/(nu_k+maha_nk), pmc.pyx:602-610), produced directly in the tile-major layout
 * the statistics kernel consumes:  buffer[(tile*K + k)*64 + lane], sample n = tile*64 + lane.
 *
 *   d_pack       K live components; kind VB with mode PMC_RESP_VB, GAUSS/STUDENT_T otherwise
 *   d_sample_w   N sample weights (NULL = 1): VB `self.weights` (variational.pyx:94) / importance
 *                weights (pmc.pyx:188)
 *   d_latent     N int64 generating component per sample (mode PMC_RESP_PMC_LATENT only)
 *   d_u          tile-major: sample_w*r (VB), sample_w*rho (Gauss PMC), sample_w*rho*gamma (Student-t)
 *   d_scratch    tile-major scratch, Student-t PMC only (else NULL)
 *   d_vsums      K x 2 doubles, Student-t PMC only (else NULL):  [k][0] = sum_n sample_w*rho_nk
 *                (pmc.pyx:612),  [k][1] = sum_n sample_w*rho_nk*log(0.5*(maha_nk+nu_k)) (the
 *                N-sized part of the degree-of-freedom condition, pmc.pyx:659-679)
 *   d_r, d_log_rho, d_exponent   optional N x ld row-major public matrices (NULL: not wanted):
 *                r / rho; normalised log_rho (VB); expectation_gauss_exponent (VB)
 *   d_scalars    8 doubles: [0] VB: sum_n sample_w sum_k r log_rho (variational.pyx:1003-1013);
 *                [3] PMC: sum_n sample_w log q_n;  others 0
 */
int pmc_responsibilities(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                         int mode, int max_init_zero, const double *d_sample_w,
                         const int64_t *d_latent, double *d_u, double *d_scratch, double *d_vsums,
                         double *d_r, double *d_log_rho, double *d_exponent, int64_t ld,
                         double *d_scalars, void *d_workspace, void *stream);

/* ---- sufficient statistics ------------------------------------------------------------------- */
/*
 * One pass over the samples accumulating, per component k and with d = x_n - mu_k (the pack's
 * shift):  sum_n u_nk | sum_n u_nk d (D) | sum_n u_nk d d^T (lower triangle, row-major i, j<=i)
 *     ->  d_stats[k*pmc_stats_stride(D) + ...]
 * These are the shifted, un-normalised forms of N_k, x-bar_k, S_k (variational.pyx:699-932) and
 * of alpha_k, mu_k, Sigma_k (pmc.pyx:188-222, :612-652); the K-sized conversion to the
 * reference's centred conventions is done by the caller.  With several GPUs each rank calls this
 * on its shard and the ranks all-reduce (sum) d_stats -- the only cross-GPU exchange of the path.
 */
int pmc_sufficient_stats(const double *d_x, int64_t N, int D, const double *d_pack, int K,
                         const double *d_u, double *d_stats, void *d_workspace, void *stream);

/* ---- the E-step in one call ---------------------------------------------------------------------- */
/*
 * GaussianInference.E_step (pypmc/mix_adapt/variational.pyx:116-127) / the N-sized part of gaussian_pmc
 * and student_t_pmc (pypmc/mix_adapt/pmc.pyx:53-118, :188-222, :602-691): responsibilities AND
 * sufficient statistics of the samples, d_stats as pmc_sufficient_stats and d_scalars / d_vsums as
 * pmc_responsibilities produce them -- without the public N x K matrices.
 *
 * For small sample dimensions (pmc_estep_is_fused() != 0: compiled dimension <= 7 and K <= 32 -- K <= 64 at
 * D = 1, K >= 9 from D = 5 on, where the two kernels are faster below -- VB, Gaussian Rao-Blackwell PMC, or
 * Student-t Rao-Blackwell PMC at D = 3 ... 7)
 * ONE kernel does both and the N x K responsibilities never leave the compute units: d_u and d_scratch may
 * then be NULL.  Ask pmc_estep_is_fused(), do not re-derive the rule.  Otherwise the call is
 * pmc_responsibilities followed by pmc_sufficient_stats through d_u (and d_scratch / d_vsums for Student-t).
 *
 * From 21 components on (groups of 32 at least 63 % full; compiled dimensions 8 ... 64; N * ceil(K / 32) >= 524288)
 * the statistics half first runs in its
 * component x monomial form: moments about ONE shift c common to all components (the midrange of their means) are a
 * plain matrix product U^T Z on v_mfma_f64_16x16x4_f64, re-centred to the components' own shifts on the device.  A
 * component whose weighted mean turns out further than sqrt(limit) of its own standard deviations from c (default
 * limit 1000: at most ~3 of the 16 digits lost in the re-centring) sends the call back to the per-component-shift
 * kernel of pmc_sufficient_stats, on the device, without a host round trip.  pmc_configure() moves both knobs:
 *   "stats_common_shift_min_k"  (default 17; a huge value switches the form off)
 *   "stats_common_shift_min_fill" (default 0.63: K >= 0.63 * 32 * ceil(K / 32) -- a group of 32 components costs
 *                                the same however few it holds, so K = 33 ... 40 stays with the per-component kernel)
 *   "stats_common_shift_min_n"  (default 524288 samples per 32 components: below, the form's three extra launches
 *                                cost more than it saves; never below 16384 samples)
 *   "stats_common_shift_limit"  (default 1000; 0 switches the form off)
 *   "estep_grouped_responsibilities" (default 1): with the common-shift statistics the responsibility half can run in
 *                                groups of 16 components -- every u_nk written once, nothing parked in HBM, no
 *                                normalisation pass; the per-(sample, group) factors w_n exp(M_g - M) / s are left to the
 *                                statistics kernel, which multiplies its weight operand with them (VB and Gaussian
 *                                Rao-Blackwell PMC).  0 never, 1 where it is faster (compiled D <= 16, 20, 24, 32, 40;
 *                                D = 30 from K = 64 on), 2 always.  The workspace holds the factors (8 ceil(K/16) bytes per sample).
 * pmc_sufficient_stats itself always takes its moments about the pack's own shifts.
 *
 * Two things a caller should know about these large-batch forms (verdict r4):
 *   1. The selection depends on the BATCH SIZE (grouped responsibilities / common-shift statistics from
 *      N * ceil(K / 32) >= 524288, the matrix-product Mahalanobis forms from N >= 49152, blocks in pieces below
 *      "split_max_rounds" rounds of the chip), and each form agrees with the
 *      exact kernels to ~1e-11 relative, not bit for bit.  Results are bit-reproducible from run to run for the SAME
 *      shard sizes; a rank or device count that moves a shard across a threshold changes low-order bits of the
 *      statistics (well inside the 1e-10 contract).  pmc_configure can pin either form on or off if bit-identity
 *      across different shardings matters more than speed: with "split_components" 0 and "maha_gemm_tolerance" 0 every
 *      PER-SAMPLE output (log q, importance weights, responsibilities) is a function of the sample and the mixture alone,
 *      bit for bit whatever the batch or shard size (the sums over samples still depend on how the samples are chunked).
 *   2. The grouped responsibilities do not materialise r_nk, and with that the reference's clamp r == 0 -> tiny
 *      (variational.pyx:751-753) is applied only to a pair whose exp underflows within its OWN group of 16; a pair that
 *      underflows only against the row maximum of another group contributes 0 instead of 2.2e-308 to N_k / x-bar_k / S_k
 *      (at most N * 2.2e-308 per sum -- far below one ulp of any sum that matters).  pmc_responsibilities -- the
 *      form that writes r -- applies the clamp exactly as the reference does.
 *
 * Components of a sample block in pieces (round 6; k_logpdf_split / k_resp_groups_split).  A workgroup of the per-sample
 * kernels walks ALL components of its 256 samples (mixture.pyx:138-151 is a loop over the components too), so a call costs
 * K component steps whatever N is until the launch fills the chip -- the reference's own batches of 1e3 ... 1e5 samples
 * (examples/pmc.py:61-65) live there -- and the last round of a launch that does fill it leaves compute units idle.  Below
 * "split_max_rounds" rounds of the chip the LAST blocks of the launch (all of them when it is less than one round) are
 * therefore walked by several workgroups, each taking a run of components with the streaming log-sum-exp and leaving
 * (maximum, sum) per sample; the one that draws the block's last ticket combines the pairs in piece order --
 * M = max_p m_p, S = sum_p s_p exp(m_p - M), log S + M, logsumexp2D's own form (_regularize.pyx:72-81) about the row
 * maximum -- and finishes the block.  Bit-reproducible from run to run (the pieces are a function of N, K, D, the
 * device's CU count and these options); log q of a block in pieces agrees with the one-workgroup walk to the rounding
 * of the merge (a few ulps); the grouped responsibilities of pmc_estep keep their bits whatever the pieces.
 *   "split_components"          (default 1; 0: never)
 *   "split_min_components"      (default 0 = 4 per piece): smallest piece of a small launch
 *   "split_fill"                (default 1): a launch of less than one round is cut until it has this many workgroups per slot
 *   "split_max_pieces"          (default 16): ... but into at most this many pieces per mixture
 *   "split_max_rounds"          (default 24): launches of more rounds than this are not cut at all
 *   "split_tail_rounds"         (default 0.25): rounds in front of the last, partial one that are walked in pieces too
 *   "split_tail_pieces"         (default 4), "split_tail_min_components" (default 0 = 8 per piece, 4 from D = 32 on)
 *   "estep_small_batch_pieces"  (default 1): pmc_estep of a batch that does not fill the chip (VB, Gaussian Rao-Blackwell PMC,
 *                                K > 16) forms its responsibilities in groups of 16 components, the groups in pieces, and the
 *                                workgroup that finishes a block multiplies the groups' factors into u itself -- the per-component
 *                                statistics kernel behind takes a complete u.  The grouped form's notes (below) apply.
 * pmc_option_get / pmc_option_default read an option's current / built-in value.
 */
int pmc_configure(const char *key, double value);
int pmc_option_get(const char *key, double *value);
int pmc_option_default(const char *key, double *value);
int pmc_estep_is_fused(int K, int D, int kind, int mode);
/*
 * pmc_estep_about is pmc_estep with the moments taken about other points than the components' own means:
 * d_shift_pack (NULL = d_pack) is a pack -- e.g. from pmc_pack_means -- whose means are the shifts of d_stats.
 * GaussianInference takes its moments about the previous iteration's x-bar_k this way: x-bar_k is bit-stable as
 * soon as the responsibilities are (the reference's two passes have that property, variational.pyx:806-932), and
 * the far-shift second pass of an update is one call.  Responsibilities, d_scalars and d_vsums do not depend on it.
 */
int pmc_estep_about(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind, int mode,
                    int max_init_zero, const double *d_sample_w, const int64_t *d_latent, double *d_u,
                    double *d_scratch, double *d_vsums, double *d_stats, double *d_scalars, void *d_workspace,
                    const double *d_shift_pack, void *stream);
int pmc_estep(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind, int mode,
              int max_init_zero, const double *d_sample_w, const int64_t *d_latent, double *d_u,
              double *d_scratch, double *d_vsums, double *d_stats, double *d_scalars, void *d_workspace,
              void *stream);

/* ---- device-side packs and conversion (round 5) --------------------------------------------------------- */
/*
 * The K-sized steps around an E-step that ran on the host -- the Cholesky factorisations of pmc_pack_components
 * (variational.pyx:116-136 hands W_k over after every M-step), pmc_pack_means for a shifted statistics pass, and the
 * conversion of the statistics into the reference's conventions (pmc_host_convert_stats in pmc_ctx.h;
 * variational.pyx:699-932, pmc.pyx:188-222) -- as small kernels on the caller's stream: at one GPU's share of an 8-way
 * sharded E-step they were 15 % of the call.  Same operations in the same order as the host functions: identical bits.
 *
 * pmc_pack_components_device: the arguments of pmc_pack_components as DEVICE arrays (d_c0 ... d_column may be NULL with
 *   the same defaults); compiled dimensions (D <= pmc_max_compiled_dim()) only.  d_status: 2 K doubles on the device
 *   ([1 + failing pivot or 0, K | its value, K], every slot written); after the stream is synchronised copy them to the host and ask
 *   pmc_pack_status(K, h_status): PMC_OK, or PMC_ENOTPOSDEF naming the lowest component whose matrix does not factorise
 *   (its pack is then unusable; kernels that read it return NaN, they do not hang).  d_shift / d_shift_pack (both or
 *   neither): the K x D shifts of a statistics pass and the pack that receives them (pmc_pack_means) in the same launch.
 * pmc_pack_means_device: pmc_pack_means alone.
 * pmc_convert_stats_device: d_stats = the K x pmc_stats_stride(D) statistics (pmc_sufficient_stats' layout), d_shift
 *   K x D, d_n_cov K or NULL, d_scalars the call's 8 scalar sums or NULL;  d_out (pmc_convert_stats_len(K, D) doubles) =
 *   [S0 K | M1 K D | mean K D | cov K D D | far K | scalars 8]: far_k = 1.0 if component k's mean lies more than 10 of its
 *   standard deviations from its shift -- everything a caller reads after an E-step, in one block for one copy.
 */
int pmc_pack_components_device(int K, int D, const double *d_mu, const double *d_prec, const double *d_c0, const double *d_c1,
                               const double *d_c2, const double *d_c3, const double *d_weight, const int32_t *d_column,
                               double *d_pack, double *d_status, const double *d_shift, double *d_shift_pack, void *stream);
int pmc_pack_status(int K, const double *h_status);
int pmc_pack_means_device(int K, int D, const double *d_mu, double *d_pack, void *stream);
int64_t pmc_convert_stats_len(int K, int D);
int pmc_convert_stats_device(int K, int D, const double *d_stats, const double *d_shift, const double *d_n_cov,
                             const double *d_scalars, double *d_out, void *stream);

/* ---- the K-sized half of a variational-Bayes iteration on the device (round 6) --------------------- */
/*
 * What pypmc's GaussianInference does between two E-steps on K-sized arrays -- M_step (pypmc/mix_adapt/variational.pyx:
 * 129-136 with _update_m :693-697 and _update_W :934-946), the expectations an E-step starts with (:759-772, :800-804)
 * and likelihood_bound (:194-209, :948-1034, Wishart_log_B :1220-1247, Dirichlet_log_C :1269-1275) -- as kernels on
 * device-resident hyper-parameters, so that nothing K x D x D crosses the bus per iteration (pmc_ctx.h: pmc_vb_state is
 * the object that owns the arrays and strings the calls together).  D <= pmc_vb_max_dim() (64): one wavefront per
 * component, its matrix in LDS.
 *
 * pmc_vb_fields: DEVICE arrays, row-major -- the prior (alpha0, beta0, nu0: K; m0: K x D; inv_W0: K x D x D; log_det_W0:
 *   K), the posterior (alpha, beta, nu, m, W, log_det_W likewise), the expectations (ln_lambda = E[ln|Lambda_k|] (10.65),
 *   ln_pi = E[ln pi_k] (10.66): K) and the latest E-step's sums (N_comp K -- a zero counts as numpy's tiny, :699-709 --,
 *   x_mean K x D, S K x D x D).
 * pmc_vb_mstep_device: alpha, beta, nu, m, W, log_det_W from the prior and the sums.  W_k = inv(W_k^-1) through the
 *   Cholesky factor (the algorithm of LAPACK's potrf / potri, which the host path calls: the two agree to rounding,
 *   not bitwise; W comes out symmetric bit for bit).  d_status: 2 K doubles ([1 + failing pivot or 0 | its value], every
 *   slot written); pmc_vb_mstep_status(K, h_status) turns a copy of them into PMC_OK / PMC_ENOTPOSDEF naming the lowest
 *   component (its W is NaN then).
 * pmc_vb_expectations_device: ln_lambda, ln_pi from alpha, nu, log_det_W; and, if not NULL, the two constants of the
 *   posterior's pack (enum pmc_kind, PMC_KIND_VB) that are not fields: d_c0 = D / beta, d_c3 = ln_lambda - D ln 2 pi.
 *   d_psi_parts (2 K doubles or NULL): the caller's own [E[ln pi_k] K | sum_i psi((nu_k + 1 - i) / 2) + D ln 2, K]; the
 *   kernel then only adds ln|W_k|.  For callers that need the REFERENCE's psi bit for bit: with the default prior
 *   nu0 = D - 1 + 1e-5 the sum holds psi(5e-6) = -2e5, and one ulp of that is 3e-11 of every exponent of the E-step.
 * pmc_vb_after_device: behind an E-step -- d_conv = pmc_convert_stats_device's block; copies its means and covariances
 *   into x_mean / S (and the means into d_shift_prev, K x D: the next E-step's shifts), N_comp = its S0 with zeros
 *   replaced, *d_log_q_Z = the E-step's first scalar (10.75), and writes what a host reads after an E-step,
 *   d_small (pmc_vb_small_len(K) = 4 K + 8 doubles) = [N_comp K | far K | 1 if x_mean_k is finite, K | 1 if S_k has a
 *   finite entry, K | the call's 8 scalars].
 * pmc_vb_newshift_device: the shifts of a second statistics pass (a mean far from its shift: variational.pyx:806-932's two
 *   passes): d_out = d_shift + M1 / S0 where S0 > 1e-200, d_shift elsewhere.
 * pmc_vb_bound_device: d_out[8] = [L(Q) | E log p(X) | E log p(Z) | E log p(pi) | E log p(mu, Lambda) | E log q(Z) |
 *   E log q(pi) | E log q(mu, Lambda)]; d_scratch: pmc_vb_bound_scratch_len(K) doubles whose LAST EIGHT are zero before the
 *   first call (a ticket counter: the workgroup that finishes last adds the terms up and leaves it zero).  Sums over
 *   components run in component order: the same input gives the same bits.
 * pmc_host_digamma / pmc_host_lgamma: the psi and ln Gamma the kernels use (x > 0; NaN otherwise), on the host, for
 *   tests: |error| <= 2e-15 (1 + |value|) for psi, 1e-14 (1 + |value|) for ln Gamma (the recurrence below x = 10 costs the
 *   difference of two logarithms of about 17).
 */
/*
 * pmc_spd_inverse_device: chol_inv_det (pypmc/tools/_linalg.pyx:41-95) of K symmetric positive definite D x D matrices on
 *   the device, D <= 64 -- the K factorisations of a PMC update (Gauss.update, gauss.pyx:46-57, called per component from
 *   pmc.pyx:227-244) without LAPACK.  d_A: K x D x D row-major; d_out: pmc_spd_inverse_len(K, D) = K (2 D^2 + 3) doubles,
 *   per matrix [lower Cholesky factor D x D | inverse D x D | ln det | 1 + failing pivot or 0 | its value] (a matrix that
 *   does not factorise: NaN and its pivot).  The algorithm is potrf / potri's; the numbers agree with LAPACK's to rounding.
 */
int64_t pmc_spd_inverse_len(int K, int D);
int pmc_spd_inverse_device(int K, int D, const double *d_A, double *d_out, void *stream);

typedef struct pmc_vb_fields {
    double *alpha0, *beta0, *nu0, *m0, *inv_W0, *log_det_W0;
    double *alpha, *beta, *nu, *m, *W, *log_det_W;
    double *ln_lambda, *ln_pi;
    double *N_comp, *x_mean, *S;
} pmc_vb_fields;
int pmc_vb_max_dim(void);
int pmc_vb_mstep_device(int K, int D, const pmc_vb_fields *f, double *d_status, void *stream);
int pmc_vb_mstep_status(int K, const double *h_status);
int pmc_vb_expectations_device(int K, int D, const pmc_vb_fields *f, const double *d_psi_parts, double *d_c0, double *d_c3,
                               void *stream);
/* pmc_vb_pack_device: the posterior's pack (pmc_pack_components_device's kernel and bits) with the E-step's expectations formed
 *   in the same launch from the caller's psi parts (pmc_vb_expectations_device with d_psi_parts, without its launch).
 * pmc_vb_convert_after_device: pmc_convert_stats_device + pmc_vb_after_device in one launch (K-sized kernels cost 4-5 us each
 *   whatever they do): d_stats / d_shift / d_scalars as there, d_conv receives the conversion's block. */
int pmc_vb_pack_device(int K, int D, const pmc_vb_fields *f, const double *d_psi_parts, double *d_pack, double *d_status,
                       const double *d_shift, double *d_shift_pack, void *stream);
int pmc_vb_convert_after_device(int K, int D, const double *d_stats, const double *d_shift, const double *d_scalars, double *d_conv,
                                const pmc_vb_fields *f, double *d_small, double *d_shift_prev, double *d_log_q_Z, void *stream);
int64_t pmc_vb_small_len(int K);
int pmc_vb_after_device(int K, int D, const double *d_conv, const pmc_vb_fields *f, double *d_small, double *d_shift_prev,
                        double *d_log_q_Z, void *stream);
int pmc_vb_newshift_device(int K, int D, const double *d_conv, const double *d_shift, double *d_out, void *stream);
int64_t pmc_vb_bound_scratch_len(int K);
int pmc_vb_bound_device(int K, int D, const pmc_vb_fields *f, const double *d_log_q_Z, double *d_scratch, double *d_out,
                        void *stream);
double pmc_host_digamma(double x);
double pmc_host_lgamma(double x);

/* ---- a PMC iteration without evaluating the proposal twice --------------------------------------- */
/*
 * The reference evaluates the proposal's component densities on the same samples twice per PMC iteration:
 * for the importance weights (pypmc/sampler/importance_sampling.py:197-215 through
 * pypmc/density/mixture.pyx:112-156) and again inside the update (pypmc/mix_adapt/pmc.pyx:23-43,
 * calculate_rho_rb; student_t_pmc also needs the Mahalanobis forms themselves, pmc.pyx:602-610).
 * The *_keep variants of the two weighting calls are the calls above with one more output: d_maha_tiles
 * (pmc_maha_tiles_size(N, K) doubles, or NULL) receives the Mahalanobis forms maha_nk of the FIRST mixture
 * (the proposal) in the library's tile-major layout, column = position in d_pack.
 *
 * pmc_estep_from_tiles is then the Rao-Blackwellised PMC E-step (pmc_estep with kind GAUSS or STUDENT_T, mode
 * PMC_RB) of the SAME samples in the SAME order, with a_nk, rho [gamma and the dof sums] formed from the kept
 * values instead of new quadratic forms: d_pack describes the components to update (any subset of the
 * proposal's, its `column` entries naming their positions in the kept tiles, K_tiles = the proposal's
 * component count) and must hold the parameters the tiles were made with.  The responsibilities equal those of
 * pmc_responsibilities bit for bit, and so do the statistics wherever pmc_estep runs k_resp + the per-component
 * statistics kernel; where pmc_estep takes its large-batch forms (grouped responsibilities, common-shift statistics,
 * the matrix-product Mahalanobis forms) the two agree to rounding (1e-11 of each component's largest sum).  d_u (K*ceil(N/64)*64 doubles) is required, d_vsums (2 K) for Student-t;
 * d_stats, d_scalars, d_workspace as for pmc_estep.
 *
 * pmc_importance_weights_emit goes one step further for a proposal whose update is known to follow (every
 * component alive, the update's sample weights = these importance weights): the weighting pass itself leaves
 * u_nk = w_n rho_nk (pmc.pyx:23-43, :188) [Student-t: w_n rho_nk gamma_nk, pmc.pyx:602-610, and in d_vsums (2 K) the
 * two sums per component pmc_responsibilities documents] in d_u (pmc_tile_buffer_len(N, K) doubles, the layout
 * pmc_responsibilities writes) -- the forms are parked there during the pass and replaced behind it, each read once,
 * with the row maximum and log-sum-exp the pass has anyway -- and pmc_estep_from_u reduces them to d_stats
 * (pmc_sufficient_stats' layout; the fast common-shift form where it applies, see pmc_estep).  No separate
 * responsibility kernel runs at all: BASELINE configuration 5's iteration is propose -> weights -> statistics.
 * Agrees with pmc_estep_from_tiles to rounding (the log-sum-exp is the streaming one of the weighting pass).
 */
int64_t pmc_maha_tiles_size(int64_t N, int K);
int pmc_importance_weights_emit(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                                const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                double *d_log_target_out, double *d_weights, double *d_scalars, void *d_workspace,
                                double *d_u, double *d_vsums, void *stream);
int pmc_estep_from_u(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind, const double *d_u,
                     double *d_stats, void *d_workspace, void *stream);
/*
 * The grouped form of the same pair (ABI 2): the responsibilities are the product  u_nk = d_u[n, k] * d_gscale[n, k / 16]
 * of a value per pair and a factor per (sample, group of 16 components) -- d_gscale: pmc_gscale_len(N, K) doubles,
 * tile-major like d_u ((tile * ceil(K / 16) + group) * 64 + lane).  It lets the weighting pass write every d_u value
 * ONCE, relative to its group's maximum, before the row's log-sum-exp is known; the statistics kernel (common-shift
 * form) multiplies its weight operand with the factors.  From compiled D = 32 on (Gaussian proposal, N >= 256,
 * K a multiple of 32 up to ~20 % padding) the pass is then the matrix-product form of the Mahalanobis forms, see
 * "maha_gemm_tolerance" below; everywhere else d_u is complete and the factors are ones.  pmc_estep_from_u_grouped may
 * complete d_u IN PLACE (the factors become ones) when the statistics kernel that applies factors is not the one the
 * shape gets or its a-posteriori test refuses the common shift: the pair (d_u, d_gscale) means the same u afterwards.
 *
 * The matrix-product form of the Mahalanobis forms (pypmc/tools/_linalg.pyx:10-39 called N K times), csrc/pmc_mgemm.hip:
 *   maha_nk = sum_m theta_km z_nm,  z_n = the monomials of x_n - c up to degree 2 about ONE centre c (midrange of the
 *   component means), theta_k from P_k = R_k^T R_k and mu_k - c -- a dense product on v_mfma_f64_16x16x4_f64 with the
 *   per-sample epilogue fused behind it.  Its rounding error grows with |P| |x - c|^2 instead of maha, so every sample is
 *   priced first: eps_g (Theta_1 |x - c|^2 + Theta_2 |x - c| + Theta_3) with the maxima over the components of
 *   s_k |P_k|_F, 2 s_k |P_k (mu_k - c)|, s_k (mu_k - c)^T P_k (mu_k - c)  (s_k = |d a / d maha|), eps_g = 3.5e-17 sqrt(number of
 *   monomials) ~ 1e-15 -- a probabilistic constant, see csrc/pmc_api.hip --; a workgroup (256 samples) that holds a sample
 *   beyond the tolerance, or a non-finite coordinate, is done by the exact kernel, launched behind in the same call.
 *   Student-t (round 5): the slope s = (nu + D) / (2 (nu + maha)) depends on the pair -- (nu + D) / (2 nu) at maha = 0, a
 *   fraction of it for all but the rare pair with maha << nu --, so the norms price maha itself (s_k = 1), and the pair's
 *   bound, price x its own slope, is tested in the epilogue where maha is known; a workgroup with a pair beyond the
 *   tolerance raises its flag there and is redone by the exact kernel like the others.  (Until round 5 the worst slope
 *   priced every pair and Student-t mixtures of small nu hardly ever took the form.)
 *   Compiled sample dimensions 32, 40, 48 and (round 5) 64, i.e. D = 31 ... 64.  pmc_mixture_logpdf[_keep] (with or without
 *   d_individual) / pmc_importance_weights[_keep, _emit_grouped] / pmc_estep take the form when they are given a workspace, no
 *   weight is negative or non-finite -- and, for the emitting passes and pmc_estep, none is zero: the passes that emit no u
 *   take components WITHOUT weight (pruned components of a PMC run) along: they stay out of the sum, and a workgroup in which
 *   such a component's value lies more than 700 above every weighted live one -- the only case in which the reference's
 *   maximum over ALL unweighted values (_regularize.pyx:73-77) changes the result: its terms underflow -- is done by the exact
 *   kernel behind --, K >= 24 pads to a multiple of 32 / 64 within 20 %, and
 *   pmc_configure("maha_gemm_tolerance", t) (default 5e-11, in units of a_nk; 0 = never) / ("maha_gemm_min_n", default
 *   49152 -- below, the exact kernels with the components of a block in pieces are the faster ones, round 6; 256 in round 5)
 *   allow it.  Compiled dimensions 20 and 24 (D = 17 ... 24, round 5) have it for the passes that emit no u only --
 *   pmc_mixture_logpdf, pmc_importance_weights -- and only with four full component tiles per pass (K pads to a multiple of
 *   64 within 20 %, K >= 96 at D <= 20, K >= 48 at D = 21 ... 24): below that, and for the emitting passes and pmc_estep, the
 *   vector kernels are the faster ones.  The log q of an emitting and of a non-emitting pass over the same samples
 *   therefore agree within the tolerance there, not bit for bit (at D >= 31 both take the same form).
 */
int64_t pmc_gscale_len(int64_t N, int K);
/* component tiles (of 16) per pass the matrix-product form would run this shape with; 0: the exact kernels */
int pmc_maha_gemm_tiles(int64_t N, int K, int D);
/* diagnostics of the last call that took the form with this workspace and shape (synchronises `stream`): the guard's
 * norms Theta_1..3 (h_norms[3]), the number of workgroups of 256 samples it refused, and the number of workgroups */
int pmc_maha_gemm_report(const void *d_workspace, int64_t N, int K, int D, void *stream, double *h_norms,
                         int64_t *h_refused, int64_t *h_workgroups);
int pmc_importance_weights_emit_grouped(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                                        const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                        double *d_log_target_out, double *d_weights, double *d_scalars, void *d_workspace,
                                        double *d_u, double *d_gscale, double *d_vsums, void *stream);
/*
 * The emitting pass of a mixture that holds PRUNED components (round 6).  gaussian_pmc / student_t_pmc prune by setting a
 * weight to 0 and leave the component in the mixture (pmc.pyx:109-117); the next weighting pass still evaluates it -- it
 * takes part in logsumexp2D's row maximum with its unweighted value (_regularize.pyx:73-77) --, the update that follows
 * forms responsibilities for the live components only (pmc.pyx:98-103, calculate_rho_rb over live_components).  The caller
 * sorts such a pack LIVE COMPONENTS FIRST: the first K_live components of d_pack have a weight > 0, the K - K_live behind them
 * have weight 0 (a precondition; a live component behind K_live would silently get no responsibilities).  d_u / d_gscale /
 * d_vsums then hold K_live columns -- pmc_tile_buffer_len(N, K_live), pmc_gscale_len(N, K_live), 2 K_live -- in the pack's
 * order, ready for pmc_estep_from_u_grouped with the pack of the K_live live components.  log q and the weights are
 * pmc_importance_weights' (all K components).  The responsibilities' denominator is this pass's log q, i.e. a log-sum-exp
 * about the maximum of ALL components' values, where calculate_rho_rb takes it about max(0, live values) (its dead columns are
 * zeros, pmc.pyx:26-34): the same number unless every live value lies below -700, where the reference's own sum leaves the
 * normal range.  K_live == K is pmc_importance_weights_emit_grouped.
 */
int pmc_importance_weights_emit_live(const double *d_x, int64_t N, int D, const double *d_pack, int K, int K_live, int kind,
                                     const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                     double *d_log_target_out, double *d_weights, double *d_scalars, void *d_workspace,
                                     double *d_u, double *d_gscale, double *d_vsums, void *stream);
int pmc_estep_from_u_grouped(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind, double *d_u,
                             double *d_gscale, double *d_stats, void *d_workspace, void *stream);
int pmc_mixture_logpdf_keep(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                            int max_init_zero, double *d_out, double *d_individual, int64_t ld,
                            const double *d_log_target, double *d_weights, const double *d_sample_w,
                            double *d_scalars, void *d_workspace, double *d_maha_tiles, void *stream);
int pmc_importance_weights_keep(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                                const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                double *d_log_target_out, double *d_weights, const double *d_sample_w,
                                double *d_scalars, void *d_workspace, double *d_maha_tiles, void *stream);
int pmc_estep_from_tiles(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                         int max_init_zero, const double *d_sample_w, const double *d_maha_tiles, int K_tiles,
                         double *d_u, double *d_vsums, double *d_stats, double *d_scalars, void *d_workspace,
                         void *stream);

/* ---- the cross-GPU exchange inside the library (optional) ---------------------------------------- */
/*
 * One process per GPU, samples sharded by rank; the only exchange of the path is a sum over ranks of the K-sized
 * statistics buffer of an update ([scalars | stats | vsums | caller's bookkeeping], one contiguous device array).
 * The reference gathers whole sample histories with mpi4py instead (pypmc/tools/parallel_sampler.py:58-71) and
 * broadcasts the adapted proposal back (examples/pmc_mpi.py:119-131).
 *
 * These entry points run that all-reduce on RCCL (ncclAllReduce, ncclDouble, ncclSum) on the caller's stream,
 * for callers that have no torch.distributed: librccl is opened at run time on first use (the copy already in the
 * process if there is one); a process that never calls them does not need it.  Bootstrap as with NCCL itself:
 * rank 0 calls pmc_comm_unique_id() and hands the PMC_COMM_ID_BYTES bytes to every rank by the caller's own
 * means (MPI, a file, a socket); every rank then calls pmc_comm_init() -- collectively -- with its rank, the
 * rank count and the HIP device it drives.  pmc_comm_allreduce_sum() sums d_buf[0..n) over the ranks in place,
 * stream-ordered, identical result on every rank (so the replicated K-sized host update needs no broadcast).
 * A communicator of one rank is valid (and is how a one-GPU box exercises the path).
 */
#define PMC_COMM_ID_BYTES 128
typedef struct pmc_comm pmc_comm;
int pmc_comm_unique_id(void *h_id);
int pmc_comm_init(int rank, int world, const void *h_id, int device, pmc_comm **out);
int pmc_comm_rank(const pmc_comm *comm, int *rank, int *world);
int pmc_comm_allreduce_sum(pmc_comm *comm, double *d_buf, int64_t n, void *stream);
int pmc_comm_destroy(pmc_comm *comm);

/* ---- the exchange without a ring: one-shot all-gather + ordered local sum (ranks of one node) ---- */
/*
 * The statistics vector of an update is small (7 464 doubles at K = 32, D = 20; 110 336 at K = 128, D = 40): a ring
 * all-reduce is 2 (G - 1) dependent hops of pure latency.  Here every rank owns a mailbox in its device memory that its
 * peers map through HIP IPC; pmc_p2p_allreduce_sum writes the rank's vector into its slot of EVERY mailbox (peer stores
 * over xGMI) followed by a flag, waits for the G flags of its own mailbox and adds the G slots in RANK ORDER: one hop, and
 * the same bits on every rank and from run to run.  Bootstrap like pmc_comm_*: every rank calls pmc_p2p_create (its
 * mailbox holds vectors of up to max_doubles), hands its PMC_P2P_HANDLE_BYTES bytes (pmc_p2p_handle) to all ranks by the
 * caller's own means (MPI_Allgather, a file, torch.distributed), and calls pmc_p2p_connect with the world x
 * PMC_P2P_HANDLE_BYTES bytes of all ranks in rank order.  The ranks must call pmc_p2p_allreduce_sum in the same order
 * with the same n; launches are asynchronous on `stream`.
 *
 * Safety (round 5).  The mailbox is fine-grained device memory (coherent while kernels of several devices run; uncached
 * or coarse-grained memory only where the runtime refuses it for IPC, or by PMC_P2P_MEMORY); the handle carries the
 * owner's host and PCI bus id, and pmc_p2p_connect -- collective -- refuses peers on another host or without a peer path
 * (hipDeviceCanAccessPeer), closes whatever it mapped when it fails, and ends with a SELF-TEST round (a known pattern per
 * rank, the rank-ordered sum compared bit for bit; PMC_P2P_SELFTEST=0 skips it).  A failure is a negative status with the
 * reason in pmc_last_error(): the caller then uses pmc_comm_* (RCCL) -- on ALL ranks, so the ranks must agree on the
 * outcome by their own means (pypmc_amd.parallel: one min all-reduce of the ok flags).
 * A rank whose peers do not arrive within PMC_P2P_TIMEOUT_S seconds (environment, default 20; values that are not
 * positive numbers are ignored) fills d_buf with NaN -- never its own unreduced numbers -- and raises the exchange's error
 * word (host memory): pmc_p2p_status (synchronises the stream) and the next pmc_p2p_allreduce_sum report it, and the
 * exchange refuses all further rounds.  pmc_p2p_info: "memory=finegrained world=4 ... selftest=passed".
 * world <= 16; one node (HIP IPC); the processes need HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts with dmabuf IPC only.
 * Replaces nothing by default: pmc_comm_allreduce_sum (RCCL) stays the default exchange.
 */
#define PMC_P2P_HANDLE_BYTES 128
typedef struct pmc_p2p pmc_p2p;
int pmc_p2p_create(int rank, int world, int64_t max_doubles, int device, pmc_p2p **out);
int pmc_p2p_handle(const pmc_p2p *p, void *h_handle);
int pmc_p2p_connect(pmc_p2p *p, const void *h_handles);
int pmc_p2p_allreduce_sum(pmc_p2p *p, double *d_buf, int64_t n, void *stream);
int pmc_p2p_status(pmc_p2p *p, void *stream);
int pmc_p2p_info(const pmc_p2p *p, char *buf, size_t buflen);
int pmc_p2p_destroy(pmc_p2p *p);

/* ---- kernel timing ----------------------------------------------------------------------------- */
/*
 * Roofline numbers for callers without a profiler.  While timing is enabled every launch of a hot kernel
 * is bracketed by HIP events on the caller's stream (the stream it is launched on).  pmc_get_timings
 * waits for the recorded events, returns one entry per kernel that ran since the last call -- launches,
 * summed milliseconds, and the ALGORITHMIC work of those launches: flops per sample K (D^2 + 4 D + 40) for
 * the log-pdf / responsibility kernels, K (1 + 2 D + D (D + 1)) for the statistics kernel; bytes 8 (D + 1)
 * per sample for the log-pdf, 8 (D + K) for responsibilities and statistics, 8 D for the fused E-step --
 * and clears the record.  Names: "k_logpdf", "k_resp", "k_stats", "k_estep_fused", "k_propose",
 * "finishing reductions".  Writes at most max_entries entries, *n_entries is the number available.
 */
typedef struct pmc_timing {
    char name[48];
    int calls;
    double ms;
    double flops;
    double bytes;
} pmc_timing;
int pmc_timing_enable(int on);
int pmc_get_timings(pmc_timing *h_out, int max_entries, int *n_entries);

#ifdef __cplusplus
}
#endif
#endif /* PMC_HIP_H */
