/*
 * pmc_ctx.h -- the handle layer of libpmc_hip.so: HOST pointers in, HOST pointers out.
 *
 * include/pmc_hip.h is the kernel-level ABI: device pointers, caller-owned memory, caller's stream, un-normalised
 * shifted sums.  This header is the layer SURVEY.md section 8(b) lists on top of it -- a context that owns the
 * device, a stream, the scratch buffers and (optionally) the RCCL communicator; handles for a mixture and for a
 * sharded sample array; and entry points that take the reference's own arrays (numpy buffers of the .pyx files)
 * and hand back the quantities the reference's loops produce, in the reference's conventions.  A Cython / C caller
 * needs nothing else: no HIP calls, no torch, no knowledge of packs, tiles or workspaces.  Everything here is
 * host-side C++ over the entry points of pmc_hip.h (pypmc_amd/csrc/pmc_ctx.hip); the kernels are the same.
 *
 * Conventions: every function returns 0 or a negative pmc_status (pmc_hip.h) and sets pmc_last_error(); calls are
 * synchronous (the result arrays are filled on return).  Threads: any thread may call into a context and its handles;
 * the calls of ONE context are serialised inside the library (a mutex per context), different contexts run side by
 * side -- each on its own stream, with its own scratch, its own options (pmc_ctx_configure) and its own timing record
 * (pmc_ctx_get_timings).  Destroying a handle while another thread still uses it is the caller's error.
 * All arrays are C-contiguous fp64 (int64 / int32 where said).  With a communicator joined (pmc_ctx_join) every rank holds
 * its shard of the samples and the K-sized results are those of ALL ranks' samples -- one all-reduce (sum) of the
 * statistics buffer per call, identical on every rank, no gather and no broadcast (the reference:
 * pypmc/tools/parallel_sampler.py:58-71 gathers the samples on the master, examples/pmc_mpi.py:119-131 broadcasts
 * the proposal back).
 */
#ifndef PMC_CTX_H
#define PMC_CTX_H

#include "pmc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmc_ctx pmc_ctx;
typedef struct pmc_mix pmc_mix;
typedef struct pmc_samples pmc_samples;

/* ---- context ---------------------------------------------------------------------------------------- */
/*
 * a context over the GPUs of one node for ONE host process -- the way pypmc's callers run (a single
 * Python process, pypmc/examples/pmc.py:53-73, examples/variational.py:54-61; its only multi-process code gathers whole
 * sample histories over MPI, pypmc/tools/parallel_sampler.py:58-66).
 *   n_devices, device_ids   the devices, in the order their shares are summed; n_devices = 0: the list in the
 *                           environment variable PMC_HIP_DEVICES ("0,1,2,3"), or every visible device.  The same
 *                           ordinal may appear more than once: virtual shards on one GPU, each with its own stream and
 *                           scratch (how the sharded path is tested and profiled on a one-GPU box).
 * The context owns one stream, one scratch set and one host thread per device.  pmc_samples_upload / _generate split
 * the rows into contiguous blocks (sizes differ by at most one, device order = row order; the library owns them);
 * pmc_mixture_create / _update copy the pack to every device; every N-sized call runs on all devices at once and the
 * K-sized vectors are added IN DEVICE ORDER -- each device's vector is copied into its slot on the first device (peer
 * copy over xGMI) and one kernel forms ((v_0 + v_1) + v_2) + ... -- so the result is bit-reproducible, and equal to the
 * ordered sum of the per-shard results of one-device contexts.  No IPC, no RCCL, no second process.  pmc_ctx_join on
 * top of it sums over the ranks of a multi-node run as before (the first device holds the communicator).
 * pmc_init(device, &ctx) is pmc_init_devices(1, &device, &ctx).
 * PMC_HIP_LOG=1: one line per N-sized call on stderr (samples, devices, milliseconds, samples per second).
 */
int pmc_init_devices(int n_devices, const int *device_ids, pmc_ctx **out);
int pmc_init(int device, pmc_ctx **out);
/* number of parts of the context; their device ordinals (returns the count, writes at most max_ids) */
int pmc_ctx_device_count(const pmc_ctx *ctx);
int pmc_ctx_devices(const pmc_ctx *ctx, int *h_device_ids, int max_ids);
/* Optional: join the ranks of a sharded run (collective; rank 0 draws the id with pmc_comm_unique_id and hands the
   PMC_COMM_ID_BYTES bytes to the others by its own means -- MPI_Bcast in a pypmc process). */
int pmc_ctx_join(pmc_ctx *ctx, int rank, int world, const void *h_id);
/*
 * The same sharded run with the one-shot exchange of pmc_hip.h (pmc_p2p_*: the ranks of ONE node; every rank writes its
 * statistics vector into every peer's mailbox through HIP IPC and adds them in rank order) instead of the RCCL
 * communicator: pmc_ctx_p2p_open creates this rank's mailbox (vectors of up to max_doubles: 8 + K (1 + D + D (D + 1) / 2)
 * + 2 K for the largest K, D the context will see) and returns its PMC_P2P_HANDLE_BYTES bytes; the caller gathers all
 * ranks' bytes in rank order (MPI_Allgather) and hands them to pmc_ctx_p2p_connect.  Collective; instead of pmc_ctx_join.
 */
int pmc_ctx_p2p_open(pmc_ctx *ctx, int rank, int world, int64_t max_doubles, void *h_handle);
/* (a negative status -- no peer path, another host, a failed self-test, pmc_hip.h -- leaves the context WITHOUT an exchange:
   join an RCCL communicator instead, on all ranks.  The failed exchange's mailbox stays allocated until pmc_shutdown: a
   peer that mapped it may still be inside its own self-test round, and this layer has no barrier across ranks) */
int pmc_ctx_p2p_connect(pmc_ctx *ctx, const void *h_handles);
/* Frees the stream, the scratch and the communicator.  Handles made from the context must be freed first. */
int pmc_shutdown(pmc_ctx *ctx);
/*
 * Library options for THIS context's calls (keys and values as pmc_configure, pmc_hip.h: "stats_common_shift_limit",
 * "maha_gemm_tolerance", ...).  A context starts from a copy of the process-wide options as they stand at pmc_init;
 * pmc_configure afterwards does not reach it, and this call changes nothing outside the context.
 */
int pmc_ctx_configure(pmc_ctx *ctx, const char *key, double value);
/*
 * Kernel timing of this context's launches only (HIP events on the context's stream around every hot kernel, see
 * pmc_get_timings in pmc_hip.h for the entries' meaning): enable, run calls, read.  Independent of the process-wide
 * pmc_timing_enable / pmc_get_timings, which never see a context's records while its own timing is on.
 * A context of several devices merges its parts' records by kernel name: calls, flops and bytes added, ms = the slowest
 * part's (the parts run side by side).
 */
int pmc_ctx_timing_enable(pmc_ctx *ctx, int on);
int pmc_ctx_get_timings(pmc_ctx *ctx, pmc_timing *h_out, int max_entries, int *n_entries);

/* ---- mixture ------------------------------------------------------------------------------------------ */
/*
 * MixtureDensity of K Gauss (family PMC_KIND_GAUSS) or StudentT (PMC_KIND_STUDENT_T) components
 * (pypmc/density/mixture.pyx:21-59) from the arrays the components hold:
 *   h_w          K   weights (mixture.pyx:56; zeros = dead components, pmc.pyx:66)
 *   h_mu         K x D
 *   h_inv_sigma  K x D x D   (gauss.pyx:112, student_t.pyx:111)
 *   h_log_norm   K   log_normalization (gauss.pyx:115, student_t.pyx:32-34)
 *   h_dof        K   degrees of freedom (StudentT only, else NULL)
 * The arrays are copied; the host stays authoritative: after component.update / a weight change call
 * pmc_mixture_update with the new arrays.  PMC_ENOTPOSDEF names the component whose inv_sigma does not factorise.
 */
int pmc_mixture_create(pmc_ctx *ctx, int family, int K, int D, const double *h_w, const double *h_mu,
                       const double *h_inv_sigma, const double *h_log_norm, const double *h_dof, pmc_mix **out);
int pmc_mixture_update(pmc_mix *mix, const double *h_w, const double *h_mu, const double *h_inv_sigma,
                       const double *h_log_norm, const double *h_dof);
int pmc_mixture_destroy(pmc_mix *mix);

/* ---- samples ------------------------------------------------------------------------------------------ */
/* This rank's N x D block of the sample array, resident on the device(s) until freed (History's samples[-1]); a context
   of several devices holds contiguous blocks of it, device order = row order (pmc_samples_shard). */
int pmc_samples_upload(pmc_ctx *ctx, const double *h_x, int64_t N, int D, pmc_samples **out);
/*
 * MixtureDensity.propose(N, trace=True, shuffle=False) on the device (mixture.pyx:159-212): h_counts (K) are the
 * component counts of the caller's generator (rng.multinomial, mixture.pyx:192 -- counts and origins stay
 * bit-exact), the normal / chi-square numbers are Philox4x32-10 keyed by `seed` and counted from the GLOBAL sample
 * index `first_sample` (shards of one logical run draw disjoint streams).  h_chol: K x D x D lower Cholesky factors of
 * the covariances (component.cholesky_sigma), or NULL = derived from inv_sigma.  The origin (generating component per
 * sample, sorted) stays with the handle: pmc_samples_origin copies it out, pmc_pmc_update_stats can use it as latent.
 */
int pmc_samples_generate(pmc_ctx *ctx, const pmc_mix *mix, const double *h_chol, const int64_t *h_counts,
                         uint64_t seed, int64_t first_sample, pmc_samples **out);
/*
 * A sample array that is ALREADY on the context's device (a torch tensor of the front-end, the output of another
 * library): N x D row-major fp64 at d_x, borrowed -- not copied, not freed by pmc_samples_free, and the caller keeps it
 * alive and unchanged while the handle is in use.  One-device contexts only.  Work that produced the array on another
 * stream must have completed (the context launches on its own stream).
 */
int pmc_samples_wrap(pmc_ctx *ctx, const double *d_x, int64_t N, int D, pmc_samples **out);
/*
 * Sample weights of the VB E-step that stay with the handle (GaussianInference normalises its weights once, in the
 * constructor, variational.pyx:86-100, and uses them in every E-step): copied to the shards from the host (h_w: N
 * doubles; NULL removes them), or a borrowed device array next to a wrapped sample array.  pmc_vb_estep uses them when it
 * is called with h_sample_w == NULL; an h_sample_w given there is uploaded for that call and wins.
 */
int pmc_samples_set_sample_weights(pmc_samples *s, const double *h_w);
int pmc_samples_wrap_sample_weights(pmc_samples *s, const double *d_w);
int64_t pmc_samples_count(const pmc_samples *s);
/* rows [*begin, *begin + *count) live on part `part` of the context; returns that part's device ordinal */
int pmc_samples_shard(const pmc_samples *s, int part, int64_t *begin, int64_t *count);
int pmc_samples_download(const pmc_samples *s, double *h_x);
int pmc_samples_origin(const pmc_samples *s, int64_t *h_origin);
int pmc_samples_free(pmc_samples *s);

/* ---- evaluation --------------------------------------------------------------------------------------- */
/*
 * MixtureDensity.multi_evaluate(x, out, individual) (mixture.pyx:112-156 with gauss.pyx:146-151 /
 * student_t.pyx:154-164 and logsumexp2D, _regularize.pyx:57-84):  h_out N (or NULL), h_individual N x K row-major
 * (or NULL).
 */
int pmc_mix_logpdf(const pmc_mix *mix, const pmc_samples *s, double *h_out, double *h_individual);
/*
 * The subset mode of the same method, multi_evaluate(x, individual=..., components=[...]) (mixture.pyx:153-156; its
 * caller: calculate_rho_rb on the live components, pmc.pyx:27): only the listed components are evaluated and only their
 * columns of h_individual (N x K row-major, K = the mixture's component count) are written; every other column keeps
 * what the caller had there, and there is no `out` (the reference returns None in this mode).
 */
int pmc_mix_logpdf_components(const pmc_mix *mix, const pmc_samples *s, const int32_t *h_components, int ncomponents,
                              double *h_individual);
/*
 * ImportanceSampler._calculate_weights (pypmc/sampler/importance_sampling.py:197-215): w_n = exp(log P(x_n) -
 * log q(x_n)).  log P either from the host (h_log_target, N: the user's target evaluated by the caller) or from a
 * second mixture (`target`; then both densities are evaluated in one pass and h_log_target_out, if not NULL,
 * receives log P -- the sampler's target_values).  h_w N (or NULL: the weights stay on the device only);
 * h_sums[3] = sum w, sum w log w (zeros masked), sum w^2 over ALL ranks' samples: perp and ess
 * (pypmc/tools/convergence.py:31-39, :67-72) follow from them.  The weights stay with the sample handle for
 * pmc_pmc_update_stats(weights_on_device = 1).
 */
int pmc_is_weights(const pmc_mix *q, pmc_samples *s, const double *h_log_target, const pmc_mix *target,
                   double *h_w, double *h_log_target_out, double *h_sums);

/* ---- VB E-step ---------------------------------------------------------------------------------------- */
/*
 * GaussianInference.E_step (pypmc/mix_adapt/variational.pyx:116-127) for K components from the variational
 * parameters the object holds --  h_m K x D, h_W K x D x D, h_nu, h_beta K, h_ln_pi = expectation_ln_pi K,
 * h_ln_lambda = expectation_det_ln_lambda K (:759-772, :800-804) -- and optional sample weights h_sample_w (N, as
 * the constructor normalised them, :86-100; NULL = the weights that stay with the handle,
 * pmc_samples_set_sample_weights, or unweighted).  Results in the reference's conventions, over ALL
 * ranks' samples:
 *   h_N_k   K          N_comp, zeros regularised to tiny (:699-709)
 *   h_xbar  K x D      x_mean_comp (:806-853)
 *   h_S     K x D x D  S, symmetric (:855-932)
 *   h_elogqz 1         sum_n w_n sum_k r_nk log rho~_nk (:1003-1013)
 *   h_r, h_log_rho     N x K row-major, this rank's rows (NULL: not materialised -- the E-step itself never needs them)
 * The moments are taken in one pass about m_k (about h_shift, K x D, if given: the previous E-step's x_mean_comp
 * makes x_mean_comp and S bit-stable as soon as r is) and repeated about the mean just found when that turns out
 * more than 10 of the component's own standard deviations away (the reference takes the mean first, then the
 * covariance about it).  Non-finite inputs come back as non-finite sums; the reference's checks of N_comp and S
 * (variational.pyx:122-126) stay with the caller.  PMC_ENOTPOSDEF names the component whose W_k does not factorise.
 * For compiled dimensions (D <= 64) nothing K-sized runs on the host: the parameters go up in one copy, the pack
 * (pmc_pack_components_device), the shift pack and the conversion (pmc_convert_stats_device) are kernels, one copy
 * brings N_comp / x_mean_comp / S / E[log q(Z)] back -- bit for bit what the host functions give.
 */
int pmc_vb_estep(pmc_ctx *ctx, const pmc_samples *s, const double *h_sample_w, int K, const double *h_m,
                 const double *h_W, const double *h_nu, const double *h_beta, const double *h_ln_pi,
                 const double *h_ln_lambda, const double *h_shift, double *h_N_k, double *h_xbar, double *h_S,
                 double *h_elogqz, double *h_r, double *h_log_rho);

/* ---- a VB fit whose K-sized state stays on the device (round 6) --------------------------------------- */
/*
 * GaussianInference.update() = M_step + E_step (pypmc/mix_adapt/variational.pyx:571-578, :129-136, :116-127) and
 * likelihood_bound() (:194-209) for an object whose hyper-parameters live on the context's device between the calls: with
 * pmc_vb_estep the K x D x D arrays W and S cross the bus twice per iteration and the M-step's K inversions, the digamma
 * sums and the bound run in the caller's interpreter -- at one GPU's share of eight that was an eighth of an iteration.
 * D <= pmc_vb_max_dim().  A context over several devices (pmc_init_devices) keeps the state on its first device; per E-step
 * the other devices get the posterior they build their pack from in one peer copy, the statistics come back as in pmc_vb_estep
 * (slots on the first device, added in device order); such a context needs the caller's psi parts (h_psi_parts != NULL).
 *
 * Fields (enum pmc_vb_field; K, K x D or K x D x D doubles, row-major): the prior ALPHA0, BETA0, NU0, M0, INV_W0 (the
 *   INVERSE of the constructor's W0), LOG_DET_W0 (of W0); the posterior ALPHA, BETA, NU, M, W, LOG_DET_W; the expectations
 *   LN_LAMBDA, LN_PI (written by every E-step); the latest E-step's N_COMP, X_MEAN, S; SHIFT_PREV (the means the next
 *   E-step may take its moments about); and, read-only, E_M ... E_LN_LAMBDA: the parameters the latest E-step ran with
 *   (the object's r / log_rho attributes belong to THEM even after a later M-step).
 * pmc_vb_state_put queues a copy from the host (pinned staging: the call returns before the copy ran); pmc_vb_state_get
 *   waits for everything queued and copies one field back.
 * pmc_vb_state_step(st, samples, flags, h_psi_parts, h_result): the steps named in `flags`, in this order --
 *   PMC_VB_DO_MSTEP   M-step from the prior and N_COMP / X_MEAN / S (pmc_vb_mstep_device);
 *   PMC_VB_DO_ESTEP   expectations -> pack -> responsibilities and statistics over `samples` (and the ranks of a joined
 *                     context) -> N_COMP, X_MEAN, S; with PMC_VB_ABOUT_PREV the moments are taken about SHIFT_PREV instead
 *                     of M (pmc_vb_estep's h_shift); a second pass about the mean just found happens inside when needed;
 *                     h_psi_parts (2 K doubles or NULL) = the caller's own psi parts of the expectations
 *                     (pmc_vb_expectations_device: [E[ln pi] | sum psi + D ln 2]; both depend on N_COMP alone, which the
 *                     caller holds) -- NULL: the device's psi;
 *   PMC_VB_DO_BOUND   the bound of the state as it is then;
 *   and ONE copy brings back h_result (pmc_vb_state_result_len(K) = 8 K + 16 doubles) = [N_comp K | far K | 1 if
 *   x_mean_k is finite, K | 1 if S_k has a finite entry, K | the E-step's 8 scalars (first: E log q(Z)) | pack status 2 K |
 *   M-step status 2 K | L(Q) and its seven terms (pmc_vb_bound_device)].  h_result may be NULL for an M-step alone: it is
 *   queued, and a W_k^-1 that does not factorise is reported (PMC_ENOTPOSDEF, naming the component) by the next call
 *   that copies a block back.  The reference's checks of N_comp and S (variational.pyx:122-126) stay with the caller.
 */
/* Lifetime: pmc_shutdown destroys the states of its context; a handle that outlives it is refused by every call
 * (PMC_EINVAL) and pmc_vb_state_destroy of it is a no-op -- a garbage collector may finalise in either order. */
typedef struct pmc_vb_state pmc_vb_state;
enum pmc_vb_field {
    PMC_VB_ALPHA0 = 0, PMC_VB_BETA0, PMC_VB_NU0, PMC_VB_M0, PMC_VB_INV_W0, PMC_VB_LOG_DET_W0,
    PMC_VB_ALPHA, PMC_VB_BETA, PMC_VB_NU, PMC_VB_M, PMC_VB_W, PMC_VB_LOG_DET_W,
    PMC_VB_LN_LAMBDA, PMC_VB_LN_PI, PMC_VB_N_COMP, PMC_VB_X_MEAN, PMC_VB_S, PMC_VB_SHIFT_PREV,
    PMC_VB_E_M, PMC_VB_E_W, PMC_VB_E_BETA, PMC_VB_E_NU, PMC_VB_E_LN_PI, PMC_VB_E_LN_LAMBDA,
    PMC_VB_NFIELDS
};
enum { PMC_VB_DO_MSTEP = 1, PMC_VB_DO_ESTEP = 2, PMC_VB_DO_BOUND = 4, PMC_VB_ABOUT_PREV = 8 };
int pmc_vb_state_create(pmc_ctx *ctx, int K, int D, pmc_vb_state **out);
int pmc_vb_state_destroy(pmc_vb_state *st);
int pmc_vb_state_put(pmc_vb_state *st, int field, const double *h);
int pmc_vb_state_get(pmc_vb_state *st, int field, double *h);
int64_t pmc_vb_state_result_len(int K);
int pmc_vb_state_step(pmc_vb_state *st, const pmc_samples *s, int flags, const double *h_psi_parts, double *h_result);
/*
 * GaussianInference.run's loop (variational.pyx:283-359) for as long as no component has to go: per iteration update() (M-step,
 * then E-step and bound as in pmc_vb_state_step) and the reference's convergence rules (:330-352), with nothing between two
 * iterations but `psi` -- the caller's psi parts (pmc_vb_state_step; called with the N_comp the queued M-step uses, while that
 * kernel runs; NULL: the device's psi).  Starts from `old_bound` (the bound of the state as it is) and h_N_comp (the latest
 * N_comp, K), moments about the previous means from the first E-step on if about_prev.  Returns when
 *   h_info[1] = PMC_VB_RUN_CONVERGED  the rules say so;
 *               PMC_VB_RUN_PRUNE      an update left a component with N_k < prune_threshold (the caller prunes and calls again);
 *               PMC_VB_RUN_LOOK       N_comp, S or the bound of the latest block is not finite (the reference's checks, :122-126,
 *                                     are the caller's);
 *               PMC_VB_RUN_CAP        max_iterations updates are done;
 * h_info[0] = updates done, h_info[2] = how often the bound decreased, h_info[3] = whether the next E-step may take its moments
 * about the latest means; h_bounds = [bound, the bound before the last update]; h_result = the last update's block.  A matrix
 * that does not factorise ends the loop with PMC_ENOTPOSDEF (h_info[0] = updates completed before it).
 */
typedef void (*pmc_vb_psi_fn)(void *user, int K, const double *h_N_comp, double *h_psi_parts);
enum { PMC_VB_RUN_CAP = 0, PMC_VB_RUN_CONVERGED = 1, PMC_VB_RUN_PRUNE = 2, PMC_VB_RUN_LOOK = 3 };
int pmc_vb_state_run(pmc_vb_state *st, const pmc_samples *s, int max_iterations, double old_bound, double prune_threshold,
                     double rel_tol, double abs_tol, int about_prev, const double *h_N_comp, pmc_vb_psi_fn psi, void *user,
                     double *h_result, int *h_info, double *h_bounds);

/* ---- PMC update --------------------------------------------------------------------------------------- */
/*
 * The N-sized part of gaussian_pmc / student_t_pmc (pypmc/mix_adapt/pmc.pyx:120-246, :499-739) for the proposal
 * `mix` and the samples it proposed: rho (Rao-Blackwellised, pmc.pyx:23-43, rb != 0) or the one-hot latent
 * responsibilities (pmc.pyx:45-51, rb == 0; latent = h_latent, or the origin pmc_samples_generate left with the
 * handle when h_latent is NULL), then the weighted sums of pmc.pyx:188-222 / :602-691, over ALL ranks' samples.
 * Weights: h_w (N), or the importance weights pmc_is_weights left on the device (weights_on_device != 0), or none.
 * Components with weight 0 are dead (pmc.pyx:66): they take part in the row maximum only and their rows of the
 * outputs are not written.  Outputs (the host does component.update, brentq, mincount pruning and the renormalisation):
 *   h_alpha     K          sum_n w rho / sum_n w                 (pmc.pyx:191-193, :612-617)
 *   h_mu        K x D      new means       (:194-197; Student-t: weighted with gamma, :620-623)
 *   h_sigma     K x D x D  new covariances (:198-204; Student-t: sum w rho gamma (x-mu)(x-mu)^T / sum w rho, :629-630)
 *   h_dof_const K          Student-t only (else NULL): the constant c_k of the dof condition
 *                          c_k + log(nu/2) - psi(nu/2) = 0 (:654-696; _DOFCondition :478-497)
 *   h_loglik    1          sum_n w_n log q(x_n) of the proposal (:388-391; Rao-Blackwellised form only: the latent form
 *                          does not evaluate the mixture), may be NULL
 *   h_norm      1          sum_n w_n (N if unweighted), may be NULL
 */
int pmc_pmc_update_stats(pmc_ctx *ctx, const pmc_mix *mix, const pmc_samples *s, const double *h_w,
                         int weights_on_device, const int64_t *h_latent, int rb, double *h_alpha, double *h_mu,
                         double *h_sigma, double *h_dof_const, double *h_loglik, double *h_norm);

/* ---- weighted moments ---------------------------------------------------------------------------------- */
/*
 * calculate_mean / calculate_covariance (pypmc/sampler/importance_sampling.py:46-83) of this context's samples -- ALL
 * ranks' with a communicator joined -- with the weights h_w (N), the importance weights pmc_is_weights left on the
 * device (weights_on_device != 0), or none (w = 1):
 *   h_mean  D        sum_n w_n x_n / sum_n w_n                                                        (:58-61)
 *   h_cov   D x D    (sum w)^2 / ((sum w)^2 - sum w^2) * sum_n w_n (x_n - mean)(x_n - mean)^T / sum w   (:76-83), may be NULL
 * One pass of the statistics kernel over the samples (moments about the first sample of rank 0's shard).
 */
int pmc_weighted_moments(pmc_ctx *ctx, const pmc_samples *s, const double *h_w, int weights_on_device, double *h_mean,
                         double *h_cov);

/* ---- host-side conversion (no device involved) ----------------------------------------------------------- */
/*
 * The K-sized step between the statistics buffer of the kernel level (pmc_sufficient_stats' layout: per component
 * sum u | sum u d (D) | sum u d d^T (lower triangle), d = x - shift_k) and the reference's conventions, for callers that
 * run the kernel level themselves (pypmc_amd's front-end does):
 *   h_S0    K          the raw sums (N_comp before its zeros are regularised, variational.pyx:699-709)
 *   h_M1    K x D      the raw shifted first moments (NULL: not wanted)
 *   h_mean  K x D      shift_k + M1_k / reg(S0_k)                                  (x_mean_comp, :806-853; pmc.pyx:194-197)
 *   h_cov   K x D x D  (M2_k - reg(S0_k) dbar dbar^T) / reg(n_cov_k), symmetric    (S, :855-932; pmc.pyx:198-204, :629-630)
 *                      h_n_cov (K) = the normalisation of the covariance when it is not S0 (Student-t PMC), else NULL
 *   h_far   1          != 0 if some component holding more than a millionth of the weight has its mean more than 10 of
 *                      its own standard deviations (in some coordinate) from its shift: the one-pass moments cancel, the
 *                      caller repeats the statistics about the mean just found (pypmc_amd.mix_adapt._stats.shift_is_far)
 * Same operations in the same order as the numpy code it replaces (bit-identical results).
 */
int pmc_host_convert_stats(int K, int D, const double *h_stats, const double *h_shift, const double *h_n_cov,
                           double *h_S0, double *h_M1, double *h_mean, double *h_cov, int *h_far);

/*
 * chol_inv_det (pypmc/tools/_linalg.pyx:41-95: potrf, potri, symmetrised inverse, log det) of a stack of K symmetric
 * D x D matrices -- the K-sized host step of every proposal / posterior update (pmc.pyx:227-244, variational.pyx:934-946)
 * -- as ONE call (the per-matrix interpreter overhead and array glue of the Python loop it replaces were most of its
 * time; PMC_HOST_THREADS > 1 spreads the matrices over threads for a LAPACK that takes concurrent calls, which scipy's
 * OpenBLAS does not).  dpotrf / dpotri: the addresses of the LAPACK routines to use
 * (Fortran convention `void f(char *uplo, int *n, double *a, int *lda, int *info)`; from Python:
 * scipy.linalg.cython_lapack.__pyx_capi__), so the results are bit for bit what the reference's scipy calls give.
 * h_lower K x D x D (L with m = L L^T, upper triangle zero), h_inverse K x D x D (symmetric), h_log_det K, h_failed K
 * (LAPACK's info per matrix, -1 for a non-finite determinant; may be NULL).  PMC_ENOTPOSDEF if any matrix fails (the
 * arrays of the others are still filled).
 */
int pmc_host_chol_inv_det_batch(int K, int D, const double *h_m, void *dpotrf, void *dpotri, double *h_lower,
                                double *h_inverse, double *h_log_det, int *h_failed);

#ifdef __cplusplus
}
#endif
#endif /* PMC_CTX_H */
