// pmc_persample.hip / pmc_stats.hip -- the gfx950 kernels of the adaptive-importance-sampling hot
// path, each compiled once per sample dimension:  hipcc -DPMC_D=<D> -DPMC_PADDED=<0|1> -c <file>
//
// Execution model (CDNA4, wave64):
//   * Per-sample kernels (k_logpdf, k_resp): one lane owns one sample; its D coordinates stay in
//     VGPRs for the whole component loop.  Everything that depends on the component only (mean,
//     whitening factor R_k, constants) is wave-uniform, so it is fetched with scalar loads
//     (address space 4 -> s_load_dwordx16 through the scalar cache) and used as the SGPR operand of
//     v_fma_f64: the triangular product y = R_k (x - mu_k) costs D(D+1)/2 v_fmac_f64 and no LDS or
//     vector-memory traffic at all.  fp64 MFMA has the same peak as fp64 VALU on gfx950 and cannot
//     exploit the triangular structure or D not a multiple of 16, so it is not used.
//   * Statistics kernel (k_stats): one wavefront owns one (component, row-subset) task and streams
//     over a chunk of samples with ~50 per-lane fp64 accumulators; the 64 x D sample tile is
//     loaded coalesced and transposed through LDS once per workgroup and shared by its wavefronts.
//   * All reductions are fixed-order trees (per lane -> wavefront shuffle -> per-block partial ->
//     one finishing kernel): bit-reproducible run to run, no fp64 atomics.
//
// Arithmetic follows the reference's operation order outside the Mahalanobis product; explicit
// fma() is used only where stated and the unit is compiled with -ffp-contract=off.
#include "pmc_device.h"

namespace {

// Minimum wavefronts per SIMD the register allocator has to leave room for.  Beyond D = 48 the
// kernels would otherwise take 256 VGPRs + AGPRs = one wavefront per SIMD; two with ~50 spilled
// registers are 1.7x faster (D = 64: 9.3 -> 5.3 ms per 2e6 samples x 16 components).
__host__ __device__ constexpr int pmc_min_waves(int D) { return D > 48 ? 2 : 1; }

// The wavefronts of a workgroup walk the components in step: a barrier per component keeps them on
// the same parameter lines, so that one wavefront's scalar-cache fill serves the other three
// (D = 20: -3 % kernel time; neutral at D = 40).
__device__ __forceinline__ void component_sync()
{
#ifndef PMC_NO_COMPONENT_SYNC
    __builtin_amdgcn_s_barrier();
#endif
}

// ---------------------------------------------------------------------------------------------
// k_logpdf: MixtureDensity.multi_evaluate (mixture.pyx:112-156) + logsumexp2D
// (_regularize.pyx:57-84) [+ importance weights, importance_sampling.py:197-215] in one pass.
// ---------------------------------------------------------------------------------------------
template <int D, bool PADDED, int KIND>
__global__ __launch_bounds__(PMC_A_WAVES * 64, pmc_min_waves(D)) void k_logpdf(const PmcArgsA a)
{
    constexpr int T = pmc_tri(D), STRIDE = pmc_pack_stride_c(D);
    const long long n = ((long long)blockIdx.x * PMC_A_WAVES * 64) + threadIdx.x;
    const bool valid = n < a.N;

    double xv[D];
    load_row<D, PADDED>(a.x, n, a.N, a.dreal, xv);

    // pass 0: the mixture itself; pass 1 (pmc_importance_weights only): the TARGET mixture of the
    // importance weights, evaluated on the same registers -- the samples are read once
    double lse = 0.0, lse_target = 0.0;
    const int npass = a.pack2 != nullptr ? 2 : 1;
    for (int which = 0; which < npass; ++which) {
        double m = (which == 0 && a.max_init_zero) ? 0.0 : -DBL_MAX, s = 0.0;
        cdouble *pk = (cdouble *)(which == 0 ? a.pack : a.pack2);
        const int K = which == 0 ? a.K : a.K2;
        for (int k = 0; k < K; ++k, pk += STRIDE) {
            component_sync();
            touch_component<D>(pk);
            const double maha = mahalanobis<D>(xv, pk);
            double expo;
            const double v = component_value<D, KIND>(maha, pk + D + T, expo);
            if (which == 0 && a.individual != nullptr) {
                const long long col = ((cint64 *)pk)[D + T + 5];
                if (valid) a.individual[n * a.ld + col] = v;
            }
            lse_step(v, pk[D + T + 4], m, s);
        }
        const double l = log(s) + m;                     // _regularize.pyx:81
        if (which == 0) lse = l;
        else lse_target = l;
    }
    if (a.out != nullptr && valid) a.out[n] = lse;
    if (a.log_target_out != nullptr && valid) a.log_target_out[n] = lse_target;

    if (a.partials == nullptr && a.log_target == nullptr && a.pack2 == nullptr) return;

    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if ((a.log_target != nullptr || a.pack2 != nullptr) && valid) {
        const double tmp = (a.pack2 != nullptr ? lse_target : a.log_target[n]) - lse;   // importance_sampling.py:204
        const double w = exp(tmp);                        // :207
        a.weights[n] = w;
        sc[0] = w;
        sc[1] = (w != 0.0) ? w * tmp : 0.0;               // convergence.py:35-36 (zeros masked)
        sc[2] = w * w;
        sc[4] = (isinf(w) && !isinf(tmp)) ? 1.0 : 0.0;    // math.exp OverflowError
    }
    if (valid) sc[3] = (a.sample_w != nullptr) ? a.sample_w[n] * lse : lse;   // pmc.pyx:388-391
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}

// ---------------------------------------------------------------------------------------------
// k_resp: responsibilities in tile-major layout.  Pass 1 = a_nk + streaming log-sum-exp; pass 2 =
// normalisation.  Between the passes one double per (sample, component) is parked in the output
// buffer itself (a lane re-reads only what it wrote: no synchronisation).
//
// What is parked is the ONE exponential the streaming log-sum-exp computes per step anyway:
//     a_k <= m:  e_k = exp(a_k - m)           (the term added to s)          parked as  +e_k
//     a_k >  m:  f_k = exp(m - a_k), m := a_k (the factor rescaling s)       parked as  -f_k
// so exp(a_k - m_final) = (e_k or 1) * prod_{j > k, j a new maximum} f_j, and pass 2 -- walking the
// components downwards with the running product -- needs no second exp per pair: ~10 instead of
// ~45 vector instructions.  sum_k e_k (a_k - m) for E[log q(Z)] is carried through pass 1 like s.
// The PMC kinds multiply the product by exp(m_final) first, so that rho = exp(log q_k) w_k /
// (exp(lse) + tiny) underflows where the reference's exp(log q_k) does.  a_k itself is parked, and
// pass 2 evaluates the reference's expressions literally, only when the caller wants the N x K
// matrix log_rho (materialised on demand, never in the E-step itself).
// ---------------------------------------------------------------------------------------------
extern __shared__ double resp_park[];                     // PMC_A_WAVES x klds x 64 doubles

template <int D, bool PADDED, int KIND>
__global__ __launch_bounds__(PMC_A_WAVES * 64, pmc_min_waves(D)) void k_resp(const PmcArgsA a)
{
    constexpr int T = pmc_tri(D), STRIDE = pmc_pack_stride_c(D);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * PMC_A_WAVES +
                           __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n = tile * 64 + lane;
    const bool valid = n < a.N;
    const bool tile_live = tile * 64 < a.N;               // wave-uniform
    const int K = a.K;

    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (!tile_live) {
        for (int k = 0; k < K; ++k) component_sync();     // keep the workgroup's barrier count
    } else {
        double xv[D];
        load_row<D, PADDED>(a.x, n, a.N, a.dreal, xv);
        double *ut = a.u + (size_t)tile * K * 64 + lane;
        // parking place between the passes: LDS for the first klds components (the ones pass 2, which
        // walks downwards, would find evicted from L2), the output buffer itself for the rest
        const int klds = a.klds;
        double *pl = resp_park + (size_t)(threadIdx.x >> 6) * klds * 64 + lane;
        double *mt = (KIND == PMC_KIND_STUDENT_T) ? a.scratch + (size_t)tile * K * 64 + lane : nullptr;
        double *vp = (KIND == PMC_KIND_STUDENT_T) ? a.vpartials + (size_t)tile * K * 2 : nullptr;

        // ---- pass 1
        // wave-uniform: park a_k itself and evaluate the reference's expressions literally (only when
        // the N x K matrix log_rho is wanted)
        const bool literal = a.log_rho != nullptr;
        // (VB starts the maximum at -1e300 instead of -DBL_MAX so that a_0 - m stays finite below)
        double m = a.max_init_zero ? 0.0 : (KIND == PMC_KIND_VB ? -1e300 : -DBL_MAX), s = 0.0, tb = 0.0;
        cdouble *pk = (cdouble *)a.pack;
        for (int k = 0; k < K; ++k, pk += STRIDE) {
            component_sync();
            touch_component<D>(pk);
            const double maha = mahalanobis<D>(xv, pk);
            double expo = 0.0;
            const double v = component_value<D, KIND>(maha, pk + D + T, expo);
            if constexpr (KIND == PMC_KIND_STUDENT_T) mt[(size_t)k * 64] = maha;
            if constexpr (KIND == PMC_KIND_VB) {
                if (a.exponent != nullptr) {
                    const long long col = ((cint64 *)pk)[D + T + 5];
                    if (valid) a.exponent[n * a.ld + col] = expo;
                }
            }
            // streaming log-sum-exp (lse_step) with its exponential kept
            const double e = exp(-fabs(v - m));
            const bool gt = v > m;
            const double w = pk[D + T + 4];
            if constexpr (KIND == PMC_KIND_VB) {
                // tb = sum_j e_j (a_j - m) relative to the running maximum (w = 1 for this kind): the
                // dominant component contributes exactly 0, so E[log q(Z)] = tb / s - log s keeps its
                // accuracy when the responsibilities are nearly one-hot.  New maximum m' = a_k:
                // every old term becomes f (e_j (a_j - m) + e_j (m - m')).
                const double dm = v - m;
                tb = gt ? e * fma(-s, dm, tb) : fma(e, dm, tb);
            }
            s = gt ? fma(s, e, w) : fma(w, e, s);
            m = gt ? v : m;
            const double parked = literal ? v : (gt ? -e : e);
            if (k < klds) pl[k * 64] = parked;            // wave-uniform branch
            else ut[(size_t)k * 64] = parked;
        }
        const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;

        // ---- pass 2, components in DESCENDING order: the values parked last are re-read first, while
        // they are still in L2, and are overwritten there before their first write-back
        pk = (cdouble *)a.pack + (size_t)(K - 1) * STRIDE;
        if constexpr (KIND == PMC_KIND_VB) {
            // variational.pyx:741-755: r = exp(log_rho - max) / norm, zeros -> tiny,
            // log_rho += log(1/norm)
            const double norm_inv = 1. / s;
            const double log_norm_inv = log(norm_inv);
            double elq = 0.0;
            if (literal) {
                for (int k = K - 1; k >= 0; --k, pk -= STRIDE) {
                    double lr = (k < klds ? pl[k * 64] : ut[(size_t)k * 64]) - m;
                    double r = exp(lr);
                    r *= norm_inv;
                    if (r == 0.0) r = TINY;
                    lr += log_norm_inv;
                    elq += r * lr;                        // variational.pyx:1003-1013
                    ut[(size_t)k * 64] = valid ? sw * r : 0.0;
                    const long long col = ((cint64 *)pk)[D + T + 5];
                    if (valid && a.r != nullptr) a.r[n * a.ld + col] = r;
                    if (valid) a.log_rho[n * a.ld + col] = lr;
                }
            } else {
                double c = norm_inv;                      // norm_inv * prod of the f_j above k
                for (int k = K - 1; k >= 0; --k, pk -= STRIDE) {
                    const double p = k < klds ? pl[k * 64] : ut[(size_t)k * 64];
                    const bool newmax = __double2hiint(p) < 0;      // sign bit (f may be -0.0)
                    double r = (newmax ? 1.0 : p) * c;
                    c = newmax ? c * -p : c;
                    if (r == 0.0) r = TINY;
                    ut[(size_t)k * 64] = valid ? sw * r : 0.0;
                    if (a.r != nullptr) {
                        const long long col = ((cint64 *)pk)[D + T + 5];
                        if (valid) a.r[n * a.ld + col] = r;
                    }
                }
                // sum_k r_k (a_k - m + log norm_inv) with sum_k r_k = 1   (variational.pyx:1003-1013)
                elq = fma(tb, norm_inv, log_norm_inv);
            }
            if (valid) sc[0] = sw * elq;
        } else {
            // pmc.pyx:36-41: rho = exp(log q_k) * w_k / (exp(log_denominator) + tiny)
            // product form: exp(log q_k) = g_k exp(m) with g_k = (e_k or 1) * prod of the f_j above k.
            // exp(m) is applied to g_k before anything else, so where the reference's exp(log q_k)
            // underflows (log q_k < -708) this product underflows with it.
            const double lse = log(s) + m;
            const double denom = exp(lse) + TINY;
            const double em = exp(m);
            double chain = 1.0;
            const long long lat = (a.mode == PMC_RESP_PMC_LATENT && valid) ? a.latent[n] : -1;
            for (int k = K - 1; k >= 0; --k, pk -= STRIDE) {
                cdouble *c = pk + D + T;
                const long long col = ((cint64 *)pk)[D + T + 5];
                double rho;
                if (a.mode == PMC_RESP_PMC_LATENT) {
                    rho = (lat == col) ? 1. : 0.;         // pmc.pyx:49-50
                } else if (literal) {
                    rho = exp(k < klds ? pl[k * 64] : ut[(size_t)k * 64]) * c[4];
                    rho /= denom;
                } else {
                    const double p = k < klds ? pl[k * 64] : ut[(size_t)k * 64];
                    const bool newmax = __double2hiint(p) < 0;
                    rho = (newmax ? 1.0 : p) * chain * em * c[4];
                    chain = newmax ? chain * -p : chain;
                    rho /= denom;
                }
                if (valid && a.r != nullptr) a.r[n * a.ld + col] = rho;
                const double wr = valid ? sw * rho : 0.0;
                if constexpr (KIND == PMC_KIND_STUDENT_T) {
                    const double maha = mt[(size_t)k * 64];
                    const double nu = c[3];
                    const double gamma = (nu + (double)a.dreal) / (nu + maha);   // pmc.pyx:610
                    ut[(size_t)k * 64] = wr * gamma;
                    // per-wavefront sums of sample_w*rho and sample_w*rho*log(.5(maha+nu)), the
                    // N-sized parts of pmc.pyx:612 (alpha) and :669 (dof condition)
                    const double s1 = wave_sum(wr);
                    const double s2 = wave_sum(wr * log(.5 * (maha + nu)));
                    if (lane == 0) {
                        vp[2 * k] = s1;
                        vp[2 * k + 1] = s2;
                    }
                } else {
                    ut[(size_t)k * 64] = wr;
                }
            }
            if (valid) sc[3] = sw * lse;
        }
    }
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}

template <int KIND> hipError_t launch_logpdf_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_logpdf<D_, P_, KIND>), dim3(grid), dim3(PMC_A_WAVES * 64), 0, st, a);
    return hipGetLastError();
}
template <int KIND> hipError_t launch_resp_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    const size_t lds = (size_t)PMC_A_WAVES * a.klds * 64 * sizeof(double);
    hipLaunchKernelGGL((k_resp<D_, P_, KIND>), dim3(grid), dim3(PMC_A_WAVES * 64), lds, st, a);
    return hipGetLastError();
}

}  // namespace

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_logpdf_d, PMC_D, PMC_PADDED)(int kind, const PmcArgsA &a,
                                                                             unsigned grid, hipStream_t st)
{
    switch (kind) {
    case PMC_KIND_GAUSS: return launch_logpdf_k<PMC_KIND_GAUSS>(a, grid, st);
    case PMC_KIND_STUDENT_T: return launch_logpdf_k<PMC_KIND_STUDENT_T>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
}

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_resp_d, PMC_D, PMC_PADDED)(int kind, const PmcArgsA &a,
                                                                           unsigned grid, hipStream_t st)
{
    switch (kind) {
    case PMC_KIND_GAUSS: return launch_resp_k<PMC_KIND_GAUSS>(a, grid, st);
    case PMC_KIND_STUDENT_T: return launch_resp_k<PMC_KIND_STUDENT_T>(a, grid, st);
    case PMC_KIND_VB: return launch_resp_k<PMC_KIND_VB>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
}
