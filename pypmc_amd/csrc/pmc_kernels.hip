// pmc_kernels.hip -- the gfx950 kernels of the adaptive-importance-sampling hot path, compiled
// once per sample dimension:  hipcc -DPMC_D=<D> -DPMC_PADDED=<0|1> -c pmc_kernels.hip
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
#include "pmc_internal.h"
#include "../../include/pmc_hip.h"

#include <cfloat>
#include <type_traits>

#ifndef PMC_D
#error "compile with -DPMC_D=<dimension>"
#endif
#ifndef PMC_PADDED
#define PMC_PADDED 0
#endif

namespace {

constexpr double TINY = 2.2250738585072014e-308;  // numpy.finfo('d').tiny

typedef __attribute__((address_space(4))) const double cdouble;        // scalar-cache loads
typedef __attribute__((address_space(4))) const long long cint64;

template <int I> using ic = std::integral_constant<int, I>;
template <int B, int E, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

// ---------------------------------------------------------------------------------------------
// sample row -> registers
// ---------------------------------------------------------------------------------------------
template <int D, bool PADDED>
__device__ __forceinline__ void load_row(const double *__restrict__ x, long long n, long long N,
                                         int dreal, double (&xv)[D])
{
    if (n < N) {
        if constexpr (!PADDED) {
            const double *p = x + n * D;
#pragma unroll
            for (int j = 0; j < D; ++j) xv[j] = p[j];
        } else {
            const double *p = x + n * (long long)dreal;
#pragma unroll
            for (int j = 0; j < D; ++j) xv[j] = j < dreal ? p[j] : 0.0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) xv[j] = 0.0;
    }
}

// maha = |R (x - mu)|^2 ; R upper triangular, packed row-major in consumption order.
// Replaces bilinear_sym(inv_sigma, x - mu) (pypmc/tools/_linalg.pyx:10-39).
template <int D> __device__ __forceinline__ double mahalanobis(const double (&xv)[D], cdouble *pk)
{
    double d[D];
#pragma unroll
    for (int j = 0; j < D; ++j) d[j] = xv[j] - pk[j];
    double maha = 0.0;
    int idx = D;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double y = 0.0;
#pragma unroll
        for (int j = i; j < D; ++j) y = fma(pk[idx++], d[j], y);
        maha = fma(y, y, maha);
    }
    return maha;
}

// a_nk from maha_nk, in the reference's operation order (see enum pmc_kind).
template <int D, int KIND>
__device__ __forceinline__ double component_value(double maha, cdouble *c, double &expo)
{
    if constexpr (KIND == PMC_KIND_GAUSS) {
        return c[0] - 0.5 * maha;                       // gauss.pyx:151
    } else if constexpr (KIND == PMC_KIND_STUDENT_T) {
        double t = maha;                                  // student_t.pyx:159-164
        t *= c[2];
        t += 1.;
        t = log(t);
        t *= c[1];
        t += c[0];
        return t;
    } else {
        expo = c[0] + c[1] * maha;                        // variational.pyx:798
        return c[2] + 0.5 * (c[3] - expo);                // variational.pyx:691
    }
}

// One step of the streaming log-sum-exp  log sum_k w_k exp(a_k) = m + log s  with
// m = running max, s = sum_k w_k exp(a_k - m)   (one exp per step).
__device__ __forceinline__ void lse_step(double a, double w, double &m, double &s)
{
    const double e = exp(-fabs(a - m));
    const bool gt = a > m;
    s = gt ? fma(s, e, w) : fma(w, e, s);
    m = gt ? a : m;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block (PMC_A_WAVES wavefronts) reduction of NS per-lane scalars -> partials[block*PMC_NSCALARS+i]
template <int NS>
__device__ __forceinline__ void block_scalars(double (&sc)[NS], double *partials)
{
    __shared__ double red[PMC_A_WAVES][PMC_NSCALARS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double v = wave_sum(sc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < PMC_NSCALARS) {
        double v = 0.0;
        if (threadIdx.x < NS) {
#pragma unroll
            for (int w = 0; w < PMC_A_WAVES; ++w) v += red[w][threadIdx.x];
        }
        partials[(size_t)blockIdx.x * PMC_NSCALARS + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// k_logpdf: MixtureDensity.multi_evaluate (mixture.pyx:112-156) + logsumexp2D
// (_regularize.pyx:57-84) [+ importance weights, importance_sampling.py:197-215] in one pass.
// ---------------------------------------------------------------------------------------------
template <int D, bool PADDED, int KIND>
__global__ __launch_bounds__(PMC_A_WAVES * 64) void k_logpdf(const PmcArgsA a)
{
    constexpr int T = pmc_tri(D), STRIDE = pmc_pack_stride_c(D);
    const long long n = ((long long)blockIdx.x * PMC_A_WAVES * 64) + threadIdx.x;
    const bool valid = n < a.N;

    double xv[D];
    load_row<D, PADDED>(a.x, n, a.N, a.dreal, xv);

    double m = a.max_init_zero ? 0.0 : -DBL_MAX, s = 0.0;
    cdouble *pk = (cdouble *)a.pack;
    for (int k = 0; k < a.K; ++k, pk += STRIDE) {
        const double maha = mahalanobis<D>(xv, pk);
        double expo;
        const double v = component_value<D, KIND>(maha, pk + D + T, expo);
        if (a.individual != nullptr) {
            const long long col = ((cint64 *)pk)[D + T + 5];
            if (valid) a.individual[n * a.ld + col] = v;
        }
        lse_step(v, pk[D + T + 4], m, s);
    }
    const double lse = log(s) + m;                       // _regularize.pyx:81
    if (a.out != nullptr && valid) a.out[n] = lse;

    if (a.partials == nullptr && a.log_target == nullptr) return;

    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (a.log_target != nullptr && valid) {
        const double tmp = a.log_target[n] - lse;        // importance_sampling.py:204
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
// k_resp: responsibilities in tile-major layout.  Pass 1 = a_nk (+ streaming log-sum-exp),
// parked in the output buffer itself; pass 2 = normalisation.  A lane re-reads only what it
// wrote, so no synchronisation is needed between the passes.
// ---------------------------------------------------------------------------------------------
template <int D, bool PADDED, int KIND>
__global__ __launch_bounds__(PMC_A_WAVES * 64) void k_resp(const PmcArgsA a)
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
    if (tile_live) {
        double xv[D];
        load_row<D, PADDED>(a.x, n, a.N, a.dreal, xv);
        double *ut = a.u + (size_t)tile * K * 64 + lane;
        double *v1t = (KIND == PMC_KIND_STUDENT_T) ? a.v1 + (size_t)tile * K * 64 + lane : nullptr;
        double *v2t = (KIND == PMC_KIND_STUDENT_T) ? a.v2 + (size_t)tile * K * 64 + lane : nullptr;

        // ---- pass 1
        double m = a.max_init_zero ? 0.0 : -DBL_MAX, s = 0.0;
        cdouble *pk = (cdouble *)a.pack;
        for (int k = 0; k < K; ++k, pk += STRIDE) {
            const double maha = mahalanobis<D>(xv, pk);
            double expo = 0.0;
            const double v = component_value<D, KIND>(maha, pk + D + T, expo);
            ut[(size_t)k * 64] = v;
            if constexpr (KIND == PMC_KIND_STUDENT_T) v1t[(size_t)k * 64] = maha;
            if constexpr (KIND == PMC_KIND_VB) {
                if (a.exponent != nullptr) {
                    const long long col = ((cint64 *)pk)[D + T + 5];
                    if (valid) a.exponent[n * a.ld + col] = expo;
                }
            }
            lse_step(v, pk[D + T + 4], m, s);
        }
        const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;

        // ---- pass 2
        pk = (cdouble *)a.pack;
        if constexpr (KIND == PMC_KIND_VB) {
            // variational.pyx:741-755: r = exp(log_rho - max) / norm, zeros -> tiny,
            // log_rho += log(1/norm)
            const double norm_inv = 1. / s;
            const double log_norm_inv = log(norm_inv);
            double elq = 0.0;
            for (int k = 0; k < K; ++k, pk += STRIDE) {
                double lr = ut[(size_t)k * 64] - m;
                double r = exp(lr);
                r *= norm_inv;
                if (r == 0.0) r = TINY;
                lr += log_norm_inv;
                elq += r * lr;                            // variational.pyx:1003-1013
                ut[(size_t)k * 64] = valid ? sw * r : 0.0;
                if (a.r != nullptr || a.log_rho != nullptr) {
                    const long long col = ((cint64 *)pk)[D + T + 5];
                    if (valid && a.r != nullptr) a.r[n * a.ld + col] = r;
                    if (valid && a.log_rho != nullptr) a.log_rho[n * a.ld + col] = lr;
                }
            }
            if (valid) sc[0] = sw * elq;
        } else {
            // pmc.pyx:36-41: rho = exp(log q_k) * w_k / (exp(log_denominator) + tiny)
            const double lse = log(s) + m;
            const double denom = exp(lse) + TINY;
            const long long lat = (a.mode == PMC_RESP_PMC_LATENT && valid) ? a.latent[n] : -1;
            for (int k = 0; k < K; ++k, pk += STRIDE) {
                cdouble *c = pk + D + T;
                const long long col = ((cint64 *)pk)[D + T + 5];
                double rho;
                if (a.mode == PMC_RESP_PMC_LATENT) {
                    rho = (lat == col) ? 1. : 0.;         // pmc.pyx:49-50
                } else {
                    rho = exp(ut[(size_t)k * 64]) * c[4];
                    rho /= denom;
                }
                if (valid && a.r != nullptr) a.r[n * a.ld + col] = rho;
                const double wr = valid ? sw * rho : 0.0;
                if constexpr (KIND == PMC_KIND_STUDENT_T) {
                    const double maha = v1t[(size_t)k * 64];
                    const double nu = c[3];
                    const double gamma = (nu + (double)a.dreal) / (nu + maha);   // pmc.pyx:610
                    ut[(size_t)k * 64] = wr * gamma;
                    v1t[(size_t)k * 64] = wr;
                    v2t[(size_t)k * 64] = wr * log(.5 * (maha + nu));             // pmc.pyx:669
                } else {
                    ut[(size_t)k * 64] = wr;
                }
            }
            if (valid) sc[3] = sw * lse;
        }
    }
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}

// ---------------------------------------------------------------------------------------------
// k_stats: per component  sum u | sum u d | sum u d d^T (lower) | sum v1 | sum v2,  d = x - mu_k.
// ---------------------------------------------------------------------------------------------
// Rows i and D-1-i form a pair of D+1 lower-triangle elements; pairs are dealt round-robin to the
// NSUB subsets so that every subset carries (almost) the same number of accumulators.
template <int D, int NSUB> __host__ __device__ constexpr int row_owner(int i)
{
    const int p = (i < D - 1 - i) ? i : D - 1 - i;
    return p % NSUB;
}

template <int D, bool PADDED, int NSUB, int WAVES, int SUB>
__device__ __forceinline__ void stats_run(const PmcArgsB &b, double *xs, int k, long long t0,
                                          long long t1, int chunk)
{
    constexpr int T = pmc_tri(D), STRIDE = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    constexpr int NT = WAVES * 64;
    constexpr int LDP = 65;                               // padded tile row: conflict-free b64 access
    const int tid = threadIdx.x, lane = tid & 63;
    const bool active = SUB >= 0;
    const int dreal = PADDED ? b.dreal : D;
    const long long total = b.N * (long long)dreal;

    double acc0 = 0.0, accv1 = 0.0, accv2 = 0.0;
    double acc1[D];
    double acc2[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        acc1[i] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) acc2[i][j] = 0.0;
    }
    cdouble *pk = (cdouble *)b.pack + (size_t)(active ? k : 0) * STRIDE;

    if constexpr (PADDED) {
        for (int e = tid; e < 2 * D * LDP; e += NT) xs[e] = 0.0;
        __syncthreads();
    }
    int buf = 0;
    for (long long t = t0; t < t1; ++t, buf ^= 1) {
        double *xb = xs + buf * (D * LDP);
        // coalesced load of the 64 x dreal tile, transposed into LDS as [j][n]
        const long long base = t * 64 * dreal;
        for (int e = tid; e < 64 * dreal; e += NT) {
            const long long g = base + e;
            const int nloc = PADDED ? e / dreal : e / D;
            const int j = PADDED ? e % dreal : e % D;
            xb[j * LDP + nloc] = (g < total) ? b.x[g] : 0.0;
        }
        __syncthreads();
        if constexpr (SUB >= 0) {
            const size_t uo = ((size_t)t * b.K + k) * 64 + lane;
            const double u = b.u[uo];
            if constexpr (SUB == 0) {
                acc0 += u;
                if (b.v1 != nullptr) {
                    accv1 += b.v1[uo];
                    accv2 += b.v2[uo];
                }
            }
            double d[D];
#pragma unroll
            for (int j = 0; j < D; ++j) d[j] = xb[j * LDP + lane] - pk[j];
            static_for<0, D>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (row_owner<D, NSUB>(i) == SUB) {
                    const double ud = u * d[i];
                    acc1[i] += ud;
                    static_for<0, i + 1>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        acc2[i][j] = fma(ud, d[j], acc2[i][j]);
                    });
                }
            });
        }
        // double-buffered tile: the next iteration writes the other buffer, and the barrier of
        // that iteration orders those writes against this iteration's reads of it two tiles on.
    }

    if constexpr (SUB >= 0) {
        double *out = b.partials + ((size_t)chunk * b.K + k) * PS;
        if constexpr (SUB == 0) {
            const double s0 = wave_sum(acc0), s1 = wave_sum(accv1), s2 = wave_sum(accv2);
            if (lane == 0) {
                out[0] = s0;
                out[1 + D + T] = s1;
                out[2 + D + T] = s2;
            }
        }
        static_for<0, D>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (row_owner<D, NSUB>(i) == SUB) {
                const double s = wave_sum(acc1[i]);
                if (lane == 0) out[1 + i] = s;
                static_for<0, i + 1>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const double q = wave_sum(acc2[i][j]);
                    if (lane == 0) out[1 + D + i * (i + 1) / 2 + j] = q;
                });
            }
        });
    }
}

template <int D, bool PADDED, int NSUB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_stats(const PmcArgsB b)
{
    extern __shared__ double xs[];                        // 2 * D * 65 doubles (double buffer)
    // XCD-aware block -> (chunk, task group): hardware places block i on XCD i % 8; all task
    // groups of one sample chunk get the same residue so they share that XCD's L2 copy of the tile.
    const int bid = blockIdx.x;
    const int q = bid >> 3;
    const int chunk = (bid & 7) + 8 * (q / b.ngroups);
    const int group = q % b.ngroups;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR: uniform k, sub
    const int task = group * WAVES + wave;
    const int k = task / NSUB, sub = task % NSUB;
    const long long t0 = (long long)chunk * b.tiles_per_chunk;
    long long t1 = t0 + b.tiles_per_chunk;
    if (t1 > b.ntiles) t1 = b.ntiles;
    // (an empty chunk still publishes zeros, so the finishing kernel sums every chunk blindly)
    if (k >= b.K) {
        stats_run<D, PADDED, NSUB, WAVES, -1>(b, xs, 0, t0, t1, chunk);
        return;
    }
    bool done = false;
    static_for<0, NSUB>([&](auto S) {
        constexpr int SUBC = decltype(S)::value;
        if (!done && sub == SUBC) {
            stats_run<D, PADDED, NSUB, WAVES, SUBC>(b, xs, k, t0, t1, chunk);
            done = true;
        }
    });
}

// ---------------------------------------------------------------------------------------------
// per-dimension configuration and launchers
// ---------------------------------------------------------------------------------------------
constexpr int D_ = PMC_D;
constexpr bool P_ = PMC_PADDED != 0;

// Accumulators per lane are about per*(D+2) fp64 values, per = row pairs in the fullest subset.
// With 8 wavefronts per workgroup the register budget is 256 VGPRs = accumulators*2 + 2*D (d) + ~30.
constexpr int stats_nsub(int D)
{
    const int pairs = (D + 1) / 2;                       // row pairs of D+1 elements each
    const int budget = D <= 24 ? 56 : (D <= 36 ? 2 * (D + 2) : D + 2);
    for (int n = 1; n <= pairs; ++n) {
        const int per = (pairs + n - 1) / n;
        if (per * (D + 2) <= budget) return n;
    }
    return pairs;
}
constexpr int NSUB_ = stats_nsub(D_);
constexpr int SW_ = D_ >= 48 ? 4 : 8;                    // wavefronts per statistics workgroup

template <int KIND> hipError_t launch_logpdf_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_logpdf<D_, P_, KIND>), dim3(grid), dim3(PMC_A_WAVES * 64), 0, st, a);
    return hipGetLastError();
}
template <int KIND> hipError_t launch_resp_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_resp<D_, P_, KIND>), dim3(grid), dim3(PMC_A_WAVES * 64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_logpdf(int kind, const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    switch (kind) {
    case PMC_KIND_GAUSS: return launch_logpdf_k<PMC_KIND_GAUSS>(a, grid, st);
    case PMC_KIND_STUDENT_T: return launch_logpdf_k<PMC_KIND_STUDENT_T>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
}
hipError_t launch_resp(int kind, const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    switch (kind) {
    case PMC_KIND_GAUSS: return launch_resp_k<PMC_KIND_GAUSS>(a, grid, st);
    case PMC_KIND_STUDENT_T: return launch_resp_k<PMC_KIND_STUDENT_T>(a, grid, st);
    case PMC_KIND_VB: return launch_resp_k<PMC_KIND_VB>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
}
hipError_t launch_stats(const PmcArgsB &b, unsigned grid, hipStream_t st)
{
    constexpr size_t lds = sizeof(double) * 2 * D_ * 65;
    if constexpr (lds > 65536) {
        static const hipError_t once = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&k_stats<D_, P_, NSUB_, SW_>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((k_stats<D_, P_, NSUB_, SW_>), dim3(grid), dim3(SW_ * 64), lds, st, b);
    return hipGetLastError();
}

}  // namespace

#define PMC_CAT3(a, b, c) a##b##c
#define PMC_KSET_NAME(d, p) PMC_CAT3(pmc_kset_##d, _p, p)
#define PMC_KSET_NAME_X(d, p) PMC_KSET_NAME(d, p)

// host-side accessor picked up by the dispatcher in pmc_api.hip
extern "C" const PmcKernelSet *PMC_KSET_NAME_X(PMC_D, PMC_PADDED)(void)
{
    static const PmcKernelSet set = {D_,          P_ ? 1 : 0,   NSUB_,        SW_,
                                     &launch_logpdf, &launch_resp, &launch_stats};
    return &set;
}
