// pmc_tiles.hip -- Rao-Blackwellised PMC responsibilities from KEPT Mahalanobis forms.
//
// A PMC iteration evaluates the proposal's component densities twice on the same samples: once for the
// importance weights (importance_sampling.py:197-215 through mixture.pyx:112-156) and once more inside the
// update (pmc.pyx:23-43 calculate_rho_rb evaluates every component again; student_t_pmc additionally needs
// the Mahalanobis form itself, pmc.pyx:602-610).  k_logpdf can keep maha_nk tile-major (PmcArgsA::atile);
// this kernel then forms a_nk, rho [and gamma, the dof sums] without touching x or the quadratic forms:
// 8 K bytes per sample read three times and written once instead of K (D^2 + 4 D + 40) flops -- at D = 40,
// K = 128 that is 9.4 ms instead of 55 per 1.25e7 samples.
//
// Same arithmetic, in the same order, as k_resp's PMC branch (pmc_persample.hip): a_nk = component_value(maha),
// row maximum, e = exp(a - M) and s = sum w e over the components in DESCENDING order,
// rho = (e exp(M)) w * (1 / (exp(log s + M) + tiny)) -- the two paths agree bit for bit
// (tests/test_gpu_kernels.py::test_estep_from_kept_logpdf).
// One unit for all sample dimensions (compiled with -DPMC_D=1, which it does not use).
#include "pmc_device.h"

namespace {

template <int KIND>
__global__ __launch_bounds__(PMC_A_WAVES * 64) void k_resp_tiles(const PmcArgsT a)
{
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * PMC_A_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n = tile * 64 + lane;
    const bool valid = n < a.N;
    const int K = a.K;
    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (tile * 64 < a.N) {                                           // wave-uniform
        const double *mt = a.mtile + (size_t)tile * a.ld * 64 + lane;
        double *ut = a.u + (size_t)tile * K * 64 + lane;
        double *vp = (KIND == PMC_KIND_STUDENT_T) ? a.vpartials + (size_t)tile * K * 2 : nullptr;
        cdouble *pk = (cdouble *)a.pack + a.coff;                    // c0 c1 c2 c3 | weight | column of component 0
        auto value = [&](int k) {
            cdouble *c = pk + (size_t)k * a.stride;
            double expo;
            return component_value<1, KIND>(mt[(size_t)((cint64 *)c)[5] * 64], c, expo);
        };
        const ExpConst EC;
        double M = a.max_init_zero ? 0.0 : -DBL_MAX;
        RowPoison rowp;
        for (int k = 0; k < K; ++k) {
            const double v = value(k);
            M = max_f64(v, M);
            rowp.see(v);
        }
        const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;
        const double swv = valid ? sw + rowp.value() : 0.0;
        // e_nk = exp(a_nk - M) is formed twice -- for the row sum and again for rho -- rather than parked between the
        // passes: the kernel is bound by its HBM traffic (three reads of the kept forms and one write of u instead
        // of two reads, a write and a read of e, and the write of u: 12.3 -> 9.4 ms at K = 128, N = 1.25e7)
        double s = 0.0;
        for (int k = K - 1; k >= 0; --k) {
            const double e = exp_clamped(max_f64(value(k) - M, -1075.0), EC);
            s += pk[(size_t)k * a.stride + 4] * e;                   // _regularize.pyx:79
        }
        const double lse = log_any(s) + M;                           // _regularize.pyx:81
        const double denom = exp(lse) + TINY;                        // pmc.pyx:41
        const double em = exp(M), inv_denom = 1. / denom;
        for (int k = K - 1; k >= 0; --k) {
            cdouble *c = pk + (size_t)k * a.stride;
            const double maha = mt[(size_t)((cint64 *)c)[5] * 64];
            double expo;
            const double e = exp_clamped(max_f64(component_value<1, KIND>(maha, c, expo) - M, -1075.0), EC);
            const double rho = ((e * em) * c[4]) * inv_denom;
            const double wr = swv * rho;
            if constexpr (KIND == PMC_KIND_STUDENT_T) {
                const double nu = c[3];
                const double gamma = (nu + (double)a.dreal) / (nu + maha);   // pmc.pyx:610
                ut[(size_t)k * 64] = wr * gamma;
                const double s1 = wave_sum(wr);                              // pmc.pyx:612 / :669, as in k_resp
                const double s2 = wave_sum(wr * log_pos(.5 * (maha + nu)));
                if (lane == 0) {
                    vp[2 * k] = s1;
                    vp[2 * k + 1] = s2;
                }
            } else {
                ut[(size_t)k * 64] = wr;
            }
        }
        sc[3] = swv * lse;                                           // pmc.pyx:388-391
    }
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}

// ---------------------------------------------------------------------------------------------
// k_dof_sums: the N-sized sums of student_t_pmc's degree-of-freedom condition (pmc.pyx:612 and :654-691),
//     s1_k = sum_n w_n rho_nk,      s2_k = sum_n w_n rho_nk log((maha_nk + nu_k) / 2),
// from what the emitting pass of k_mgemm left behind (round 6).  That pass writes u'_nk = rho'_nk gamma_nk relative to its
// pass's maximum before the row's log-sum-exp is known, so the sums -- which need the row's factor -- cannot be formed
// there; they would be two more N x K arrays if they were left to the statistics kernel.  But one number per pair is
// enough: with t = 1 + maha / nu,
//     u = gscale u' = F_n w_k exp(c0 + c1 log t) ((nu + D) / nu) / t,          F_n = w_n / (exp(log q_n) + tiny)
//     => log t = (log u - log F_n - kappa_k) / (c1 - 1),    kappa_k = c0 + log w_k + log((nu + D) / nu)   (c1 - 1 <= -3/2)
//     w_n rho_nk = u t nu / (nu + D),      log((maha + nu) / 2) = log t + log(nu / 2)
// -- a logarithm and an exponential per pair, 8 K bytes per sample read once.  The same formula holds for the workgroups
// the guard sent to the exact kernel (their u is complete, their factors are ones).  log t inherits |log u| eps / |c1 - 1|
// ~ 3e-15.  Grid: (groups of 16 components, chunks of tiles); fixed summation order.
// ---------------------------------------------------------------------------------------------
// (Work layout: a workgroup owns one group of 16 components and a chunk of tiles; it takes the chunk in rounds of 32 tiles,
//  8 per wavefront.  Per round a wavefront first forms log(gscale / F_n) of its 8 x 64 samples -- three library logarithms /
//  exponentials per sample, not per pair -- into LDS (every lane reads back only what it wrote), then walks the 16
//  components with two accumulators.  The first build kept 32 accumulators per lane and all 16 loads of a tile in flight:
//  282 registers, one wavefront per SIMD, 2.8 ms per 2e6 samples x 128 components; this one: see profiles/r06_dof_sums.txt.)
__global__ __launch_bounds__(256, 4) void k_dof_sums(const PmcArgsV a)
{
    constexpr int GS = PMC_RESP_GROUP, TW = 8;
    __shared__ double red[4][2 * GS];
    __shared__ double cst[GS][4];                         // kappa, 1 / (c1 - 1), nu / (nu + D), log(nu / 2) per component
    __shared__ double s_lgs[4][TW][64], s_gsc[4][TW][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x, k0 = g * GS, kn = a.K - k0 < GS ? a.K - k0 : GS;
    const int G = (a.K + GS - 1) / GS;
    const ExpConst EC;
    if ((int)threadIdx.x < kn) {
        const double *c = a.pack + (size_t)(k0 + threadIdx.x) * a.stride + a.coff;
        const double nu = c[3];
        cst[threadIdx.x][0] = (c[0] + log(c[4])) + log(-2.0 * c[1] / nu);
        cst[threadIdx.x][1] = 1.0 / (c[1] - 1.0);
        cst[threadIdx.x][2] = nu / (nu + (double)a.dreal);
        cst[threadIdx.x][3] = log(.5 * nu);
    }
    if (threadIdx.x < 4 * 2 * GS) (&red[0][0])[threadIdx.x] = 0.0;
    __syncthreads();
    const long long t0 = (long long)blockIdx.y * a.tiles_per_chunk;
    long long t1 = t0 + a.tiles_per_chunk;
    if (t1 > a.ntiles) t1 = a.ntiles;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (long long sub = t0; sub < t1; sub += 4 * TW) {
        const long long tw0 = sub + (long long)wave * TW;  // this wavefront's tiles of the round: tw0 ... tw0 + TW - 1
        // ---- per sample: log(gscale / F_n), F_n = w_n / (exp(log q_n) + tiny) -- and what kind of row it is
        for (int i = 0; i < TW; ++i) {
            const long long tile = tw0 + i;
            if (tile >= t1) break;                         // wave-uniform
            const long long n = tile * 64 + lane;
            const bool valid = n < a.N;
            double lF = 0.0, gsc = 0.0;
            if (valid) {
                lF = log(a.weights[n] / (exp(a.lse[n]) + TINY));
                gsc = a.gscale[((size_t)tile * G + g) * 64 + lane];
            }
            const double lgs = log(gsc) - lF;
            // (F = 0: a sample without weight, gscale = 0: nothing to add; NaN: a poisoned row -- its sums become NaN as the
            //  reference's do)
            const bool row_ok = valid && gsc > 0.0 && lgs == lgs && fabs(lgs) <= DBL_MAX;
            const bool row_nan = valid && (gsc != gsc || lF != lF);
            s_lgs[wave][i][lane] = row_ok ? lgs : 0.0;
            s_gsc[wave][i][lane] = row_nan ? qnan : (row_ok ? gsc : 0.0);
        }
        // ---- per component of the group: the two sums over the wavefront's samples of the round
        for (int j = 0; j < kn; ++j) {
            const double kappa = cst[j][0], ic1m = cst[j][1], cn = cst[j][2], lnu2 = cst[j][3];
            const double *uj = a.u + ((size_t)tw0 * a.K + k0 + j) * 64 + lane;
            double a1 = 0.0, a2 = 0.0;
#pragma unroll 4
            for (int i = 0; i < TW; ++i) {
                if (tw0 + i < t1) {                        // wave-uniform
                    const double up = uj[(size_t)i * a.K * 64];
                    const double gsc = s_gsc[wave][i][lane];
                    const bool live = gsc > 0.0 && up > 0.0;         // (u' = 0: the pair underflowed against its pass's maximum)
                    double lt = ((log_any(live ? up : 1.0) + s_lgs[wave][i][lane]) - kappa) * ic1m;
                    lt = max_f64(lt, 0.0);                           // t = 1 + maha / nu >= 1
                    const double wr = ((gsc * up) * exp_clamped(lt < 700.0 ? lt : 700.0, EC)) * cn;
                    const double bad = (gsc != gsc || up != up) ? qnan : 0.0;
                    a1 += live ? wr : bad;
                    a2 += live ? wr * (lt + lnu2) : bad;
                }
            }
            a1 = wave_sum(a1);
            a2 = wave_sum(a2);
            if (lane == 0) {
                red[wave][2 * j] += a1;
                red[wave][2 * j + 1] += a2;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * kn) {
        const int j = threadIdx.x;
        a.vpartials[((size_t)blockIdx.y * a.K + k0) * 2 + j] = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
    }
}

}  // namespace

extern "C" hipError_t pmc_launch_dof_sums(const PmcArgsV &a, unsigned ngroups, unsigned nchunks, hipStream_t st)
{
    hipLaunchKernelGGL(k_dof_sums, dim3(ngroups, nchunks), dim3(256), 0, st, a);
    return hipGetLastError();
}

extern "C" hipError_t pmc_launch_resp_tiles(int kind, const PmcArgsT &a, unsigned grid, hipStream_t st)
{
    if (kind == PMC_KIND_GAUSS)
        hipLaunchKernelGGL(k_resp_tiles<PMC_KIND_GAUSS>, dim3(grid), dim3(PMC_A_WAVES * 64), 0, st, a);
    else if (kind == PMC_KIND_STUDENT_T)
        hipLaunchKernelGGL(k_resp_tiles<PMC_KIND_STUDENT_T>, dim3(grid), dim3(PMC_A_WAVES * 64), 0, st, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
