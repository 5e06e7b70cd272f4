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

}  // namespace

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
