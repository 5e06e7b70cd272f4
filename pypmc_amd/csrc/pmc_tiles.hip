// pmc_tiles.hip -- Rao-Blackwellised PMC responsibilities from KEPT component log-densities.
//
// A PMC iteration evaluates the proposal's component densities twice on the same samples: once for the
// importance weights (importance_sampling.py:197-215 through mixture.pyx:112-156) and once more inside the
// update (pmc.pyx:23-43 calculate_rho_rb evaluates every component again).  k_logpdf can keep its a_nk
// tile-major (PmcArgsA::atile); this kernel then forms rho without touching x or the Mahalanobis forms:
// 8 K bytes per sample read twice and written once instead of K (D^2 + 4 D + 40) flops -- at D = 40, K = 128
// that is 6 ms instead of 55 per 1.25e7 samples.
//
// Same arithmetic, in the same order, as k_resp's PMC branch (pmc_persample.hip): row maximum, e = exp(a - M)
// and s = sum w e over the components in DESCENDING order, rho = (e exp(M)) w / (exp(log s + M) + tiny) -- the
// two paths agree bit for bit (tests/test_gpu_kernels.py::test_estep_from_kept_logpdf).
// One unit for all sample dimensions (compiled with -DPMC_D=1, which it does not use).
#include "pmc_device.h"

namespace {

__global__ __launch_bounds__(PMC_A_WAVES * 64) void k_resp_tiles(const PmcArgsT a)
{
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * PMC_A_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n = tile * 64 + lane;
    const bool valid = n < a.N;
    const int K = a.K;
    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (tile * 64 < a.N) {                                           // wave-uniform
        const double *at = a.atile + (size_t)tile * a.ld * 64 + lane;
        double *ut = a.u + (size_t)tile * K * 64 + lane;
        cdouble *pk = (cdouble *)a.pack + a.woff;
        const ExpConst EC;
        double M = a.max_init_zero ? 0.0 : -DBL_MAX, poison = 0.0;
        for (int k = 0; k < K; ++k) {
            const long long col = ((cint64 *)(pk + (size_t)k * a.stride))[1];
            const double v = at[(size_t)col * 64];
            M = max_f64(v, M);
            poison = fma(0.0, v, poison);
        }
        const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;
        const double swv = valid ? sw + poison : 0.0;
        double s = 0.0;
        for (int k = K - 1; k >= 0; --k) {
            cdouble *c = pk + (size_t)k * a.stride;
            const double v = at[(size_t)((cint64 *)c)[1] * 64];
            const double lr = max_f64(v - M, -1075.0);
            const double e = exp_clamped(lr, EC);
            s += c[0] * e;                                           // _regularize.pyx:79
            ut[(size_t)k * 64] = e;
        }
        const double lse = log(s) + M;                               // _regularize.pyx:81
        const double denom = exp(lse) + TINY;                        // pmc.pyx:41
        const double em = exp(M);
        for (int k = K - 1; k >= 0; --k) {
            double rho = (ut[(size_t)k * 64] * em) * pk[(size_t)k * a.stride];
            rho /= denom;
            ut[(size_t)k * 64] = swv * rho;
        }
        sc[3] = swv * lse;                                           // pmc.pyx:388-391
    }
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}

}  // namespace

extern "C" hipError_t pmc_launch_resp_tiles(const PmcArgsT &a, unsigned grid, hipStream_t st)
{
    hipLaunchKernelGGL(k_resp_tiles, dim3(grid), dim3(PMC_A_WAVES * 64), 0, st, a);
    return hipGetLastError();
}
