// pmc_convert.h -- pmc_host_convert_stats on the device, as a function of one workgroup of 256 threads per component, shared by
// k_convert_stats (pmc_api.hip) and the VB state's fused conversion (pmc_vbstate.hip): the same operations in the same order as
// the host function, the same bits.
#pragma once
#include <hip/hip_runtime.h>

#ifndef PMC_CONVERT_NSCALARS
#define PMC_CONVERT_NSCALARS 8
#endif

// out = [S0 K | M1 K D | mean K D | cov K D D | far K (0 / 1 per component) | scalars 8 (a copy of `scalars`, or zeros)]:
// everything a caller reads after an E-step in ONE block of memory.  blockIdx.x = component, blockDim.x = 256.
__device__ inline void pmc_convert_stats_block(const double *stats, const double *shift, const double *ncov, int K, int D,
                                               const double *scalars, double *out)
{
    __shared__ double s0s[1024];
    __shared__ double total_s;
    const int k = blockIdx.x, PS = 1 + D + D * (D + 1) / 2;
    const double *b = stats + (size_t)k * PS;
    double *S0 = out, *M1 = S0 + K, *mean = M1 + (size_t)K * D, *cov = mean + (size_t)K * D, *far = cov + (size_t)K * D * D;
    const double tiny = 2.2250738585072014e-308;
    // the total over the components, added in their order (shift_is_far's threshold): loads side by side, one lane adds
    double total = 0.0;
    for (int q0 = 0; q0 < K; q0 += 1024) {
        const int nq = K - q0 < 1024 ? K - q0 : 1024;
        __syncthreads();
        for (int q = threadIdx.x; q < nq; q += 256) s0s[q] = stats[(size_t)(q0 + q) * PS];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int q = 0; q < nq; ++q)
                if (isfinite(s0s[q])) total += s0s[q];
    }
    if (threadIdx.x == 0) total_s = total;
    const double s0 = b[0];
    const double nm = s0 == 0.0 ? tiny : s0;
    const double ncr = ncov ? ncov[k] : s0;
    const double nc = ncr == 0.0 ? tiny : ncr;
    if (threadIdx.x == 0) S0[k] = s0;
    if (k == 0 && threadIdx.x < PMC_CONVERT_NSCALARS) (far + K)[threadIdx.x] = scalars ? scalars[threadIdx.x] : 0.0;
    for (int i = threadIdx.x; i < D; i += 256) {
        M1[(size_t)k * D + i] = b[1 + i];
        mean[(size_t)k * D + i] = shift[(size_t)k * D + i] + b[1 + i] / nm;
    }
    for (int e = threadIdx.x; e < D * D; e += 256) {
        const int i = e / D, jj = e % D;
        const int hi = i > jj ? i : jj, lo = i > jj ? jj : i;
        const double m2 = b[1 + D + hi * (hi + 1) / 2 + lo];
        const double prod = (b[1 + i] / nm) * (b[1 + jj] / nm);
        cov[((size_t)k * D + i) * D + jj] = (m2 - nm * prod) / nc;
    }
    __syncthreads();
    // _stats.py::shift_is_far (limit 100) for this component: the D coordinates side by side, any hit counts
    int hit = 0;
    if (isfinite(s0) && s0 > 1e-200 && s0 > 1e-6 * total_s) {
        for (int i = threadIdx.x; i < D; i += 256) {
            const double db = b[1 + i] / s0, dbar2 = db * db;
            const double raw = b[1 + D + i * (i + 1) / 2 + i] / s0;
            const double v = raw - dbar2, t = 1e-14 * raw;
            const double var = (v != v || t != t) ? v + t : (v > t ? v : t);
            if (dbar2 > 100. * var) hit = 1;
        }
    }
    hit = __syncthreads_or(hit);
    if (threadIdx.x == 0) far[k] = hit ? 1.0 : 0.0;
}
