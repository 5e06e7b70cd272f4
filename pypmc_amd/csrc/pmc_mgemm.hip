// pmc_mgemm.hip -- the Mahalanobis forms of ALL components of a mixture as one matrix product on the fp64 matrix pipe,
// with the per-sample epilogues of k_logpdf / k_resp_groups fused behind it (compiled dimensions 20, 24, 32, 40, 48, 64):
//
//     maha_nk = (x_n - mu_k)^T P_k (x_n - mu_k) = sum_m theta_km z_nm,      P_k = R_k^T R_k,
//     z_n = the (D + 1)(D + 2) / 2 monomials of d = x_n - c up to degree 2 about ONE centre c common to all components,
//     theta_k = (P_ii | 2 P_ij, i < j | -2 (P delta)_i | delta^T P delta),   delta = mu_k - c
//
// on v_mfma_f64_16x16x4_f64: A = theta (16 components x 4 monomials), B = z (4 monomials x 16 samples).  It replaces
// bilinear_sym (pypmc/tools/_linalg.pyx:10-39) called N K times by Gauss / StudentT.multi_evaluate
// (density/gauss.pyx:146-151, student_t.pyx:154-164) and by the VB exponent (mix_adapt/variational.pyx:774-798).
//
// Why: from D = 32 on the per-sample kernels' engine (4 x 4 x 4 blocks of the triangular factor, pmc_persample.hip)
// keeps the matrix pipe 67 % busy; this form is dense (97 % of the slots useful at D = 40), uses the instruction that
// holds the highest clock, and shares every monomial product among NCT component tiles: 27 instead of 35 ps per pair
// at D = 40, K = 128 (scripts/microbench/maha_gemm2.hip, profiles/r03_maha_gemm_prototype.txt).
//
// The price is numerical: the expanded form's rounding error is eps * sum_m |theta_km z_nm| -- it grows with
// |P| |x - c|^2, not with maha.  A guard prices it per sample BEFORE any work is done,
//     E_n = eps_g (Theta_1 |d_n|^2 + Theta_2 |d_n| + Theta_3),
//     Theta_1 = max_k s_k |P_k|_F,  Theta_2 = max_k 2 s_k |P_k delta_k|,  Theta_3 = max_k s_k delta_k^T P_k delta_k
// (s_k = |d a_nk / d maha_nk|: 1/2 Gauss, nu_k / 2 VB; Student-t: 1 -- its slope (nu + D) / (2 (nu + maha)) is the pair's and is
// applied in the epilogue, where a pair beyond the tolerance raises the workgroup's flag a posteriori), and a workgroup with a sample
// beyond the tolerance (or a non-finite coordinate) writes nothing but its flag: the exact kernel launched behind
// (k_logpdf / k_resp_groups with PmcArgsA::blockflag) does exactly the flagged workgroups' samples.
//
// Work decomposition: workgroup = 4 wavefronts = 4 tiles of 64 samples (the per-sample kernels' blocks, so scalar
// partials and flags line up).  The wavefront's samples sit in LDS as d[j][sample] with four views rotated by D / 4
// coordinates, one per lane group g = lane >> 4, so that ONE compile-time pair of rows per step gives the four lane
// groups four different monomials (pairs (a, a + delta), a < D / 4, delta = 0 ... D / 2; then D / 4 linear steps, one
// constant step).  theta arrives by LDS-DMA in chunks of 16 steps, double buffered, shared by the four wavefronts.
// Epilogue in the accumulator layout (lane (g, s) holds components g + 4 r of each tile for samples 16 t + s): the
// pass's 16 NCT components are one group -- maximum and sums by two cross-lane steps, u' = [w_k] exp(a - M_pass)
// written once (k_resp_groups' form with the pass as the group: the factor per (sample, group) is left to
// k_stats_gemm) -- and lane l = sample l carries the row's running maximum / sum / bound term across the passes.
#include "pmc_device.h"

namespace {

typedef double md4 __attribute__((ext_vector_type(4)));
typedef double md2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void mg_gvoid_t;
typedef __attribute__((address_space(3))) void mg_lvoid_t;

template <int D> struct MgCfg {
    // 32 ... 64: every call that qualifies.  20 and 24 (round 5): the log-pdf / importance-weight pass of mixtures with many
    // components only (pmc_api.hip::mgemm_pick) -- with 61 / 85 steps per pass the epilogue weighs three times what it does
    // at D = 40, and the emitting epilogue (the E-step) loses against k_resp_groups there (profiles/r05_mgemm_small.txt)
    static constexpr bool ENABLED = D == 20 || D == 24 || D == 32 || D == 40 || D == 48 || D == 64;
    static constexpr int Q = D / 4;
    static constexpr int ND = 2 * Q + 1;                   // deltas per a
    static constexpr int NQ = Q * ND;                      // quadratic steps
    static constexpr int NSTEP = NQ + Q + 1;
    // steps per staged chunk of theta: 16; 8 at D = 64, where the image of 256 samples (133 KB) leaves 27 KB for the
    // two theta buffers.  The chunk count is rounded up to an even number (the buffer a step reads is a compile-time
    // constant of the step: passes must start on buffer 0), padding steps carry zero coefficients.
    static constexpr int CH = D == 64 ? 8 : 16;
    static constexpr int NCH = ((NSTEP + CH - 1) / CH + 1) / 2 * 2;
    static constexpr int NSTEPP = NCH * CH;
    // (the kernel's own chunk: finer where the pass holds more tiles than the LDS has room for at CH steps -- four tiles per
    //  pass: 8 steps at D = 48, 4 at D = 64 (154 KB in all); the image's layout does not depend on it, a tile's steps are
    //  contiguous)
    static constexpr int ch_of(int ntp) { return (D == 48 && ntp == 4) ? 8 : ((D == 64 && ntp == 4) ? 4 : CH); }
    // row stride of the LDS image of d (doubles): >= 64 and Q RS = 16 mod 32, so that the two lane groups of a
    // half-wavefront read 32 banks apart
    static constexpr int RS = D == 20 ? 80 : (D == 24 ? 72 : (D == 32 ? 66 : (D == 40 ? 72 : (D == 48 ? 68 : 65))));
    // component tiles that share a monomial product: what the LDS holds next to the image of 256 samples
    // four everywhere since round 5 (two wavefronts per SIMD with two tiles each); the A/B switches rebuild round 4's two
    // tiles on one wavefront at D = 48 / 64 (scripts/mgemm_d48_ab.py, scripts/mgemm_d64_four_ab.py)
#if defined(PMC_MG_D48_TWO_TILES)
    static constexpr int NCT_MAX = D <= 40 ? 4 : 2;
#elif defined(PMC_MG_D64_TWO_TILES)
    static constexpr int NCT_MAX = D <= 48 ? 4 : 2;
#else
    static constexpr int NCT_MAX = 4;
#endif
    static constexpr size_t lds_bytes(int nct)
    {
        return sizeof(double) * (size_t)(2 * nct * ch_of(nct) * 64 + 4 * D * RS + 2 * nct * 64);
    }
};

template <int IMM> __device__ __forceinline__ void mg_read64(double &v, unsigned addr)
{
    static_assert(IMM >= 0 && IMM < 65536 && IMM % 8 == 0, "ds_read_b64 offset");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
}
__device__ __forceinline__ void mg_wait5(double &a, double &b, double &c, double &d, double &e)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}
__device__ __forceinline__ void mg_wait1(double &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }

__device__ __forceinline__ double mg_xor16(double v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ double mg_xor32(double v) { return __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------------------------------------
// k_theta_build: the coefficient image, the epilogue's constants, the centre and the guard's norms from the pack.
// One workgroup per component (padding components up to a multiple of 16 NCT: zero coefficients, a value of -DBL_MAX).
//   img[(k / 16) * NSTEPP + s][16 g + k % 16] = coefficient of component k for the monomial of lane group g in step s
//   ctab[k][4] = c0 + log w_k, c1, 0, 0 (what the Student-t epilogue needs behind the product; the other kinds' constants
//                are folded into the image)
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_theta_build(const double *__restrict__ pack, int K, int kind,
                                                     double *__restrict__ img, double *__restrict__ ctab,
                                                     double *__restrict__ center, unsigned long long *__restrict__ guard)
{
    using C = MgCfg<D>;
    constexpr int STRIDE = pmc_pack_stride_c(D), T = pmc_tri(D), Q = C::Q, ND = C::ND, NQ = C::NQ;
    __shared__ double cen[D], dlt[D], Pd[D], Rm[D][D + 1], Pm[D][D + 1], red[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    // (bit 8 of `kind`: components without weight are allowed -- see `dead` below; bit 9: an emitting pass -- Student-t: the
    //  third constant is log((nu + D) / nu), what the epilogue's u' = rho' gamma needs, instead of log w, what `individual` needs)
    const bool allow_dead = (kind & 0x100) != 0, emitting = (kind & 0x200) != 0;
    kind &= 0xff;
    // the common centre: midrange of the component means per coordinate (every workgroup for itself, same bits: minimum
    // and maximum do not depend on the order).  Four groups of 64 threads take every fourth component, eight loads in
    // flight each: as one chain of K dependent L2 round trips per coordinate this was most of the kernel's 65 us.
    {
        const int j = tid & 63, part = tid >> 6;
        double lo = DBL_MAX, hi = -DBL_MAX;
        if (j < D) {
            int q = part;
            for (; q + 28 < K; q += 32) {
                double m[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) m[t] = pack[(size_t)(q + 4 * t) * STRIDE + j];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    lo = m[t] < lo ? m[t] : lo;
                    hi = m[t] > hi ? m[t] : hi;
                }
            }
            for (; q < K; q += 4) {
                const double m = pack[(size_t)q * STRIDE + j];
                lo = m < lo ? m : lo;
                hi = m > hi ? m : hi;
            }
        }
        red[tid] = lo;
        __syncthreads();
        if (tid < D) lo = fmin(fmin(red[tid], red[64 + tid]), fmin(red[128 + tid], red[192 + tid]));
        __syncthreads();
        red[tid] = hi;
        __syncthreads();
        if (tid < D) {
            hi = fmax(fmax(red[tid], red[64 + tid]), fmax(red[128 + tid], red[192 + tid]));
            const double c = 0.5 * lo + 0.5 * hi;
            cen[tid] = (c == c && fabs(c) <= DBL_MAX) ? c : 0.0;
            if (k == 0) center[tid] = cen[tid];
        }
        __syncthreads();
    }
    double *ik = img + ((size_t)(k / 16) * C::NSTEPP) * 64 + (k % 16);
    if (k >= K) {                                          // padding: no coefficients, a value no maximum ever takes
        for (int idx = tid; idx < C::NSTEPP * 4; idx += 256) ik[(size_t)(idx >> 2) * 64 + 16 * (idx & 3)] = 0.0;
        // (Gauss / VB: the value is the constant monomial, 4 x a quarter; Student-t: log 1 = 0 times c1 = 0, plus c0)
        __syncthreads();
        if (kind != PMC_KIND_STUDENT_T && tid < 4) ik[(size_t)(NQ + Q) * 64 + 16 * tid] = -0.25 * DBL_MAX;
        if (kind == PMC_KIND_STUDENT_T && tid < 4) ik[(size_t)(NQ + Q) * 64 + 16 * tid] = 0.25;
        if (tid < 4) ctab[(size_t)k * 4 + tid] = tid == 0 ? -DBL_MAX : 0.0;
        return;
    }
    const double *pk = pack + (size_t)k * STRIDE;
    for (int idx = tid; idx < D * D; idx += 256) {
        const int i = idx / D, j = idx - i * D;
        Rm[i][j] = j >= i ? pk[D + i * D - i * (i + 1) / 2 + j] : 0.0;
    }
    __syncthreads();
    if (tid < D) dlt[tid] = pk[tid] - cen[tid];
    // P = R^T R (symmetric; thread per element of the upper triangle)
    double fro = 0.0;
    for (int idx = tid; idx < D * D; idx += 256) {
        const int i = idx / D, j = idx - i * D;
        if (j < i) continue;
        double s = 0.0;
        for (int l = 0; l <= i; ++l) s = fma(Rm[l][i], Rm[l][j], s);
        Pm[i][j] = s;
        Pm[j][i] = s;
        fro += (i == j ? 1.0 : 2.0) * s * s;
    }
    __syncthreads();
    double pd2 = 0.0, cst = 0.0;
    if (tid < D) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s = fma(Pm[tid][j], dlt[j], s);
        Pd[tid] = s;
        pd2 = s * s;
        cst = s * dlt[tid];
    }
    __syncthreads();
    // block sums of fro, pd2, cst in a fixed order
    double sums[3];
    const double vals[3] = {fro, pd2, cst};
    for (int q = 0; q < 3; ++q) {
        red[tid] = vals[q];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        sums[q] = red[0];
        __syncthreads();
    }
    // What the product returns is the component's value itself where that is affine in the form (one instruction less
    // per pair, no constants in the epilogue):
    //   Gauss      a + log w = (c0 + log w) - maha / 2                          (gauss.pyx:151, the weight of logsumexp2D folded in)
    //   VB         a = c2 + (c3 - c0) / 2 - c1 maha / 2                         (variational.pyx:798, :691)
    //   Student-t  t = 1 + c2 maha, then a + log w = (c0 + log w) + c1 log t    (student_t.pyx:159-164)
    // i.e. every coefficient times `scale`, the constant monomial plus `shift`.
    const double *c = pk + D + T;
    const bool vb = kind == PMC_KIND_VB;
    // A component WITHOUT weight (a pruned component of a PMC run: pmc.pyx:109-117 sets its weight to 0 and leaves it in the
    // mixture) adds nothing to the sum but takes part in the reference's row maximum with its unweighted value
    // (logsumexp2D, _regularize.pyx:73-77).  Where the caller allows it (the passes that emit no u), its value comes out of
    // the product WITHOUT a log-weight, its column number is stored negated as the mark, and k_mgemm keeps its values out of
    // sum and maximum but tracks their maximum per sample: see the a-posteriori test at the end of k_mgemm.
    const bool dead = !vb && allow_dead && c[4] == 0.0;
    const double logw = (vb || dead) ? 0.0 : log(c[4]);
    const double scale = kind == PMC_KIND_GAUSS ? -0.5 : (vb ? -0.5 * c[1] : c[2]);
    const double shift = kind == PMC_KIND_GAUSS ? c[0] + logw : (vb ? c[2] + 0.5 * (c[3] - c[0]) : 1.0);
    for (int idx = tid; idx < C::NSTEPP * 4; idx += 256) {
        const int s = idx >> 2, g = idx & 3;
        double v = 0.0;
        if (s < NQ) {
            const int a = s / ND, d = s - a * ND;
            const int i = (a + g * Q) % D, j = (a + d + g * Q) % D;
            if (d == 0) v = scale * Pm[i][i];
            else if (d == 2 * Q) v = g < 2 ? scale * (2.0 * Pm[i][j]) : 0.0;   // (i, i + D / 2): the pairs of groups 2, 3 repeat 0, 1
            else v = scale * (2.0 * Pm[i][j]);
        } else if (s < NQ + Q) {
            v = scale * (-2.0 * Pd[(s - NQ + g * Q) % D]);
        } else if (s == NQ + Q) {
            v = 0.25 * (scale * sums[2] + shift);
        }
        ik[(size_t)s * 64 + 16 * g] = v;
    }
    if (tid == 0) {
        double *o = ctab + (size_t)k * 4;                  // (o[0], o[1]: the Student-t epilogue; o[2], o[3]: `individual`)
        o[0] = c[0] + logw;
        o[1] = c[1];
        o[2] = logw;                                       // the product returns a_nk + log w_k: `individual` takes it off again
        if (emitting && kind == PMC_KIND_STUDENT_T) o[2] = log(-2.0 * c[1] / c[3]);   // gamma = (nu + D) / (nu t): its constant
        o[3] = (double)((const long long *)c)[5];          // the component's output column (mixture.pyx:138: individual[:, k])
        if (dead) {
            o[3] = -(o[3] + 1.0);
            atomicAdd((int *)guard + 10, 1);               // (the head's count of dead components: k_mgemm's switch)
        }
        // |d a / d maha|
        // (Student-t: 1 -- the norms price the error of maha itself; the slope (nu + D) / (2 (nu + maha)) depends on the pair and
        //  is applied where maha is known, in k_mgemm's epilogue: see `viol` there)
        const double sk = kind == PMC_KIND_GAUSS ? 0.5 : (kind == PMC_KIND_STUDENT_T ? 1.0 : 0.5 * fabs(c[1]));
        double th[3] = {sk * sqrt(sums[0]), 2.0 * sk * sqrt(sums[1]), sk * fabs(sums[2])};
        // A weight that is negative or not finite -- or zero where the caller's epilogue has no place for dead components
        // (the emitting passes) -- keeps the mixture with the exact kernels: a NaN norm refuses every sample
        if (!vb && !dead && !(c[4] > 0.0 && c[4] <= DBL_MAX)) th[0] = __longlong_as_double(0x7ff8000000000000LL);
        for (int q = 0; q < 3; ++q) {
            // non-negative doubles order like their bit patterns; a NaN's pattern lies above every number's, so it wins
            // and the guard refuses every sample
            const double t = th[q] == th[q] ? th[q] : __longlong_as_double(0x7ff8000000000000LL);
            atomicMax(guard + q, (unsigned long long)__double_as_longlong(t));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_mgemm
// ---------------------------------------------------------------------------------------------
// NCT: component tiles per wavefront and pass.  HW = 1: four wavefronts, one per 64-sample tile, all tiles of a pass.
// HW = 2: EIGHT wavefronts, two per SIMD -- wavefront (tile, half) takes tiles 2 half, 2 half + 1 of the pass's four on its
// tile's samples: each forms its own monomial products (one v_mul_f64 per two matrix instructions instead of one per
// four), but the two wavefronts of a SIMD cover each other's LDS waits, multiplies and epilogues, which one wavefront
// per SIMD leaves exposed (matrix pipe 78 % busy at D = 40, K = 128).  The halves of a sample row meet in LDS at the end.
template <int D, int NCT, int HW>
__global__ __launch_bounds__(256 * HW) void k_mgemm(const PmcArgsQ q)
{
    constexpr int NTP = NCT * HW;                          // component tiles per pass
    using C = MgCfg<D>;
    constexpr int Q = C::Q, RS = C::RS, CH = C::ch_of(NTP), NCH = C::NSTEPP / CH, ND = C::ND, NQ = C::NQ, NSTEP = C::NSTEP;
    static_assert(NCH * CH == C::NSTEPP, "the kernel's chunk divides the image's padded step count");
    static_assert(RS >= 64 && (Q * RS) % 32 == 16, "bank spread of the rotated views");
    static_assert(NCH % 2 == 0, "the chunk's buffer is a compile-time constant of the step");
    extern __shared__ double lds[];
    double *th = lds;                                      // [2][NTP][CH * 64]
    double *s_price = th;                                  // [4][64] the guard's price per sample (Student-t epilogue): in the
                                                           // first theta buffer, read before the barrier in front of stage(0)
    double *dl = lds + 2 * NTP * CH * 64;                  // [4][D][RS]
    double *cts = dl + 4 * D * RS;                         // [2][NTP * 16][4]
    __shared__ int s_flag;
    const PmcArgsA &a = q.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, half = wave8 >> 2;         // sample tile of the workgroup, half of the pass's tiles
    const int s16 = lane & 15, g = lane >> 4;
    const long long tile = blockIdx.x * 4LL + wave;
    const long long n = tile * 64 + lane;
    const bool valid = n < a.N;
    const bool tile_live = tile * 64 < a.N;               // wave-uniform
    const int K = a.K, kind = q.kind;
    double *dw = dl + wave * D * RS;

    // ---- the wavefront's samples minus the centre -> LDS, and the guard
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (half == 0) {                                       // (wave-uniform: a tile's image is written once)
        const long long nc = valid ? n : a.N - 1;
        const double *xr = a.x + nc * (long long)a.dreal;
        double dsq = 0.0;
#pragma unroll 8
        for (int j = 0; j < D; ++j) {
#ifndef PMC_MG_AB_NOPRO
            const double v = j < a.dreal ? xr[j] - q.center[j] : 0.0;
#else
            const double v = 1e-3 * (double)(j + lane);
#endif
            dw[j * RS + lane] = v;
            dsq = fma(v, v, dsq);
        }
        const double dn = sqrt(dsq);
        const double e = fma(fma(q.guard[0], dn, q.guard[1]), dn, q.guard[2]);
        s_price[wave * 64 + lane] = e;
        // Gauss / VB: the price IS the error bound of a_nk (constant slope).  Student-t: it prices maha; the slope
        // |da / dmaha| = (nu + D) / (2 (nu + maha)) is at most (nu + D) / (2 nu) but a fraction of that for all but the rare
        // pair with maha << nu, so the pairs are tested one by one behind the product (epilogue) and only samples that are
        // hopeless at any conceivable slope (or not finite) are refused here
        const double lim = kind == PMC_KIND_STUDENT_T ? 4096.0 * q.eps_tol : q.eps_tol;
        if (__any(!(e <= lim)) && lane == 0) s_flag = 1;
    }
    __syncthreads();
    const int flagged = s_flag;
    if (tid == 0) {
        q.blockflag[blockIdx.x] = flagged;
        if (flagged) *q.redo = 1;                          // (benign race: every writer stores 1)
    }
    if (flagged) return;                                   // the exact kernel behind does this workgroup's samples

    // Student-t: the prices of the four samples this lane holds pairs of (accumulator layout: samples 16 t + s16), and the
    // verdict of the pair-by-pair test
    double Et[4] = {0.0, 0.0, 0.0, 0.0};
    bool viol = false;
    if (kind == PMC_KIND_STUDENT_T) {
#pragma unroll
        for (int t = 0; t < 4; ++t) Et[t] = s_price[wave * 64 + 16 * t + s16];
    }
    const unsigned dwa = (unsigned)(uintptr_t)(mg_lvoid_t *)dw;
    const unsigned tha = (unsigned)(uintptr_t)(mg_lvoid_t *)th + 8u * lane + 8u * (unsigned)(half * NCT * CH * 64);
    unsigned base[4];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) base[jq] = dwa + 8u * (unsigned)((((jq + g) * Q) % D) * RS + s16);

    const int npass = q.npass, nchunks = npass * NCH;
    auto stage = [&](int cg) {                             // chunk cg = (pass, chunk of the pass) -> theta buffer cg & 1
        const int pass = cg / NCH, ch = cg - pass * NCH;
#pragma unroll
        for (int c = 0; c < NTP; ++c) {
            const double *src = q.img + ((size_t)(pass * NTP + c) * NCH + ch) * CH * 64;
            double *dst = th + ((cg & 1) * NTP + c) * CH * 64;
#pragma unroll
            for (int p = 0; p < (CH * 64 / 128 + 4 * HW - 1) / (4 * HW); ++p) {
                const int piece = wave8 + 4 * HW * p;
                if (piece < CH * 64 / 128)
                    __builtin_amdgcn_global_load_lds((mg_gvoid_t *)(src + piece * 128 + 2 * lane),
                                                     (mg_lvoid_t *)(dst + piece * 128), 16, 0, 0);
            }
        }
    };
    auto stage_consts = [&](int pass) {                    // the pass's 16 NCT x 4 constants -> cts[pass & 1]
        constexpr int PIECES = (NTP * 64 + 127) / 128;
        if (wave8 < PIECES) {
            const double *src = q.ctab + (size_t)pass * NTP * 64 + wave8 * 128;
            int o = 2 * lane;
            if (NTP * 64 < 128 && o > NTP * 64 - 2) o = NTP * 64 - 2;
            __builtin_amdgcn_global_load_lds((mg_gvoid_t *)(src + o), (mg_lvoid_t *)(cts + (pass & 1) * NTP * 64 + wave8 * 128),
                                             16, 0, 0);
        }
    };

    // per-sample running state of the row (lane l = sample l of the tile): maximum, sum, VB bound term
    double Mrun = -DBL_MAX, srun = 0.0, tbrun = 0.0;
    // components without weight (k_theta_build): the running maximum of THEIR values per sample, for the test at the end
#ifdef PMC_MG_AB_PLAIN                                     // (A/B, timing only: without the dead-component and Student-t pair tests)
    constexpr int has_dead = 0;
#else
    const int has_dead = __builtin_amdgcn_readfirstlane(((const int *)q.guard)[10]);
#endif
    double Mdrun = -DBL_MAX;
    const ExpConst EC;
    const bool emit = a.u != nullptr;
    // columns of u: the components that get responsibilities -- all K, or the first a.ku when pruned components (no weight:
    // their values are -DBL_MAX by the time u' is formed, nothing of them is stored) stand at the end of the pack
    const int KU = (emit && a.ku > 0) ? a.ku : K;
    const int G = (KU + PMC_RESP_GROUP - 1) / PMC_RESP_GROUP;
    double *ut = emit ? a.u + (size_t)(tile_live ? tile : 0) * KU * 64 + s16 : nullptr;
    double *gs = emit ? a.gscale + (size_t)(tile_live ? tile : 0) * G * 64 + lane : nullptr;

    md4 acc[NCT][4];

    // ---- epilogue of one pass: a_nk from the forms, the pass's maximum and sums, u'
    const bool keep_forms = a.atile != nullptr && tile_live && kind != PMC_KIND_VB;
    double *mt = keep_forms ? a.atile + (size_t)tile * K * 64 + s16 : nullptr;
    auto epilogue = [&](int pass) {
        const double *ct = cts + (pass & 1) * NTP * 64 + half * NCT * 64 + 4 * g;
        auto kkeep = [&](int c, int r) { return (pass * NTP + half * NCT + c) * 16 + g + 4 * r; };
        auto kkeep_ok = [&](int c, int r) { return kkeep(c, r) < K; };
        double Mp[4] = {-DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
        // The product already is the component's value a_nk [+ log w_k] for the Gauss and VB kinds (k_theta_build folds
        // their constants into the image); Student-t: t = 1 + maha / nu came out, a = (c0 + log w) + c1 log t
        if (kind == PMC_KIND_STUDENT_T) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const md2 c01 = *(const md2 *)(ct + (c * 16 + 4 * r) * 4);
                    // the pair's error bound: price of maha x slope, slope = |c1| / (nu t) with t = 1 + maha / nu just computed
                    // and nu = -2 c1 - D (c1 = -(nu + D) / 2); tested as  |c1| / nu * price > tolerance * t  (1 % for the
                    // approximate reciprocal; padding components have c1 = 0)
                    const double nu_cr = -2.0 * c01[1] - (double)a.dreal;
                    const double cc = 1.01 * fabs(c01[1]) * __builtin_amdgcn_rcp(nu_cr);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#ifndef PMC_MG_AB_PLAIN
                        viol |= cc * Et[t] > q.eps_tol * acc[c][t][r];
#endif
                        // kept forms (pmc_*_keep): maha = nu (t - 1), tile-major, column = position in the pack
                        if (keep_forms && kkeep_ok(c, r)) mt[(size_t)kkeep(c, r) * 64 + 16 * t] = nu_cr * (acc[c][t][r] - 1.0);
                        double tt = log_pos(acc[c][t][r]);               // student_t.pyx:161-164
                        tt *= c01[1];
                        tt += c01[0];
                        acc[c][t][r] = tt;
                    }
                }
        }
        // kept forms of a Gauss mixture (pmc_*_keep -> pmc_estep_from_tiles): the product returned a + log w = (c0 + log w) -
        // maha / 2, so maha = 2 ((c0 + log w) - value): an absolute error of a few ulps of |c0|, 1e-14, where the exact kernel
        // stores the form itself -- "to rounding", as include/pmc_hip.h says of the large-batch forms
        if (keep_forms && kind == PMC_KIND_GAUSS) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kkeep_ok(c, r)) {
                        const double c0lw = ct[(c * 16 + 4 * r) * 4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) mt[(size_t)kkeep(c, r) * 64 + 16 * t] = 2.0 * (c0lw - acc[c][t][r]);
                    }
        }
        // `individual` (mixture.pyx:138-151: the N x K component log-densities, the reference's own intermediate): straight
        // from the accumulator layout -- lane (g, s) holds components g + 4 r of tile c for the samples 16 t + s, the four
        // lane groups write four neighbouring columns of a row
        if (a.individual != nullptr && kind != PMC_KIND_VB) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = (pass * NTP + half * NCT + c) * 16 + g + 4 * r;
                    if (kk < K) {
                        const md2 lc = *(const md2 *)(ct + (c * 16 + 4 * r) * 4 + 2);
                        const long long col = (long long)(lc[1] < 0.0 ? -lc[1] - 1.0 : lc[1]);   // (negated: no weight)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const long long nt = tile * 64 + 16 * t + s16;
                            if (nt < a.N) a.individual[nt * a.ld + col] = acc[c][t][r] - lc[0];
                        }
                    }
                }
        }
        if (has_dead) {
            // values of components without weight: into their own maximum, out of the pass's (a value no maximum takes and
            // whose exp is 0, as the padding components have)
            double Md[4] = {-DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dd = ct[(c * 16 + 4 * r) * 4 + 3] < 0.0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        Md[t] = max_f64(Md[t], dd ? acc[c][t][r] : -DBL_MAX);
                        acc[c][t][r] = dd ? -DBL_MAX : acc[c][t][r];
                    }
                }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Md[t] = max_f64(Md[t], mg_xor16(Md[t]));
                Md[t] = max_f64(Md[t], mg_xor32(Md[t]));
            }
            Mdrun = max_f64(Mdrun, g == 0 ? Md[0] : (g == 1 ? Md[1] : (g == 2 ? Md[2] : Md[3])));
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) Mp[t] = max_f64(acc[c][t][r], Mp[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            Mp[t] = max_f64(Mp[t], mg_xor16(Mp[t]));
            Mp[t] = max_f64(Mp[t], mg_xor32(Mp[t]));
        }
        double sp[4] = {0.0, 0.0, 0.0, 0.0}, tb[4] = {0.0, 0.0, 0.0, 0.0};
        // MODE 0: Gauss (and every pass that emits nothing), 1: VB, 2: the emitting pass of a Student-t mixture -- u = w rho gamma
        // (pmc.pyx:602-610) with gamma = (nu + D) / (nu + maha) = ((nu + D) / nu) / t: the value a = c0' + c1 log t is in hand,
        // so 1 / t = exp(-log t) = exp((c0' - a) / c1) and u' = exp(a - M) exp((c0' - a) / c1 + log((nu + D) / nu)): a second
        // exponential per pair instead of sixteen more live registers for t or a division per pair (the subtraction loses
        // |a| eps / |c1| ~ 1e-15 of log t).  The sums of the degree-of-freedom condition need the row's factor and follow in
        // a kernel of their own (k_dof_sums, pmc_tiles.hip).
        auto exps = [&](auto MODE_) {
            constexpr int MODE = decltype(MODE_)::value;
            constexpr bool VB = MODE == 1;
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = (pass * NTP + half * NCT + c) * 16 + g + 4 * r;
                    const bool st = emit && tile_live && kk < KU;
                    double c0s = 0.0, ic1 = 0.0, lcg = 0.0;
                    if constexpr (MODE == 2) {
                        const md2 c01 = *(const md2 *)(ct + (c * 16 + 4 * r) * 4);
                        c0s = c01[0];
                        lcg = ct[(c * 16 + 4 * r) * 4 + 2];
                        ic1 = __builtin_amdgcn_rcp(c01[1]);              // (padding components: c1 = 0, nothing of theirs is stored)
                        ic1 = fma(fma(-c01[1], ic1, 1.0), ic1, ic1);
                        ic1 = fma(fma(-c01[1], ic1, 1.0), ic1, ic1);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double lr = max_f64(acc[c][t][r] - Mp[t], -1075.0);   // variational.pyx:741 / _regularize.pyx:79
                        const double e = exp_clamped(lr, EC);
                        double uo;
                        if constexpr (VB) {
                            tb[t] = fma(e, lr, tb[t]);
                            sp[t] += e;
                            uo = zero_to_tiny(e);                        // variational.pyx:751-753
                        } else if constexpr (MODE == 2) {
                            sp[t] += e;
                            uo = e * exp_clamped(max_f64(fma(c0s - acc[c][t][r], ic1, lcg), -1075.0), EC);
                        } else {
                            uo = e;                                      // w_k exp(a - M): _regularize.pyx:79 / pmc.pyx:39
                            sp[t] += e;
                        }
                        // (non-temporal: 8 K bytes per sample stream out while every workgroup re-reads the coefficient
                        //  image out of L2 -- k_mgemm -1.2 % at D = 40, K = 128; no effect on k_resp_groups at D = 20)
                        if (st) __builtin_nontemporal_store(uo, ut + (size_t)kk * 64 + 16 * t);
                    }
                }
        };
        if (kind == PMC_KIND_VB) exps(ic<1>{});
        else if (kind == PMC_KIND_STUDENT_T && emit) exps(ic<2>{});
        else exps(ic<0>{});
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sp[t] += mg_xor16(sp[t]);
            sp[t] += mg_xor32(sp[t]);
        }
        if (kind == PMC_KIND_VB) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                tb[t] += mg_xor16(tb[t]);
                tb[t] += mg_xor32(tb[t]);
            }
        }
        // lane l = sample l: the pass joins the row's running maximum / sum / bound term
        const double Mo = g == 0 ? Mp[0] : (g == 1 ? Mp[1] : (g == 2 ? Mp[2] : Mp[3]));
        const double so = g == 0 ? sp[0] : (g == 1 ? sp[1] : (g == 2 ? sp[2] : sp[3]));
        const double to = g == 0 ? tb[0] : (g == 1 ? tb[1] : (g == 2 ? tb[2] : tb[3]));
        const double Mn = max_f64(Mo, Mrun);
        const double ar = max_f64(Mrun - Mn, -1075.0), ag = max_f64(Mo - Mn, -1075.0);
        const double cr = exp_clamped(ar, EC), cg = exp_clamped(ag, EC);
        if (kind == PMC_KIND_VB) tbrun = cr * fma(ar, srun, tbrun) + cg * fma(ag, so, to);
        srun = cr * srun + cg * so;
        Mrun = Mn;
        if (emit && tile_live) {
            // the pass's maximum waits in the factor's place (every group of 16 of the pass has the same)
#pragma unroll
            for (int c = 0; c < NCT; ++c)
                if (pass * NTP + half * NCT + c < G) gs[(size_t)(pass * NTP + half * NCT + c) * 64] = Mo;
        }
    };

    // ---- the passes
    __syncthreads();
    stage(0);
    for (int kt = 0; kt <= npass; ++kt) {
        if (kt < npass) {
            // chunk (kt, 0) has landed, nobody reads the other buffer any more
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt * NCH + 1 < nchunks) stage(kt * NCH + 1);
            stage_consts(kt);
        }
#ifndef PMC_MG_AB_NOEPI                                    // (A/B switches, timing only: scripts/mgemm_ab.sh)
        if (kt > 0) epilogue(kt - 1);                      // (behind the barrier: its stores are old when the next one waits)
#else
        if (kt > 0) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) srun += acc[c][t][r];
        }
#endif
        if (kt == npass) break;
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[c][t] = md4{0.0, 0.0, 0.0, 0.0};
        double ar[2][4], bv[2][4], tv[NCT][3], z[2][4];
        // three stages, one step apart: fetch (LDS reads) -> mul (the monomials) -> the matrix instructions
        auto fetch = [&](auto S_) {
            constexpr int s = decltype(S_)::value, set = s & 1;
            constexpr int ch = s / CH, i = s % CH;
#ifdef PMC_MG_AB_NOTHREAD
            if constexpr (s < 3)
#endif
            static_for<0, NCT>([&](auto C_) {
                constexpr int c = decltype(C_)::value;
                constexpr int TH = 8 * ((((ch & 1) * NTP + c) * CH + i) * 64);
                mg_read64<TH>(tv[c][s % 3], tha);
            });
#ifdef PMC_MG_AB_NODREAD
            constexpr bool dread = s < 2;
#else
            constexpr bool dread = true;
#endif
            if constexpr (!dread) {
            } else if constexpr (s < NQ) {
                constexpr int aa = s / ND, dlt = s % ND, b = aa + dlt, jq = b / Q, br = b - jq * Q;
                static_for<0, 4>([&](auto T_) {
                    constexpr int t = decltype(T_)::value;
                    if constexpr (dlt == 0) mg_read64<8 * (aa * RS + 16 * t)>(ar[aa & 1][t], base[0]);
                    else mg_read64<8 * (br * RS + 16 * t)>(bv[set][t], base[jq]);
                });
            } else if constexpr (s < NQ + Q) {
                constexpr int r = s - NQ;
                static_for<0, 4>([&](auto T_) {
                    constexpr int t = decltype(T_)::value;
                    mg_read64<8 * (r * RS + 16 * t)>(bv[set][t], base[0]);
                });
            }
        };
        auto arrive = [&](auto S_) {
            constexpr int s = decltype(S_)::value, set = s & 1;
            if constexpr (s < NQ && s % ND == 0) {
                constexpr int aa = s / ND;
                mg_wait5(tv[0][s % 3], ar[aa & 1][0], ar[aa & 1][1], ar[aa & 1][2], ar[aa & 1][3]);
            } else if constexpr (s < NQ + Q) mg_wait5(tv[0][s % 3], bv[set][0], bv[set][1], bv[set][2], bv[set][3]);
            else mg_wait1(tv[0][s % 3]);
            static_for<1, NCT>([&](auto C_) { mg_wait1(tv[decltype(C_)::value][s % 3]); });   // (already there: ties the registers)
        };
        auto mul = [&](auto S_, auto T_) {
            constexpr int s = decltype(S_)::value, set = s & 1, t = decltype(T_)::value;
            if constexpr (s < NQ) {
                constexpr int aa = s / ND, dlt = s % ND;
#ifdef PMC_MG_AB_NOMUL
                z[set][t] = dlt == 0 ? ar[aa & 1][t] : bv[set][t];
#else
                z[set][t] = dlt == 0 ? ar[aa & 1][t] * ar[aa & 1][t] : ar[aa & 1][t] * bv[set][t];
#endif
            } else if constexpr (s < NQ + Q) z[set][t] = bv[set][t];
            else z[set][t] = 1.0;
        };
        auto boundary = [&](auto S_) {                     // in front of the first fetch of a chunk (not the pass's first)
            constexpr int s = decltype(S_)::value;
#ifdef PMC_MG_AB_NOBAR
            if constexpr (false) {
#else
            if constexpr (s % CH == 0 && s > 0) {
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int cg = kt * NCH + s / CH;
                if (cg + 1 < nchunks) stage(cg + 1);
            }
        };
        fetch(ic<0>{});
        arrive(ic<0>{});
        static_for<0, 4>([&](auto T_) { mul(ic<0>{}, T_); });
        if constexpr (NSTEP > 1) {
            boundary(ic<1>{});
            fetch(ic<1>{});
        }
        static_for<0, NSTEP>([&](auto S_) {
            constexpr int s = decltype(S_)::value;
            if constexpr (s + 1 < NSTEP) arrive(ic<s + 1>{});
            if constexpr (s + 2 < NSTEP) {
                boundary(ic<s + 2>{});
                fetch(ic<s + 2>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            // The next step's four products in ONE block in front of this step's matrix instructions, not one behind every
            // pair of them: v_mul_f64 runs on the units the fp64 matrix instructions run on, and every change between
            // the two streams costs more than the multiply itself (D = 40, K = 128: 28.5 -> 27.1 ps per pair; behind the
            // block of matrix instructions 27.5; without any products 26.0 -- scripts/mgemm_ab.sh).
#if !defined(PMC_MG_MUL_INTERLEAVED) && !defined(PMC_MG_MUL_AFTER)
            if constexpr (s + 1 < NSTEP) static_for<0, 4>([&](auto T_) { mul(ic<s + 1>{}, T_); });
            __builtin_amdgcn_sched_barrier(0);
#endif
            static_for<0, 4>([&](auto T_) {
                constexpr int t = decltype(T_)::value;
                static_for<0, NCT>([&](auto C_) {
                    constexpr int c = decltype(C_)::value;
                    acc[c][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[c][s % 3], z[s & 1][t], acc[c][t], 0, 0, 0);
                });
#if defined(PMC_MG_MUL_INTERLEAVED)
                if constexpr (s + 1 < NSTEP) mul(ic<s + 1>{}, T_);
#endif
            });
#if defined(PMC_MG_MUL_AFTER)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s + 1 < NSTEP) static_for<0, 4>([&](auto T_) { mul(ic<s + 1>{}, T_); });
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        // chunks of padding steps behind the last real one (D = 64: the chunk count is rounded up to an even number): no
        // step reads them, but the boundary in front of each is what stages the chunk behind it -- the last one the NEXT
        // pass's first chunk
        static_for<(NSTEP + CH - 1) / CH, NCH>([&](auto C_) { boundary(ic<decltype(C_)::value * CH>{}); });
    }

    // ---- HW = 2: the two halves of a sample row meet -- half 1 hands its running maximum / sum / bound term to half 0
    // through LDS (the theta buffers: behind the barrier nobody reads them any more)
    double *xch = th;                                      // [4 tiles][64][4]
    if constexpr (HW == 2) {
        __syncthreads();
        double *mine = xch + (size_t)(wave * 64 + lane) * 4;
        if (half == 1) {
            mine[0] = Mrun;
            mine[1] = srun;
            mine[2] = tbrun;
            mine[3] = Mdrun;
        }
        __syncthreads();
        if (half == 0) {
            const double Mo = mine[0], so = mine[1], to = mine[2];
            Mdrun = max_f64(Mdrun, mine[3]);
            const double Mn = max_f64(Mo, Mrun);
            const double ar = max_f64(Mrun - Mn, -1075.0), ag = max_f64(Mo - Mn, -1075.0);
            const double cr = exp_clamped(ar, EC), cg = exp_clamped(ag, EC);
            if (kind == PMC_KIND_VB) tbrun = cr * fma(ar, srun, tbrun) + cg * fma(ag, so, to);
            srun = cr * srun + cg * so;
            Mrun = Mn;
        }
    }

    // ---- components without weight, a posteriori: the reference takes its row maximum over ALL components' unweighted values
    // (logsumexp2D, _regularize.pyx:73-77), the sum here is relative to the maximum of the weighted values of those that
    // have a weight -- the same number unless a dead component's value lies so far above every live one that the
    // reference's terms exp(a_k - max) leave the normal range (it then returns a degraded sum, or log 0 = -inf).  The
    // weighted live maximum is a lower bound of the unweighted one (log w <= 0): within 700 of the dead maximum every
    // live term of the reference is a normal number and the two agree to rounding; a sample beyond that sends its
    // workgroup to the exact kernel behind, which does the reference's arithmetic (its outputs overwrite these).
    if (has_dead && half == 0) {
        const bool far = valid && !(Mdrun - Mrun <= 700.0);
        if (__any(far) && lane == 0) {
            q.blockflag[blockIdx.x] = 1;
            *q.redo = 1;
        }
    }
    // ---- Student-t, a posteriori: a pair whose error bound exceeds the tolerance (every wavefront for the pairs it held)
    if (kind == PMC_KIND_STUDENT_T && __any(viol) && lane == 0) {
        q.blockflag[blockIdx.x] = 1;
        *q.redo = 1;
    }

    // ---- per sample (lane l = sample l, half 0): the row's log-sum-exp / normalisation, outputs, the factor's row part
    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double f = 0.0;                                        // VB: w_n / s (times exp(M_g - M) per group); else w_n / (exp(lse) + tiny)
    if (half == 0) {
        if (kind == PMC_KIND_VB) {
            // variational.pyx:748-755, :1003-1013
            const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;
            const double swv = valid ? sw : 0.0;
            const double norm_inv = 1. / srun;
            sc[0] = swv * fma(tbrun, norm_inv, log_any(norm_inv));
            f = swv * norm_inv;
        } else {
            const double lse = log_any(srun) + Mrun;      // _regularize.pyx:81
            if (a.out != nullptr && valid) a.out[n] = lse;
            double wn = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;
            if (a.log_target != nullptr && valid) {
                const double tmp = a.log_target[n] - lse; // importance_sampling.py:204
                const double w = exp(tmp);                // :207
                a.weights[n] = w;
                sc[0] = w;
                sc[1] = (w != 0.0) ? w * tmp : 0.0;       // convergence.py:35-36 (zeros masked)
                sc[2] = w * w;
                sc[4] = (isinf(w) && !isinf(tmp)) ? 1.0 : 0.0;
                if (emit) wn = w;
            }
            if (valid) sc[3] = (a.sample_w != nullptr) ? a.sample_w[n] * lse : lse;   // pmc.pyx:388-391
            // pmc.pyx:36-41: rho = exp(log q_k) w_k / (exp(lse) + tiny), times the sample's weight
            if (emit) f = valid ? wn / (exp(lse) + TINY) : 0.0;
        }
    }
    if (emit) {
        if constexpr (HW == 2) {                           // (the row's part of the factors travels back to half 1)
            double *mine = xch + (size_t)(wave * 64 + lane) * 4;
            __syncthreads();
            if (half == 0) {
                mine[0] = f;
                mine[1] = Mrun;
            }
            __syncthreads();
            f = mine[0];
            Mrun = mine[1];
        }
        if (tile_live) {
            // every wavefront completes the factors of the groups it parked its maxima for
            for (int gg = 0; gg < G; ++gg) {
                if (HW == 2 && ((gg % NTP) / NCT) != half) continue;
                if (kind == PMC_KIND_VB) gs[(size_t)gg * 64] = f * exp_clamped(max_f64(gs[(size_t)gg * 64] - Mrun, -1075.0), EC);
                else gs[(size_t)gg * 64] = f * exp(gs[(size_t)gg * 64]);
            }
        }
    }
    if (a.partials != nullptr) {
        if constexpr (HW == 1) {
            block_scalars<5>(sc, a.partials);
        } else {
            __shared__ double red2[4][PMC_NSCALARS];       // (block_scalars' tree over the four half-0 wavefronts)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const double v = wave_sum(sc[i]);
                if (half == 0 && lane == 0) red2[wave][i] = v;
            }
            __syncthreads();
            if (threadIdx.x < PMC_NSCALARS) {
                double v = 0.0;
                if (threadIdx.x < 5) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red2[w][threadIdx.x];
                }
                a.partials[(size_t)blockIdx.x * PMC_NSCALARS + threadIdx.x] = v;
            }
        }
    }
}

}  // namespace

// what the dispatcher needs: steps per component tile in the image (0: this dimension has no such kernel), the
// largest number of component tiles per pass
extern "C" void PMC_UNIT_NAME_X(pmc_mgemm_config_d, PMC_D, PMC_PADDED)(int *nstepp, int *nct_max)
{
    using C = MgCfg<D_>;
    *nstepp = C::ENABLED ? C::NSTEPP : 0;
    *nct_max = C::ENABLED ? C::NCT_MAX : 0;
}

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_theta_d, PMC_D, PMC_PADDED)(const double *pack, int K, int Kpad, int kind,
                                                                            double *img, double *ctab, double *center,
                                                                            unsigned long long *guard, hipStream_t st)
{
    if constexpr (!MgCfg<D_>::ENABLED) return hipErrorNotSupported;
    else {
        hipLaunchKernelGGL((k_theta_build<D_>), dim3((unsigned)Kpad), dim3(256), 0, st, pack, K, kind, img, ctab, center, guard);
        return hipGetLastError();
    }
}

template <int NCT, int HW> static hipError_t mgemm_launch(const PmcArgsQ &q, unsigned grid, hipStream_t st)
{
    using C = MgCfg<D_>;
    constexpr size_t lds = C::lds_bytes(NCT * HW);
    static_assert(lds + 1024 <= 160 * 1024, "k_mgemm: the sample image and the theta buffers exceed the LDS");
    const hipError_t once = PMC_SET_LDS_PER_DEVICE((&k_mgemm<D_, NCT, HW>), lds);
    if (once != hipSuccess) return once;
    hipLaunchKernelGGL((k_mgemm<D_, NCT, HW>), dim3(grid), dim3(256 * HW), lds, st, q);
    return hipGetLastError();
}

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_mgemm_d, PMC_D, PMC_PADDED)(int nct, const PmcArgsQ &q, unsigned grid,
                                                                            hipStream_t st)
{
    if constexpr (!MgCfg<D_>::ENABLED) return hipErrorNotSupported;
    else {
        // nct = component tiles per PASS: 2 (one wavefront per SIMD, both tiles) or 4 -- as two wavefronts per SIMD with
        // two tiles each (PMC_MGEMM_ONE_WAVE: one wavefront with all four, the A/B alternative)
        if (nct == 2) return mgemm_launch<2, 1>(q, grid, st);
        if constexpr (MgCfg<D_>::NCT_MAX >= 4) {
#ifdef PMC_MGEMM_ONE_WAVE
            if (nct == 4) return mgemm_launch<4, 1>(q, grid, st);
#else
            if (nct == 4) return mgemm_launch<2, 2>(q, grid, st);
#endif
        }
        return hipErrorInvalidValue;
    }
}
