// pmc_fused.hip -- the E-step in ONE kernel for small sample dimensions: responsibilities and
// sufficient statistics of a block of samples without the N x K responsibility matrix ever leaving
// the compute unit (the separate kernels write it tile-major to HBM and read it back: 16 K bytes per
// sample against 8 D for the samples themselves -- at D <= 8 that round trip, not the arithmetic, is
// what the E-step costs).  Compiled once per sample dimension D <= PMC_FUSED_MAX_DIM:
//     hipcc -DPMC_D=<D> -DPMC_PADDED=<0|1> -c pmc_fused.hip
//
// Replaces, for the VB E-step, _update_expectation_gauss_exponent / _update_log_rho / _update_r /
// _update_N_comp / _update_x_mean_comp / _update_S / _update_expectation_log_q_Z
// (pypmc/mix_adapt/variational.pyx:675-932, :1003-1013) and, for the Gaussian PMC update,
// calculate_rho_rb + the einsum reductions (pypmc/mix_adapt/pmc.pyx:23-43, :188-222).
//
// Two forms.  D <= PMC_F_REG_MAX_DIM: k_estep_reg, everything in registers (described at the kernel).
// Above it, k_estep_fused:
// One workgroup = 8 wavefronts, persistent over its share of the samples, "rounds" of TPR = 8 / QS
// tiles of 64 samples:
//   phase A (lane = sample, parameters as SGPR operands exactly like k_logpdf): a tile's K components
//     are split among QS wavefronts (<= 8 components each, a_nk parked in registers).  The soft-max
//     is the reference's own two-pass form -- row maximum first, then ONE exp per pair -- with the
//     maxima and partial sums of the QS wavefronts exchanged through LDS.  u_nk = w_n r_nk (VB) or
//     w_n rho_nk (PMC) goes to an LDS buffer, tile-major like the HBM buffer it replaces.
//   phase B (lane = (sample, coordinate) of a 16-sample sub-step): wavefront w owns components
//     w, w + 8, ... of ALL tiles of the round and accumulates 4 x 4 blocks of sum u d d^T on the fp64
//     matrix pipe (v_mfma_f64_4x4x4_4b_f64, lane layout of pmc_stats.hip).  If D is not a multiple
//     of 4 the vector is augmented, d~ = (d, 1): sum u d and sum u then come out of the same blocks.
// The accumulators stay in registers over all rounds; partial sums per workgroup are written in the
// layout of k_stats' partials and summed by the same fixed-order finishing kernel.
#include "pmc_device.h"

#if PMC_D <= PMC_FUSED_MAX_DIM

namespace {

constexpr int FW = PMC_F_WAVES;          // wavefronts per workgroup

template <int D> struct FusedGeom {
    static constexpr int G = (D + 3) / 4;                         // coordinate groups of 4
    static constexpr bool AUG = (D % 4) != 0;                     // room for the constant 1 in the last group
    static constexpr int PIT = ((4 * G + 3) / 8) * 8 + 4;         // doubles per LDS sample row, = 4 mod 8:
    static constexpr int NBLK = G * (G + 1) / 2;                  //   8 rows x 4 coordinates hit 32 bank pairs
};

// LDS doubles of a launch
__host__ __device__ constexpr int fused_lds_doubles(int D, int QS, int K)
{
    const int G = (D + 3) / 4, PIT = ((4 * G + 3) / 8) * 8 + 4, TPR = FW / QS;
    return TPR * 64 * PIT + TPR * K * 64 + 64 + (QS > 1 ? 3 * FW * 64 : 0);
}

// wavefronts per SIMD the registers must leave room for: three workgroups per compute unit while the
// Mahalanobis form is short (measured: D = 2 0.50 -> 0.47 ms per 4e6 samples x 32 components; from D = 5
// on the allocator spills at 80 registers and 4 is faster)
__host__ __device__ constexpr int fused_min_waves(int D)
{
#ifdef PMC_F_MIN_WAVES
    return PMC_F_MIN_WAVES;
#else
    return D <= 4 ? 6 : 4;
#endif
}

template <int D, bool PADDED, int KIND, int NCH>
__global__ __launch_bounds__(FW * 64, fused_min_waves(D)) void k_estep_fused(const PmcArgsF a)
{
    using GEO = FusedGeom<D>;
    constexpr int G = GEO::G, PIT = GEO::PIT, NBLK = GEO::NBLK, T = pmc_tri(D);
    constexpr bool AUG = GEO::AUG;
    constexpr int STRIDE = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    extern __shared__ double lds[];
    const int K = a.K;
    const int QS = a.qs, TPR = FW / QS;                            // wavefronts per tile, tiles per round
    double *xi = lds;                                              // [TPR][64][PIT] sample rows
    double *ub = xi + TPR * 64 * PIT;                              // [TPR][K][64]     a_nk, then u_nk, tile-major
    double *zero = ub + (size_t)TPR * K * 64;                      // [64]             u of a component slot beyond K
    double *red = zero + 64;                                       // [3][FW][64]      soft-max exchange
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // phase A role: tile slot ta, component range [k0, k1)
    const int ta = w / QS, q = w % QS;
    const int k0 = q * a.kq, k1 = (k0 + a.kq < K) ? k0 + a.kq : K;
    // phase B role: components cb + CW j (j < NCH) of the tile slots ts0, ts0 + TS, ...
    const int CW = a.cw, TS = FW / CW;
    const int cb = w % CW, ts0 = w / CW;
    const int ci = lane & 3, blk = (lane >> 2) & 3, ks = lane >> 4;
    const int srow = 2 * blk + (ks & 1) + 8 * (ks >> 1);           // sample of the 16-sample sub-step

    double acc2[NCH][NBLK], acc1[NCH][G], acc0[NCH], mu[NCH][G];
    int coff[NCH], toff[NCH];                                      // component's offset in a tile of ub, tile stride
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = cb + CW * j;
        // (the moments are taken about the pack's means unless the caller names other points)
        const double *pk = (a.shift_pack ? a.shift_pack : a.pack) + (size_t)(c < K ? c : 0) * STRIDE;
        coff[j] = c < K ? c * 64 : TPR * K * 64 - srow;            // beyond K: the zero slot, for every tile
        toff[j] = c < K ? K * 64 : 0;
        acc0[j] = 0.0;
#pragma unroll
        for (int I = 0; I < G; ++I) {
            acc1[j][I] = 0.0;
            mu[j][I] = (4 * I + ci < D) ? pk[4 * I + ci < D ? 4 * I + ci : 0] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc2[j][b] = 0.0;
    }
    if (threadIdx.x < 64) zero[threadIdx.x] = 0.0;                 // (the first barrier of a round orders it)
    double sc_a = 0.0;                                             // VB: E[log q(Z)] part; PMC: sum w log q
    // Student-t: the two sums of the degree-of-freedom condition (pmc.pyx:612, :659-679) of the wavefront's
    // phase-A components, lane j keeping those of component k0 + j; log(nu / 2) of that component likewise
    double vs1 = 0.0, vs2 = 0.0, lognu2 = 0.0;
    if constexpr (KIND == PMC_KIND_STUDENT_T) {
        const int kk = k0 + lane < k1 ? k0 + lane : (k1 > k0 ? k1 - 1 : 0);
        lognu2 = log(.5 * a.pack[(size_t)kk * STRIDE + D + T + 3]);
    }

    const long long nrounds = (a.ntiles + TPR - 1) / TPR;
    const long long r0 = (long long)blockIdx.x * a.rounds_per_wg;
    long long r1 = r0 + a.rounds_per_wg;
    if (r1 > nrounds) r1 = nrounds;
    for (long long round = r0; round < r1; ++round) {
        // ------------------------------------------------------------------ phase A
        const long long tile = round * TPR + ta;
        const long long n = tile * 64 + lane;
        const bool valid = n < a.N;
        double *ut = ub + (size_t)ta * K * 64 + lane;
        double M = a.max_init_zero ? 0.0 : -DBL_MAX;               // _regularize.pyx:73 / pmc.pyx:24-34
        {
            double xv[D];
            load_row<D, PADDED>(a.x, n, a.N, a.dreal, xv);
            if (q == 0) {                                          // wave-uniform
                double *row = xi + (size_t)(ta * 64 + lane) * PIT;
#pragma unroll
                for (int j = 0; j < 4 * G; ++j) row[j] = j < D ? xv[j] : ((AUG && j == D) ? 1.0 : 0.0);
            }
            cdouble *pk = (cdouble *)a.pack + (size_t)k0 * STRIDE;
#pragma unroll PMC_F_UNROLL_A
            for (int k = k0; k < k1; ++k, pk += STRIDE) {          // a_nk parked in the u buffer
                const double maha = mahalanobis<D>(xv, pk);
                double expo;
                const double v = component_value<D, KIND>(maha, pk + D + T, expo);
                ut[(size_t)k * 64] = v;
                if (v > M) M = v;
            }
        }
        if (QS > 1) {
            red[(size_t)w * 64 + lane] = M;
            __syncthreads();
            M = a.max_init_zero ? 0.0 : -DBL_MAX;
            for (int qq = 0; qq < QS; ++qq) {
                const double v = red[(size_t)(ta * QS + qq) * 64 + lane];
                if (v > M) M = v;
            }
        }
        double s = 0.0, tb = 0.0;
        {
            cdouble *pk = (cdouble *)a.pack + (size_t)k0 * STRIDE;
#pragma unroll PMC_F_UNROLL_A
            for (int k = k0; k < k1; ++k, pk += STRIDE) {
                const double lr = ut[(size_t)k * 64] - M;          // variational.pyx:741
                const double e = exp(lr);                          // :742
                if constexpr (KIND == PMC_KIND_VB) {
                    s += e;                                        // :743
                    tb = fma(e, lr, tb);                           // sum_k e_k (a_k - M), for E[log q(Z)]
                    ut[(size_t)k * 64] = e;
                } else {
                    s += pk[D + T + 4] * e;                        // _regularize.pyx:79
                    // (Student-t: a_nk stays parked -- the last pass recovers the Mahalanobis form from it)
                    if constexpr (KIND != PMC_KIND_STUDENT_T) ut[(size_t)k * 64] = e;
                }
            }
        }
        if (QS > 1) {
            red[(size_t)(FW + w) * 64 + lane] = s;
            if constexpr (KIND == PMC_KIND_VB) red[(size_t)(2 * FW + w) * 64 + lane] = tb;
            __syncthreads();
            s = 0.0;
            tb = 0.0;
            for (int qq = 0; qq < QS; ++qq) {
                s += red[(size_t)(FW + ta * QS + qq) * 64 + lane];
                if constexpr (KIND == PMC_KIND_VB) tb += red[(size_t)(2 * FW + ta * QS + qq) * 64 + lane];
            }
        }
        const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;
        const double swv = valid ? sw : 0.0;
        if constexpr (KIND == PMC_KIND_VB) {
            const double norm_inv = 1. / s;                        // variational.pyx:748-755
            for (int k = k0; k < k1; ++k) {
                double r = ut[(size_t)k * 64] * norm_inv;
                if (r == 0.0) r = TINY;
                ut[(size_t)k * 64] = swv * r;
            }
            // sum_k r_k (a_k - M + log norm_inv) with sum_k r_k = 1   (variational.pyx:1003-1013)
            if (q == 0) sc_a += swv * fma(tb, norm_inv, log_any(norm_inv));
        } else {
            const double lse = log_any(s) + M;                     // _regularize.pyx:81
            const double denom = exp(lse) + TINY;                  // pmc.pyx:41
            const double em = exp(M), inv_denom = 1. / denom;
            cdouble *pk = (cdouble *)a.pack + (size_t)k0 * STRIDE;
            for (int k = k0; k < k1; ++k, pk += STRIDE) {
                // exp(log q_k) = e exp(M) first: it underflows where the reference's does (pmc.pyx:39)
                if constexpr (KIND == PMC_KIND_STUDENT_T) {
                    // a = c0 + c1 log(1 + maha / nu) (student_t.pyx:159-164)  =>  L = log(1 + maha / nu) = (a - c0) / c1,
                    // nu + maha = nu exp(L): gamma = (nu + D) / (nu + maha) (pmc.pyx:610) and
                    // log((maha + nu) / 2) = log(nu / 2) + L (pmc.pyx:669) without keeping maha next to a.
                    // L is exact to eps |a| / |c1| in absolute terms, which is its weight in both uses.
                    const double v = ut[(size_t)k * 64];
                    const double e = exp(v - M);
                    const double rho = ((e * em) * pk[D + T + 4]) * inv_denom;
                    const double wr = swv * rho;
                    const double nu = pk[D + T + 3];
                    const double L = (v - pk[D + T]) / pk[D + T + 1];
                    const double gamma = (nu + (double)D) / (nu * exp(L));
                    ut[(size_t)k * 64] = wr * gamma;
                    const double s1 = wave_sum(wr);
                    const double s2 = wave_sum(wr * (__shfl(lognu2, k - k0, 64) + L));
                    if (lane == k - k0) {
                        vs1 += s1;
                        vs2 += s2;
                    }
                } else {
                    const double rho = ((ut[(size_t)k * 64] * em) * pk[D + T + 4]) * inv_denom;
                    ut[(size_t)k * 64] = swv * rho;
                }
            }
            if (q == 0) sc_a += swv * lse;                         // pmc.pyx:388-391
        }
        __syncthreads();

        // ------------------------------------------------------------------ phase B
        {
            for (int t = ts0; t < TPR; t += TS) {
                const double *xt = xi + (size_t)(t * 64 + srow) * PIT + ci;
                const double *utile = ub + srow;
                // operands of sub-step ss + 1 are read before the arithmetic of sub-step ss
                double xb[2][G], uq[2][NCH];                           // ping-pong operand registers
                auto fetch = [&](int ss, double (&xx)[G], double (&uu)[NCH]) {
    #pragma unroll
                    for (int I = 0; I < G; ++I) xx[I] = xt[ss * 16 * PIT + 4 * I];
    #pragma unroll
                    for (int j = 0; j < NCH; ++j) uu[j] = utile[t * toff[j] + coff[j] + ss * 16];
                };
                fetch(0, xb[0], uq[0]);
                static_for<0, 4>([&](auto S_) {
                    constexpr int ss = decltype(S_)::value, cur = ss & 1, nxt = cur ^ 1;
                    // scheduling fences (no instructions): pin the prefetch behind the previous sub-step's
                    // arithmetic and in front of this one's
    #pragma unroll
                    for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(acc2[j][0]) : : "memory");
                    if constexpr (ss + 1 < 4) fetch(ss + 1, xb[nxt], uq[nxt]);
    #pragma unroll
                    for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(uq[cur][j]) : : "memory");
    #pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const double u = uq[cur][j];
                        double d[G];
    #pragma unroll
                        for (int I = 0; I < G; ++I) d[I] = xb[cur][I] - mu[j][I];
                        if constexpr (!AUG) acc0[j] += u;
                        int b = 0;
    #pragma unroll
                        for (int I = 0; I < G; ++I) {
                            const double ud = u * d[I];
                            if constexpr (!AUG) acc1[j][I] += ud;
    #pragma unroll
                            for (int J = 0; J <= I; ++J, ++b)
                                acc2[j][b] = __builtin_amdgcn_mfma_f64_4x4x4f64(ud, d[J], acc2[j][b], 0, 0, 0);
                        }
                    }
                });
            }
        }
        __syncthreads();                                           // LDS buffers are rewritten next round
    }

    // ---------------------------------------------------------------------- results
    const long long chunk = (long long)blockIdx.x * TS + ts0;
    {
        // accumulator lane layouts: acc2 -- lane 16 i + 4 blk + j; acc0 / acc1 -- per (sample, ci)
    #pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = cb + CW * j;
            if (c >= K) continue;
            double *out = a.partials + ((size_t)chunk * K + c) * PS;
            if constexpr (!AUG) {
                double s0 = acc0[j];                                   // every sample appears in 4 lanes (ci)
                s0 += __shfl_xor(s0, 4, 64);
                s0 += __shfl_xor(s0, 8, 64);
                s0 += __shfl_xor(s0, 16, 64);
                s0 += __shfl_xor(s0, 32, 64);
                if (lane == 0) out[0] = s0;
            }
            int b = 0;
    #pragma unroll
            for (int I = 0; I < G; ++I) {
                if constexpr (!AUG) {
                    double m = acc1[j][I];
                    m += __shfl_xor(m, 4, 64);
                    m += __shfl_xor(m, 8, 64);
                    m += __shfl_xor(m, 16, 64);
                    m += __shfl_xor(m, 32, 64);
                    if (lane < 4 && 4 * I + lane < D) out[1 + 4 * I + lane] = m;
                }
    #pragma unroll
                for (int J = 0; J <= I; ++J, ++b) {
                    double v = acc2[j][b];
                    v += __shfl_xor(v, 4, 64);                         // sum of the 4 batch blocks
                    v += __shfl_xor(v, 8, 64);
                    const int gi = 4 * I + (lane >> 4), gj = 4 * J + (lane & 3);
                    if (blk == 0 && gj <= gi) {
                        if (gi < D) out[1 + D + gi * (gi + 1) / 2 + gj] = v;
                        else if (AUG && gi == D) out[gj < D ? 1 + gj : 0] = v;     // sum u d_gj / sum u
                    }
                }
            }
        }
    }
    if constexpr (KIND == PMC_KIND_STUDENT_T) {
        // one entry per (workgroup, tile slot): its QS wavefronts fill their component ranges
        double *vp = a.vpartials + ((size_t)blockIdx.x * TPR + ta) * K * 2;
        if (k0 + lane < k1) {
            vp[2 * (k0 + lane)] = vs1;
            vp[2 * (k0 + lane) + 1] = vs2;
        }
    }
    // scalars: one value per workgroup
    __shared__ double sred[FW];
    const double v = wave_sum(sc_a);
    if (lane == 0) sred[w] = v;
    __syncthreads();
    if (threadIdx.x < PMC_NSCALARS) {
        double tot = 0.0;
        const int slot = (KIND == PMC_KIND_VB) ? 0 : 3;
        if ((int)threadIdx.x == slot) {
#pragma unroll
            for (int i = 0; i < FW; ++i) tot += sred[i];
        }
        a.spartials[(size_t)blockIdx.x * PMC_NSCALARS + threadIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------------------------
// D <= PMC_F_REG_MAX_DIM: the whole E-step of a (tile, component group) in REGISTERS.
//
// A wavefront keeps lane = sample throughout, owns KQ <= pmc_freg_kqmax(D) components of its tile and
// carries the 1 + D + D(D+1)/2 moments of each of them per lane over all its rounds: a_nk, e_nk and u_nk
// never leave the vector registers, nothing is parked in LDS and the statistics need no second data layout.
// LDS only carries the soft-max exchange between the QS wavefronts that share a tile (row maximum, then the
// sum -- two barriers per round, none if K <= KQ).  The 64 per-lane moments are summed once, at the end.
// Per (sample, component) at D = 2: 13 + 20 + 5 + 10 vector instructions against 77 of the LDS form above
// (profiles/r02_fused_small_d.txt).
// ---------------------------------------------------------------------------------------------

#if PMC_D <= PMC_F_REG_MAX_DIM

__host__ __device__ constexpr int freg_min_waves(int D, int KQ)
{
#ifdef PMC_F_REG_MIN_WAVES
    return PMC_F_REG_MIN_WAVES;
#else
    return (pmc_stats_stride_c(D) * KQ <= 24) ? 4 : 2;      // accumulators: 2 registers each
#endif
}

template <int D, bool PADDED, int KIND, int KQ, bool FULL>
__global__ __launch_bounds__(FW * 64, freg_min_waves(D, KQ)) void k_estep_reg(const PmcArgsF a)
{
    constexpr int T = pmc_tri(D), STRIDE = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    __shared__ double red[3][FW][64];
    const int K = a.K;
    const int QS = a.qs, TPR = FW / QS;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ta = w / QS, q = w % QS;
    const int k0 = q * KQ;
    const int nk = FULL ? KQ : ((K - k0 < KQ) ? K - k0 : KQ);      // >= 1 (geometry: (QS - 1) KQ < K)
    // parameters of component slot j; slots beyond nk repeat the last component: harmless for the row
    // maximum, skipped (wave-uniform branches) everywhere else
    int slot[KQ];
#pragma unroll
    for (int j = 0; j < KQ; ++j) slot[j] = ((FULL || j < nk) ? k0 + j : k0 + nk - 1) * STRIDE;

    const ExpConst EC;
    double acc[KQ][PS];
#pragma unroll
    for (int j = 0; j < KQ; ++j)
#pragma unroll
        for (int p = 0; p < PS; ++p) acc[j][p] = 0.0;
    double sc_a = 0.0;

    const long long nrounds = (a.ntiles + TPR - 1) / TPR;
    const long long r0 = (long long)blockIdx.x * a.rounds_per_wg;
    long long r1 = r0 + a.rounds_per_wg;
    if (r1 > nrounds) r1 = nrounds;
    // the next round's sample row and weight are loaded a round ahead
    double xn[D], swn;
    auto fetch = [&](long long round) {
        const long long n = (round * TPR + ta) * 64 + lane;
        load_row<D, PADDED>(a.x, n, a.N, a.dreal, xn);
        swn = n < a.N ? (a.sample_w != nullptr ? a.sample_w[n] : 1.0) : 0.0;
    };
    fetch(r0);
    for (long long round = r0; round < r1; ++round) {
        double xv[D];
#pragma unroll
        for (int i = 0; i < D; ++i) xv[i] = xn[i];
        double swv = swn;
        if (round + 1 < r1) fetch(round + 1);
        // the parameters are re-read through the scalar cache every round: kept live over the loop, KQ x 10
        // doubles do not fit the scalar registers and come back as v_readlane spills (16 extra vector
        // instructions per pair at KQ = 8)
        cdouble *pb = (cdouble *)a.pack;
        asm volatile("" : "+s"(pb));
        cdouble *pks[KQ];
#pragma unroll
        for (int j = 0; j < KQ; ++j) pks[j] = pb + slot[j];
        // the points the moments are taken about: the components' means unless the caller names others
        cdouble *sb = (cdouble *)(a.shift_pack ? a.shift_pack : a.pack);
        asm volatile("" : "+s"(sb));
        cdouble *sps[KQ];
#pragma unroll
        for (int j = 0; j < KQ; ++j) sps[j] = sb + slot[j];
        {   // a NaN or infinite coordinate makes every a_nk NaN in the reference; exp_le0 would turn that into
            // zeros, so the sample's weight carries the NaN instead (0 * finite = 0 otherwise)
            double t = xv[0];
#pragma unroll
            for (int i = 1; i < D; ++i) t += xv[i];
            swv = fma(0.0, t, swv);
        }
        // ---- pass 1: a_nk and the row maximum                       (variational.pyx:675-727, pmc.pyx:24-34)
        double av[KQ];
        double M = a.max_init_zero ? 0.0 : -DBL_MAX;
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const double maha = mahalanobis<D>(xv, pks[j]);
            double expo;
            av[j] = component_value<D, KIND>(maha, pks[j] + D + T, expo);
            M = max_f64(av[j], M);                                  // NaN never wins, as with the reference's >
        }
        if (QS > 1) {
            red[0][w][lane] = M;
            __syncthreads();
            M = a.max_init_zero ? 0.0 : -DBL_MAX;
            for (int qq = 0; qq < QS; ++qq) M = max_f64(red[0][ta * QS + qq][lane], M);
        }
        // ---- pass 2: one exp per pair, the row sum                   (variational.pyx:741-743, _regularize.pyx:79)
        double s = 0.0, tb = 0.0;
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            if (FULL || j < nk) {
                const double lr = av[j] - M;
                const double e = exp_le0(lr, EC);
                if constexpr (KIND == PMC_KIND_VB) {
                    s += e;
                    tb = fma(e, max_f64(lr, -1075.0), tb);             // sum_k e_k (a_k - M), for E[log q(Z)]; e = 0 there
                } else {
                    s += pks[j][D + T + 4] * e;
                }
                av[j] = e;
            }
        }
        if (QS > 1) {
            red[1][w][lane] = s;
            if constexpr (KIND == PMC_KIND_VB) red[2][w][lane] = tb;
            __syncthreads();
            s = 0.0;
            tb = 0.0;
            for (int qq = 0; qq < QS; ++qq) {
                s += red[1][ta * QS + qq][lane];
                if constexpr (KIND == PMC_KIND_VB) tb += red[2][ta * QS + qq][lane];
            }
        }
        // the per-sample part of the scalar sum (a log) takes turns among the tile's wavefronts: they meet at the
        // next barrier anyway, and the one that always did it was always the last to arrive
        const int logq = (int)(round & (QS - 1));
        // ---- pass 3: u_nk and its moments                            (variational.pyx:748-755, :855-932; pmc.pyx:36-43, :188-222)
        double f0, f1 = 0.0;
        if constexpr (KIND == PMC_KIND_VB) {
            f0 = 1. / s;
            if (q == logq) sc_a += swv * fma(tb, f0, log_any(f0));   // variational.pyx:1003-1013
        } else {
            const double lse = log_any(s) + M;                       // _regularize.pyx:81
            f0 = exp(M);
            f1 = 1. / (exp(lse) + TINY);                             // pmc.pyx:41 (one division per sample)
            if (q == logq) sc_a += swv * lse;                        // pmc.pyx:388-391
        }
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            if (FULL || j < nk) {
                double u;
                if constexpr (KIND == PMC_KIND_VB) {
                    u = swv * zero_to_tiny(av[j] * f0);
                } else {
                    // exp(log q_k) = e exp(M) first: it underflows where the reference's does (pmc.pyx:39)
                    u = swv * (((av[j] * f0) * pks[j][D + T + 4]) * f1);
                }
                double d[D];
#pragma unroll
                for (int i = 0; i < D; ++i) d[i] = xv[i] - sps[j][i];
                acc[j][0] += u;
                int p = 1 + D;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const double ud = u * d[i];
                    acc[j][1 + i] += ud;
#pragma unroll
                    for (int jj = 0; jj <= i; ++jj, ++p) acc[j][p] = fma(ud, d[jj], acc[j][p]);
                }
            }
        }
    }

    // ---- results: one partial vector per (workgroup, tile slot), layout of k_stats' partials
    const long long chunk = (long long)blockIdx.x * TPR + ta;
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        if (FULL || j < nk) {
            double *out = a.partials + ((size_t)chunk * K + k0 + j) * PS;
#pragma unroll
            for (int p = 0; p < PS; ++p) {
                const double v = wave_sum(acc[j][p]);
                if (lane == 0) out[p] = v;
            }
        }
    }
    __shared__ double sred[FW];
    const double v = wave_sum(sc_a);
    if (lane == 0) sred[w] = v;
    __syncthreads();
    if (threadIdx.x < PMC_NSCALARS) {
        double tot = 0.0;
        const int slot = (KIND == PMC_KIND_VB) ? 0 : 3;
        if ((int)threadIdx.x == slot) {
#pragma unroll
            for (int i = 0; i < FW; ++i) tot += sred[i];
        }
        a.spartials[(size_t)blockIdx.x * PMC_NSCALARS + threadIdx.x] = tot;
    }
}

template <int KIND, int KQ> hipError_t launch_reg_kq(const PmcArgsF &a, unsigned grid, hipStream_t st)
{
    if constexpr (KQ <= pmc_freg_kqmax(D_)) {
        if (a.K == a.qs * KQ)
            hipLaunchKernelGGL((k_estep_reg<D_, P_, KIND, KQ, true>), dim3(grid), dim3(FW * 64), 0, st, a);
        else
            hipLaunchKernelGGL((k_estep_reg<D_, P_, KIND, KQ, false>), dim3(grid), dim3(FW * 64), 0, st, a);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

template <int KIND> hipError_t launch_reg_k(const PmcArgsF &a, unsigned grid, hipStream_t st)
{
    switch (a.kq) {
    case 1: return launch_reg_kq<KIND, 1>(a, grid, st);
    case 2: return launch_reg_kq<KIND, 2>(a, grid, st);
    case 3: return launch_reg_kq<KIND, 3>(a, grid, st);
    case 4: return launch_reg_kq<KIND, 4>(a, grid, st);
    case 5: return launch_reg_kq<KIND, 5>(a, grid, st);
    case 6: return launch_reg_kq<KIND, 6>(a, grid, st);
    case 7: return launch_reg_kq<KIND, 7>(a, grid, st);
    case 8: return launch_reg_kq<KIND, 8>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
}

#endif   // PMC_D <= PMC_F_REG_MAX_DIM

#if PMC_D > PMC_F_REG_MAX_DIM

template <int KIND, int NCH> hipError_t launch_fused_kq(const PmcArgsF &a, unsigned grid, hipStream_t st)
{
    const size_t lds = sizeof(double) * fused_lds_doubles(D_, a.qs, a.K);
    if (lds > 65536) {
        // grows with K; per DEVICE (the attribute is a device's, and one process may drive several: pmc_init_devices) and
        // safe against two host threads arriving at once (a second identical call is harmless)
        static std::atomic<size_t> configured[256];
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::atomic<size_t> &have = configured[dev & 255];
        if (lds > have.load(std::memory_order_acquire)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_estep_fused<D_, P_, KIND, NCH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            size_t cur = have.load(std::memory_order_relaxed);
            while (cur < lds && !have.compare_exchange_weak(cur, lds, std::memory_order_release)) {}
        }
    }
    hipLaunchKernelGGL((k_estep_fused<D_, P_, KIND, NCH>), dim3(grid), dim3(FW * 64), lds, st, a);
    return hipGetLastError();
}

template <int KIND> hipError_t launch_fused_k(int nch, const PmcArgsF &a, unsigned grid, hipStream_t st)
{
    switch (nch) {
    case 1: return launch_fused_kq<KIND, 1>(a, grid, st);
    case 2: return launch_fused_kq<KIND, 2>(a, grid, st);
    case 4: return launch_fused_kq<KIND, 4>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
}

#endif   // PMC_D > PMC_F_REG_MAX_DIM

}  // namespace

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_fused_d, PMC_D, PMC_PADDED)(int kind, int nch, const PmcArgsF &a,
                                                                            unsigned grid, hipStream_t st)
{
#if PMC_D <= PMC_F_REG_MAX_DIM
    (void)nch;
    if (!a.reg) return hipErrorInvalidValue;               // the dispatcher's geometry and this unit disagree
    switch (kind) {
    case PMC_KIND_GAUSS: return launch_reg_k<PMC_KIND_GAUSS>(a, grid, st);
    case PMC_KIND_VB: return launch_reg_k<PMC_KIND_VB>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
#else
    if (a.reg) return hipErrorInvalidValue;
    switch (kind) {
    case PMC_KIND_GAUSS: return launch_fused_k<PMC_KIND_GAUSS>(nch, a, grid, st);
    case PMC_KIND_STUDENT_T: return launch_fused_k<PMC_KIND_STUDENT_T>(nch, a, grid, st);
    case PMC_KIND_VB: return launch_fused_k<PMC_KIND_VB>(nch, a, grid, st);
    default: return hipErrorInvalidValue;
    }
#endif
}

extern "C" int PMC_UNIT_NAME_X(pmc_fused_lds_bytes_d, PMC_D, PMC_PADDED)(int qs, int K)
{
    return (int)(sizeof(double) * fused_lds_doubles(D_, qs, K));
}

#else   // this dimension has no fused kernel

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_fused_d, PMC_D, PMC_PADDED)(int, int, const PmcArgsF &, unsigned,
                                                                            hipStream_t)
{
    return hipErrorNotSupported;
}

extern "C" int PMC_UNIT_NAME_X(pmc_fused_lds_bytes_d, PMC_D, PMC_PADDED)(int, int) { return -1; }

#endif
