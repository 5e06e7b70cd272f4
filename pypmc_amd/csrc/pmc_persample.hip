// pmc_persample.hip / pmc_stats.hip -- the gfx950 kernels of the adaptive-importance-sampling hot
// path, each compiled once per sample dimension:  hipcc -DPMC_D=<D> -DPMC_PADDED=<0|1> -c <file>
//
// Execution model (CDNA4, wave64):
//   * Per-sample kernels (k_logpdf, k_resp): one lane owns one sample; its D coordinates stay in
//     VGPRs for the whole component loop.  Everything that depends on the component only (mean,
//     whitening factor R_k, constants) is wave-uniform, so it is fetched with scalar loads
//     (address space 4 -> s_load_dwordx16 through the scalar cache) and used as the SGPR operand of
//     v_fma_f64: the triangular product y = R_k (x - mu_k) costs D(D+1)/2 v_fmac_f64 and no LDS or
//     vector-memory traffic at all.  From D = 40 on the parameters outgrow the scalar cache and the
//     product moves to the fp64 matrix pipe with the parameters staged in LDS (MahaEngine below).
//   * Statistics kernel (k_stats, pmc_stats.hip): one wavefront owns one component and streams over
//     a chunk of samples, accumulating 4 x 4 blocks of the second moments with the fp64 MFMA.
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
// (RESP: the responsibility kernel.  Measured, ms per 4e6 samples x 32 components: log-pdf D = 16 with 1 / 4 / 5:
// 1.13 / 1.08 / 1.05;  responsibilities D = 24 with 1 / 4: 2.02 / 1.90, D = 16 with 5: worse)
__host__ __device__ constexpr int pmc_min_waves(int D, bool RESP = false)
{
#ifdef PMC_MIN_WAVES
    return PMC_MIN_WAVES;
#else
    return D >= 40 ? 2 : ((!RESP && D == 16) ? 5 : ((RESP && D == 24) ? 4 : 1));
#endif
}

// The wavefronts of a workgroup walk the components in step: a barrier per component keeps them on
// the same parameter lines, so that one wavefront's scalar-cache fill serves the other three
// (D = 20: -3 % kernel time; neutral at D = 40).
__device__ __forceinline__ void component_sync()
{
#ifndef PMC_NO_COMPONENT_SYNC
    __builtin_amdgcn_s_barrier();
#endif
}

#ifndef PMC_RESIDENT_MAX_DIM_RESP
#define PMC_RESIDENT_MAX_DIM_RESP 8
#endif
#ifndef PMC_RESP_SYNC_FROM
#define PMC_RESP_SYNC_FROM 20
#endif

extern __shared__ double dyn_lds[];   // [MFMA engine: 2 parameter buffers] [k_resp: parked values]

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// ---------------------------------------------------------------------------------------------
// Mahalanobis engines: maha_nk = |R_k (x_n - mu_k)|^2 for the wavefront's 64 samples, lane = sample.
// ---------------------------------------------------------------------------------------------
// D < 32: the triangular product on the vector pipe with the parameters as SGPR operands (above).
template <int D, bool PADDED, int ENGINE> struct MahaEngine {
    static constexpr int LDS_DOUBLES = 0;
    double xv[D];
    __device__ __forceinline__ void load(const PmcArgsA &a, long long tile, int lane)
    {
        load_row<D, PADDED>(a.x, tile * 64 + lane, a.N, a.dreal, xv);
    }
    // A short pack needs neither the touch prefetch nor the barrier that keeps a workgroup on one
    // component: with few lines per component they only add two scalar round trips to each.  Measured, ms per
    // 4e6 samples x 32 components with / without: D = 2 0.224 / 0.192, D = 5 0.315 / 0.292, D = 8 0.453 /
    // 0.425, D = 10 0.544 / 0.510, D = 12 (24.5 KB) 0.684 / 0.644 -- and D = 16 1.09 / 1.28, D = 20 1.38 / 2.24.
#ifndef PMC_RESIDENT_BYTES
#define PMC_RESIDENT_BYTES 26000
#endif
#ifndef PMC_RESIDENT_MAX_DIM
#define PMC_RESIDENT_MAX_DIM 12
#endif
    // (the responsibility kernel, whose wavefronts also park their values, keeps both from D = 10 on:
    //  D = 12 0.85 -> 1.18 ms without)
    bool resident, sync;
    static __device__ __forceinline__ bool fits(int K, int maxdim = PMC_RESIDENT_MAX_DIM)
    {
        return D <= maxdim && K * pmc_pack_stride_c(D) * 8 <= PMC_RESIDENT_BYTES;
    }
    // The per-component barrier pays from D = 20 on in the responsibility kernel (whose wavefronts also park their
    // values and drift apart less): ms per 4e6 samples x 32 components with / without it: D = 12 0.768 / 0.729,
    // D = 16 1.085 / 1.042, D = 20 1.303 / 1.381, D = 24 1.73 / 1.99, D = 30 2.72 / 3.27; the log-pdf kernel keeps it
    // from D = 16 on (D = 16 neutral, D = 20 1.32 / 1.41, D = 30 2.75 / 2.99).
    static_assert(PMC_RESIDENT_MAX_DIM != PMC_RESIDENT_MAX_DIM_RESP, "the two kernels are told apart by this bound");
    static __device__ __forceinline__ bool syncs(int maxdim)
    {
        return maxdim == PMC_RESIDENT_MAX_DIM || D >= PMC_RESP_SYNC_FROM;
    }
    __device__ __forceinline__ void begin(const double *, int K, int maxdim = PMC_RESIDENT_MAX_DIM)
    {
        resident = fits(K, maxdim);
        sync = syncs(maxdim);
    }
    __device__ __forceinline__ double eval(cdouble *pk, int)
    {
        // D >= 8: the scalar loads scheduled by hand, one group of lines ahead (mahalanobis_sp, pmc_device.h).  ms per 4e6
        // samples, responsibilities of K = 32 / 64 components, [compiler's schedule] -> one line ahead -> two lines:
        //   D =  8  0.50 / 0.90 -> 0.50 / 0.87 -> 0.49 / 0.875      D = 20  1.285 / 2.455 -> 1.21 / 2.325 -> 1.23 / 2.31
        //   D = 12  0.69 / 1.26 -> 0.665 / 1.18 -> 0.67 / 1.20      D = 24  1.76 / 3.75 -> 1.68 / 3.49 -> 1.64 / 3.43
        //   D = 16  0.955 / 1.91 -> 0.895 / 1.75 -> 0.90 / 1.67     D = 30  2.68 / 5.48 -> 2.68 / 5.47 -> 2.41 / 5.05
#ifndef PMC_SP_GROUP
#define PMC_SP_GROUP ((D <= 12 || D == 20) ? 1 : 2)
#endif
        if constexpr (D >= 8 && PMC_SP_GROUP != 0) {
            if (!resident && sync) component_sync();      // workgroup-uniform
            return mahalanobis_sp<D, (PMC_SP_GROUP != 0 ? PMC_SP_GROUP : 1)>(xv, pk, !resident);
        }
        if (!resident) {                                  // workgroup-uniform
            if (sync) component_sync();
            touch_component<D>(pk);
        }
        return mahalanobis<D>(xv, pk);
    }
    // a wavefront without samples keeps the workgroup's barrier count
    __device__ static __forceinline__ void idle(const double *, int K, int maxdim = PMC_RESIDENT_MAX_DIM)
    {
        if (fits(K, maxdim) || !syncs(maxdim)) return;
        for (int k = 0; k < K; ++k) component_sync();
    }
};

// PMC_DPP_FROM <= D < 32: the triangular product on the vector pipe with the coefficients streamed
// through a window of W VGPRs -- 16 consecutive coefficients per register, the same 16 in every 16-lane
// row, loaded straight from the pack with ordinary vector loads (L2 / L1 hits) one window ahead of their
// use, and consumed by v_fmac_f64 with a DPP row broadcast (fmac_bcast above).  Nothing but the mean
// and the five constants of a component goes through the scalar cache any more (4 lines instead of
// 30 at D = 20: a K = 32 pack stays resident), there is no touch prefetch, no wait for a scalar fill
// and no barrier -- which is what kept the SGPR form at 87-89 % of the issue slots.
// The factor is stored with a unit diagonal, maha = sum_i s_i (d_i + sum_{j>i} U_ij d_j)^2: row i
// accumulates IN PLACE in d_i (row i is the last reader of d_i), so no accumulator has to be zeroed
// and the instruction count equals the SGPR form's (D subtractions, T - D + 2 D multiply-adds).
template <int D, bool PADDED> struct MahaEngine<D, PADDED, PMC_ENG_DPP> {
    static constexpr int LDS_DOUBLES = 0;
    static constexpr int T = pmc_tri(D), STRIDE = pmc_pack_stride_c(D);
    static constexpr int NPR = (T + 15) / 16;                     // registers' worth of coefficients
    // window: a divisor of the (possibly padded by one) register count, so that the register of a
    // coefficient is the same for every component
    static constexpr int pick_w(int np)
    {
#ifdef PMC_DPP_W
        for (int w = PMC_DPP_W; w >= 2; --w) if (np % w == 0) return w;
#else
        for (int w = 8; w >= 5; --w) if (np % w == 0) return w;
#endif
        return 0;
    }
    static constexpr int NP = pick_w(NPR) ? NPR : NPR + 1;
    static constexpr int W = pick_w(NP) ? pick_w(NP) : NP;
    static_assert(NP % W == 0, "window must divide the register count");
    double xv[D];
    double rb[W];
    const double *lanebase;      // pack + D + (lane & 15): this lane's coefficient of register 0, component 0
    int lastrel;                 // offset of the last (partial) register's coefficient, clamped into the factor
    int K;

    __device__ __forceinline__ void load(const PmcArgsA &a, long long tile, int lane)
    {
        load_row<D, PADDED>(a.x, tile * 64 + lane, a.N, a.dreal, xv);
        const int n = lane & 15;
        lastrel = (16 * (NPR - 1) + n < T ? 16 * (NPR - 1) + n : T - 1) - n;
    }
    // register v of component k (clamped to the pack's last component)
    template <int V> __device__ __forceinline__ double fetch(int k) const
    {
        const double *p = lanebase + (size_t)(k < K ? k : K - 1) * STRIDE;
        if constexpr (V < NPR - 1) return p[16 * V];
        else return p[lastrel];                                      // the partial register / window padding
    }
    __device__ __forceinline__ void begin(const double *pack, int K_, int = 0)
    {
        K = K_;
        lanebase = pack + D + (threadIdx.x & 15);
        static_for<0, W>([&](auto V) { rb[decltype(V)::value] = fetch<decltype(V)::value>(0); });
    }
    // coefficient c of the stream: multiply-add it into acc, then -- if it was its register's last one --
    // refill the register with the one a window further on (the scheduling barrier keeps the load HERE:
    // left alone the scheduler sinks it to just in front of its first use)
    template <int C> __device__ __forceinline__ void use(double &acc, double operand, int k)
    {
        constexpr int v = C / 16, n = C % 16;
        fmac_bcast<n>(acc, rb[v % W], operand);
        if constexpr (n == 15 || C == T - 1) refill<v>(k);
    }
    template <int V> __device__ __forceinline__ void refill(int k)
    {
        if constexpr (V + W < NP) rb[V % W] = fetch<V + W>(k);
        else rb[V % W] = fetch<V + W - NP>(k + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ double eval(cdouble *pk, int k)
    {
        double d[D];
#pragma unroll
        for (int j = 0; j < D; ++j) d[j] = xv[j] - pk[j];
        double ma = 0.0, mb = 0.0;
        // Rows in pairs (i, i + 1), their multiply-adds alternating so that no instruction depends on its
        // predecessor; the coefficients sit in the pack in exactly this order (pmc_pack_components):
        //   U_i,i+1 | U_i+1,j  U_i,j  (j = i+2 .. D-1) | s_i  s_i+1
        static_for<0, (D + 1) / 2>([&](auto Q_) {
            constexpr int i = 2 * decltype(Q_)::value;
            constexpr int c0 = i * D - i * (i - 1) / 2;             // coefficients in front of row i
            if constexpr (i + 1 < D) {
                use<c0>(d[i], d[i + 1], k);
                static_for<0, D - 2 - i>([&](auto T_) {
                    constexpr int t = decltype(T_)::value, j = i + 2 + t;
                    use<c0 + 1 + 2 * t>(d[i + 1], d[j], k);
                    use<c0 + 2 + 2 * t>(d[i], d[j], k);
                });
                const double ta = d[i] * d[i], tb = d[i + 1] * d[i + 1];
                use<c0 + 2 * D - 3 - 2 * i>(ma, ta, k);
                use<c0 + 2 * D - 2 - 2 * i>(mb, tb, k);
            } else {
                const double ta = d[i] * d[i];
                use<c0>(ma, ta, k);
            }
        });
        // (window padding beyond the last coefficient is refilled like a register)
        static_for<NPR, NP>([&](auto V_) { refill<decltype(V_)::value>(k); });
        return ma + mb;
    }
    __device__ static __forceinline__ void idle(const double *, int, int = 0) {}
};

// D >= 32 (multiples of 4; measured break-even: D = 32 log-pdf -17 %, responsibilities +2 %;
// D = 24 +15 %): the scalar path cannot feed the vector pipe any more (6.9 KB = 108 cache lines per
// component at D = 40, whose fill takes as long as the arithmetic; 54 % utilisation).  Here the
// workgroup stages each component's parameters ONCE in LDS (LDS-DMA, double buffered, one barrier per
// component) and the triangular product runs on the matrix pipe, 4 x 4 blocks of R against 16 samples
// per v_mfma_f64_4x4x4_4b_f64 (lane layout: pmc_stats.hip):
//     A[blk][i][c] = R[4I+i][4J+c]  lane 16c + 4blk + i   (the same block in all 4 batch slots)
//     B[blk][c][j] = d[s][4J+c]     lane 16c + 4blk + j   (sample s = 4 blk + j of the sub-tile)
//     C[blk][i][j] = y[s][4I+i]     lane 16i + 4blk + j
// so a lane holds coordinate (lane >> 4) of sample (lane & 15) of each of the tile's four sub-tiles,
// squares its y, and after the block rows two cross-lane additions give every lane the sub-tile's
// |y|^2; lane l then keeps sub-tile (l >> 4): sample l of the tile, the layout the epilogues expect.
template <int D, bool PADDED> struct MahaEngine<D, PADDED, PMC_ENG_MFMA> {
    static constexpr int G = D / 4, STRIDE = pmc_pack_stride_c(D);
    static constexpr int PIECES = (STRIDE * 8 + 1023) / 1024;     // 1-KiB DMA pieces per component
    static constexpr int PBUF = PIECES * 128;                      // doubles per parameter buffer
    static constexpr int LDS_DOUBLES = 2 * PBUF;
#ifdef PMC_MFMA_NT
    static constexpr int NT = PMC_MFMA_NT;
#else
    // sub-tiles sharing an A block: as many as the registers hold (measured, ms per 2e6 samples:
    // D=40 K=128: NT 1/2/4 = 11.1 / 9.0 / 8.4;  D=48 K=32: 2/4 = 3.37 / 2.99;  D=64 K=16: 1/2 = 3.48 / 3.23)
    static constexpr int NT = G <= 12 ? 4 : 2;
#endif
    static_assert(D % 4 == 0, "MFMA engine needs whole coordinate groups");
    double xm[4][G];             // x[16 s + (lane & 15)][4J + (lane >> 4)]
    int rowbase[G];              // double index of R[4I + (lane & 3)][lane >> 4] in a component's pack
    const double *pack;
    int K, lane, wave;

    __device__ __forceinline__ void load(const PmcArgsA &a, long long tile, int lane_)
    {
        lane = lane_;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int c = lane >> 4, s16 = lane & 15, i = lane & 3;
        const int dreal = PADDED ? a.dreal : D;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long long row = tile * 64 + s * 16 + s16;
            const double *p = a.x + (row < a.N ? row : 0) * (long long)dreal;
#pragma unroll
            for (int J = 0; J < G; ++J) {
                const int col = 4 * J + c;
                double v = 0.0;
                if (row < a.N && (!PADDED || col < dreal)) v = p[col];
                xm[s][J] = v;
            }
        }
#pragma unroll
        for (int I = 0; I < G; ++I) {
            const int r = 4 * I + i;                      // row of R; element (r, col) sits at
            rowbase[I] = D + r * D - r * (r - 1) / 2 - r + c;   // rowbase + 4 J   (col = 4J + c >= r)
        }
    }
    // HBM/L2 -> LDS copy of component k's parameters into buffer (k & 1)
    __device__ __forceinline__ void stage(int k)
    {
        const double *src = pack + (size_t)k * STRIDE;
        double *dst = dyn_lds + (k & 1) * PBUF;
#pragma unroll
        for (int q = 0; q < (PIECES + PMC_A_WAVES - 1) / PMC_A_WAVES; ++q) {
            const int piece = wave + q * PMC_A_WAVES;     // wave-uniform
            if (piece < PIECES) {
                int o = piece * 128 + 2 * lane;           // doubles
                if (o > STRIDE - 2) o = STRIDE - 2;       // the last piece may be partial
                __builtin_amdgcn_global_load_lds((gvoid_t *)(src + o), (lvoid_t *)(dst + piece * 128), 16, 0, 0);
            }
        }
    }
    __device__ __forceinline__ void begin(const double *pack_, int K_, int = 0)
    {
        pack = pack_;
        K = K_;
        __syncthreads();                                  // nobody still reads buffer 0
        stage(0);
    }
    __device__ __forceinline__ double eval(cdouble *, int k)
    {
        dma_barrier();                                    // component k has landed; k-1 is consumed
        if (k + 1 < K) stage(k + 1);
        const double *buf = dyn_lds + (k & 1) * PBUF;
        const int c = lane >> 4, i = lane & 3;
        // A operand of block (I, J); diagonal blocks hold zeros below the diagonal
        auto block = [&](int I, int J) -> double {
            const double A = buf[rowbase[I] + 4 * J];
            return (J == I && c < i) ? 0.0 : A;
        };
        double total[4];
        // NT sub-tiles of 16 samples at a time share every A block; two block rows per sweep: 2 NT
        // independent accumulator chains for the matrix pipe
#pragma unroll
        for (int s0 = 0; s0 < 4; s0 += NT) {
            double d[NT][G], q[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                q[t] = 0.0;
#pragma unroll
                for (int J = 0; J < G; ++J) d[t][J] = xm[s0 + t][J] - buf[4 * J + c];
            }
#pragma unroll
            for (int I = 0; I < G; I += 2) {
                // fence: the A blocks of this sweep are not read before the previous sweep is done
                // (without it every ds_read of the unrolled component is hoisted to its top: spills)
                asm volatile("" : "+v"(q[0]) : : "memory");
                double ya[NT], yb[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) ya[t] = yb[t] = 0.0;
#pragma unroll
                for (int J = I; J < G; ++J) {
                    const double Aa = block(I, J);
#pragma unroll
                    for (int t = 0; t < NT; ++t) ya[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(Aa, d[t][J], ya[t], 0, 0, 0);
                    if (J > I && I + 1 < G) {
                        const double Ab = block(I + 1, J);
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            yb[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ab, d[t][J], yb[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    q[t] = fma(ya[t], ya[t], q[t]);
                    q[t] = fma(yb[t], yb[t], q[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double v = q[t];
                v += __shfl_xor(v, 16, 64);               // sum over the 4 coordinates of each group
                v += __shfl_xor(v, 32, 64);
                total[s0 + t] = v;
            }
        }
        return c == 0 ? total[0] : (c == 1 ? total[1] : (c == 2 ? total[2] : total[3]));
    }
    __device__ static __forceinline__ void idle(const double *pack_, int K_, int = 0)
    {
        MahaEngine e;
        e.lane = threadIdx.x & 63;
        e.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        e.begin(pack_, K_);
        for (int k = 0; k < K_; ++k) {
            dma_barrier();
            if (k + 1 < K_) e.stage(k + 1);
        }
    }
};

// Compiled "dimension" 0: the sample dimension is a run-time value (D > PMC_MAX_DIM).  k_big_maha (pmc_big.hip)
// has already written maha_nk tile-major -- into the kept-tiles buffer, the responsibility buffer itself or a
// scratch -- and the per-sample kernels below run unchanged on top of it: eval() is one coalesced load.
template <bool PADDED> struct MahaEngine<0, PADDED, PMC_ENG_TILES> {
    static constexpr int LDS_DOUBLES = 0;
    const double *m1, *m2, *pack1, *cur;
    size_t tile;
    int lane;
    __device__ __forceinline__ void load(const PmcArgsA &a, long long tile_, int lane_)
    {
        tile = (size_t)(tile_ * 64 < a.N ? tile_ : 0);     // a wavefront beyond the samples reads tile 0 (discarded)
        lane = lane_;
        m1 = a.mtile;
        m2 = a.mtile2;
        pack1 = a.pack;
    }
    __device__ __forceinline__ void begin(const double *pack, int K, int = 0)
    {
        cur = (pack == pack1 ? m1 : m2) + tile * K * 64 + lane;
    }
    __device__ __forceinline__ double eval(cdouble *, int k) { return cur[(size_t)k * 64]; }
    __device__ static __forceinline__ void idle(const double *, int, int = 0) {}
};

// D, D(D+1)/2 and the pack's component stride: compile-time constants of a compiled dimension, run-time values of
// unit 0
template <int D> struct Dims {
    static constexpr int T = pmc_tri(D), STRIDE = pmc_pack_stride_c(D), DT = D + pmc_tri(D);
    __device__ __forceinline__ explicit Dims(int) {}
};
template <> struct Dims<0> {
    int T, STRIDE, DT;
    __device__ __forceinline__ explicit Dims(int d) : T(pmc_tri(d)), STRIDE(pmc_pack_stride_c(d)), DT(d + pmc_tri(d)) {}
};

template <int D> __host__ __device__ constexpr int pmc_use_mfma() { return pmc_engine(D); }

// ---------------------------------------------------------------------------------------------
// k_logpdf: MixtureDensity.multi_evaluate (mixture.pyx:112-156) + logsumexp2D
// (_regularize.pyx:57-84) [+ importance weights, importance_sampling.py:197-215] in one pass.
// ---------------------------------------------------------------------------------------------
template <int D, bool PADDED, int KIND, int KIND2>
__global__ __launch_bounds__(PMC_A_WAVES * 64, pmc_min_waves(D)) void k_logpdf(const PmcArgsA a)
{
    // behind k_mgemm: only the workgroups its guard refused (block-uniform, in front of every barrier)
    if (a.blockflag != nullptr && (*a.redo == 0 || a.blockflag[blockIdx.x] == 0)) return;
    const Dims<D> dm(a.dreal);
    const long long n = ((long long)blockIdx.x * PMC_A_WAVES * 64) + threadIdx.x;
    const bool valid = n < a.N;

    MahaEngine<D, PADDED, pmc_use_mfma<D>()> engine;
    engine.load(a, (long long)blockIdx.x * PMC_A_WAVES + (threadIdx.x >> 6), threadIdx.x & 63);

    // the mixture itself (kind KIND), then -- pmc_importance_weights only -- the TARGET mixture of the
    // importance weights (kind KIND2), evaluated on the same registers: the samples are read once
    const ExpConst EC;
    RowPoison rowp;                                      // a NaN / +inf component value makes the row NaN (lse_step drops it)
    // kept Mahalanobis forms (pmc_*_keep: a.atile), or -- pmc_importance_weights_emit: a.u -- the place where the
    // responsibilities of the PMC update will stand: the forms are parked there and replaced below
    double *const mkeep = a.u != nullptr ? a.u : a.atile;
    const bool keep_tile = mkeep != nullptr && ((n >> 6) << 6) < a.N;       // (the buffer ends with the last live tile)
    // columns of the place the forms are kept in: all K, or -- emitting, with pruned components sorted to the end of the
    // pack (PmcArgsA::ku) -- the components that get responsibilities
    const int kcols = (a.u != nullptr && a.ku > 0) ? a.ku : a.K;
    double m_first = 0.0;                                // row maximum of the FIRST mixture's component values
    auto mixture = [&](auto kind, const double *gpack, const int K, const bool first) -> double {
        constexpr int KD = decltype(kind)::value;
        double m = (first && a.max_init_zero) ? 0.0 : -DBL_MAX, s = 0.0;
        cdouble *pk = (cdouble *)gpack;
        engine.begin(gpack, K);                          // (the global pointer: vector loads of the DPP engine)
        for (int k = 0; k < K; ++k, pk += dm.STRIDE) {
            const double maha = engine.eval(pk, k);
            double expo;
            const double v = component_value<D, KD>(maha, pk + dm.DT, expo);
            if (first && a.individual != nullptr) {
                const long long col = ((cint64 *)pk)[dm.DT + 5];
                if (valid) a.individual[n * a.ld + col] = v;
            }
            if (first && keep_tile && k < kcols)         // wave-uniform: keep maha_nk for the PMC update of these samples
                mkeep[((size_t)(n >> 6) * kcols + k) * 64 + (threadIdx.x & 63)] = maha;
            lse_step(v, pk[dm.DT + 4], m, s, EC);
            rowp.see(v);
        }
        if (first) m_first = m;
        return (log_any(s) + m) + rowp.value();              // _regularize.pyx:81
    };
    const double lse = mixture(ic<KIND>{}, a.pack, a.K, true);
    double lse_target = 0.0;
    if (a.pack2 != nullptr) lse_target = mixture(ic<KIND2>{}, a.pack2, a.K2, false);
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
    if (a.u != nullptr && keep_tile) {
        // pmc_importance_weights_emit: the Rao-Blackwellised responsibilities of the update that follows, weighted
        // with the importance weight just formed, u_nk = w_n rho_nk with rho = exp(log q_k) w_k / (exp(lse) + tiny)
        // (pmc.pyx:23-43) -- the arithmetic of k_resp's PMC branch / k_resp_tiles (exp(a - M) exp(M), so that rho
        // underflows where the reference's exp(log q_k) does), with the row maximum and the log-sum-exp this pass has
        // already: every parked form is read ONCE (k_resp_tiles: three times, in a launch of its own).
        static_assert(KIND == PMC_KIND_GAUSS || KIND == PMC_KIND_STUDENT_T, "emit: density kinds only");
        const double wn = (a.log_target != nullptr || a.pack2 != nullptr) ? sc[0] : 1.0;
        const double swv = valid ? wn + rowp.value() : 0.0;
        const double denom = exp(lse) + TINY;                               // pmc.pyx:41
        const double em = exp(m_first), inv_denom = 1. / denom;
        double *ut = a.u + (size_t)(n >> 6) * kcols * 64 + (threadIdx.x & 63);
        cdouble *pk = (cdouble *)a.pack + (size_t)(kcols - 1) * dm.STRIDE + dm.DT;
        // (behind k_mgemm the sums of the degree-of-freedom condition are k_dof_sums' for every workgroup: a.vpartials is NULL)
        double *vp = (KIND == PMC_KIND_STUDENT_T && a.vpartials != nullptr) ? a.vpartials + (size_t)(n >> 6) * kcols * 2 : nullptr;
        if (a.gscale != nullptr) {
            // standing in for k_mgemm, which leaves a factor per (sample, group of 16 components) to the statistics
            // kernel: this u is complete
            const int G = (kcols + PMC_RESP_GROUP - 1) / PMC_RESP_GROUP;
            for (int gq = 0; gq < G; ++gq) a.gscale[((size_t)(n >> 6) * G + gq) * 64 + (threadIdx.x & 63)] = 1.0;
        }
        for (int k = kcols - 1; k >= 0; --k, pk -= dm.STRIDE) {             // last written first: still in L2
            double expo;
            const double maha = ut[(size_t)k * 64];
            const double v = component_value<D, KIND>(maha, pk, expo);
            const double e = exp_clamped(max_f64(v - m_first, -1075.0), EC);
            const double wr = swv * (((e * em) * pk[4]) * inv_denom);
            if constexpr (KIND == PMC_KIND_STUDENT_T) {
                // gamma and the two sums of the degree-of-freedom condition, as k_resp / k_resp_tiles form them
                const double nu = pk[3];
                const double gamma = (nu + (double)a.dreal) / (nu + maha);          // pmc.pyx:610
                ut[(size_t)k * 64] = wr * gamma;
                if (vp != nullptr) {                                                // wave-uniform
                    const double s1 = wave_sum(wr);                                 // pmc.pyx:612 / :669
                    const double s2 = wave_sum(wr * log_pos(.5 * (maha + nu)));
                    if ((threadIdx.x & 63) == 0) {
                        vp[2 * k] = s1;
                        vp[2 * k + 1] = s2;
                    }
                }
            } else {
                ut[(size_t)k * 64] = wr;
            }
        }
    }
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}

// ---------------------------------------------------------------------------------------------
// k_logpdf2: A/B experiment of round 6 (verdict r5 #3), built only with -DPMC_TWO_PER_LANE and taken only for the plain
// log q / importance-weight pass at D = 8 ... 30: every lane owns TWO samples (n and n + 256 of a block of 512), every
// scalar coefficient feeds two multiply-adds.  Same operations per sample in the same order: k_logpdf's bits.
// What it showed: profiles/r06_two_per_lane_ab.txt.
// ---------------------------------------------------------------------------------------------
#ifdef PMC_TWO_PER_LANE
template <int D, bool PADDED, int KIND, int KIND2>
__global__ __launch_bounds__(PMC_A_WAVES * 64, 2) void k_logpdf2(const PmcArgsA a)
{
    constexpr int STRIDE = pmc_pack_stride_c(D), DT = D + pmc_tri(D);
    const long long na = (long long)blockIdx.x * (2 * PMC_A_WAVES * 64) + threadIdx.x, nb = na + PMC_A_WAVES * 64;
    const bool va = na < a.N, vb = nb < a.N;
    double xa[D], xb[D];
    load_row<D, PADDED>(a.x, na, a.N, a.dreal, xa);
    load_row<D, PADDED>(a.x, nb, a.N, a.dreal, xb);
    const ExpConst EC;
    RowPoison pa, pb;
    auto mixture = [&](auto kind, const double *gpack, const int K, double &lsa, double &lsb) {
        constexpr int KD = decltype(kind)::value;
        double ma = -DBL_MAX, sa = 0.0, mb = -DBL_MAX, sb = 0.0;
        cdouble *pk = (cdouble *)gpack;
        for (int k = 0; k < K; ++k, pk += STRIDE) {
            component_sync();
            double qa, qb, expo;
            mahalanobis_sp2<D, ((D <= 12 || D == 20) ? 1 : 2)>(xa, xb, pk, true, qa, qb);
            const double v_a = component_value<D, KD>(qa, pk + DT, expo), v_b = component_value<D, KD>(qb, pk + DT, expo);
            lse_step(v_a, pk[DT + 4], ma, sa, EC);
            lse_step(v_b, pk[DT + 4], mb, sb, EC);
            pa.see(v_a);
            pb.see(v_b);
        }
        lsa = (log_any(sa) + ma) + pa.value();
        lsb = (log_any(sb) + mb) + pb.value();
    };
    double la, lb, ta = 0.0, tb = 0.0;
    mixture(ic<KIND>{}, a.pack, a.K, la, lb);
    if (a.pack2 != nullptr) mixture(ic<KIND2>{}, a.pack2, a.K2, ta, tb);
    if (a.out != nullptr) {
        if (va) a.out[na] = la;
        if (vb) a.out[nb] = lb;
    }
    if (a.partials == nullptr && a.log_target == nullptr && a.pack2 == nullptr) return;
    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    auto weigh = [&](bool valid, long long n, double lse, double lt) {
        if ((a.log_target != nullptr || a.pack2 != nullptr) && valid) {
            const double tmp = (a.pack2 != nullptr ? lt : a.log_target[n]) - lse;
            const double w = exp(tmp);
            a.weights[n] = w;
            sc[0] += w;
            sc[1] += (w != 0.0) ? w * tmp : 0.0;
            sc[2] += w * w;
            sc[4] += (isinf(w) && !isinf(tmp)) ? 1.0 : 0.0;
        }
        if (valid) sc[3] += lse;
    };
    weigh(va, na, la, ta);
    weigh(vb, nb, lb, tb);
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}
#endif

// ---------------------------------------------------------------------------------------------
// k_logpdf_split: k_logpdf with the components of a sample block split over workgroups (round 6).
//
// A workgroup of k_logpdf walks ALL K components of its 256 samples, so a call costs K x (1 us at D = 20 ... 2.7 us at
// D = 40) whatever N is until the launch fills the chip, and the last round of a launch that does fill it leaves
// compute units idle for up to one such walk.  Here the launch is [whole blocks | pieces]: workgroups [0, split_b1)
// are k_logpdf's; behind them every remaining block is walked by s1 + s2 PIECES, workgroups that take split_c1
// components of the mixture (or split_c2 of the target mixture) with the streaming log-sum-exp of k_logpdf and leave
// their (m, s) -- the row's NaN riding on s -- in split_part.  The piece that draws the block's last ticket combines the
// pairs IN PIECE ORDER (whichever piece that is: same bits from run to run),
//     M = max_p m_p,   S = sum_p s_p exp(m_p - M),   lse = log S + M       (_regularize.pyx:72-81 about the row maximum:
//     a zero-weight component takes part in M through its piece's m_p and adds nothing to S, as in the reference)
// and does what follows the component loop in k_logpdf: log q, the importance weights and their sums.  Per-pair outputs
// (`individual`, kept Mahalanobis forms) leave from the pieces themselves.  Small batches (the reference's own:
// examples/pmc.py:61-65 draws 1e3 samples per step) spread over the chip; large ones end on short pieces.
// A whole block computes exactly k_logpdf's numbers; a block in pieces agrees with them to the rounding of the merge
// (a few ulps of lse).  Not for the emitting pass (u needs the row's maximum before the walk ends).
// ---------------------------------------------------------------------------------------------
template <int D, bool PADDED, int KIND, int KIND2>
__global__ __launch_bounds__(PMC_A_WAVES * 64, pmc_min_waves(D)) void k_logpdf_split(const PmcArgsA a)
{
    const Dims<D> dm(a.dreal);
    const int S = a.split_s1 + a.split_s2;
    long long blk = blockIdx.x;
    int piece = -1;                                      // workgroup-uniform
    if ((int)blockIdx.x >= a.split_b1) {
        const int p = (int)blockIdx.x - a.split_b1;
        blk = a.split_b1 + p / S;
        piece = p % S;
    }
    const long long n = blk * (PMC_A_WAVES * 64) + threadIdx.x;
    const bool valid = n < a.N;

    MahaEngine<D, PADDED, pmc_use_mfma<D>()> engine;
    engine.load(a, blk * PMC_A_WAVES + (threadIdx.x >> 6), threadIdx.x & 63);

    int kb1 = 0, ke1 = a.K, kb2 = 0, ke2 = a.pack2 != nullptr ? a.K2 : 0;
    if (piece >= 0) {
        if (piece < a.split_s1) {
            kb1 = piece * a.split_c1;
            ke1 = kb1 + a.split_c1 < a.K ? kb1 + a.split_c1 : a.K;
            ke2 = 0;
        } else {
            ke1 = 0;
            kb2 = (piece - a.split_s1) * a.split_c2;
            ke2 = kb2 + a.split_c2 < a.K2 ? kb2 + a.split_c2 : a.K2;
        }
    }
    const ExpConst EC;
    RowPoison rowp;
    const bool keep_tile = a.atile != nullptr && ((n >> 6) << 6) < a.N;
    double m = a.max_init_zero ? 0.0 : -DBL_MAX, s = 0.0, mt = -DBL_MAX, st = 0.0;
    if (ke1 > kb1) {                                     // workgroup-uniform (the engines' barriers)
        cdouble *pk = (cdouble *)a.pack + (size_t)kb1 * dm.STRIDE;
        engine.begin(a.pack + (size_t)kb1 * dm.STRIDE, ke1 - kb1);
        for (int k = kb1; k < ke1; ++k, pk += dm.STRIDE) {
            const double maha = engine.eval(pk, k - kb1);
            double expo;
            const double v = component_value<D, KIND>(maha, pk + dm.DT, expo);
            if (a.individual != nullptr) {
                const long long col = ((cint64 *)pk)[dm.DT + 5];
                if (valid) a.individual[n * a.ld + col] = v;
            }
            if (keep_tile) a.atile[((size_t)(n >> 6) * a.K + k) * 64 + (threadIdx.x & 63)] = maha;
            lse_step(v, pk[dm.DT + 4], m, s, EC);
            rowp.see(v);
        }
    }
    const double poison1 = rowp.value();
    if (ke2 > kb2) {
        cdouble *pk = (cdouble *)a.pack2 + (size_t)kb2 * dm.STRIDE;
        engine.begin(a.pack2 + (size_t)kb2 * dm.STRIDE, ke2 - kb2);
        for (int k = kb2; k < ke2; ++k, pk += dm.STRIDE) {
            const double maha = engine.eval(pk, k - kb2);
            double expo;
            const double v = component_value<D, KIND2>(maha, pk + dm.DT, expo);
            lse_step(v, pk[dm.DT + 4], mt, st, EC);
            rowp.see(v);
        }
    }
    double lse, lse_target = 0.0;
    if (piece < 0) {
        lse = (log_any(s) + m) + poison1;                 // _regularize.pyx:81
        if (a.pack2 != nullptr) lse_target = (log_any(st) + mt) + rowp.value();
    } else {
        const bool second = piece >= a.split_s1;
        double *part = a.split_part + (size_t)(blk - a.split_b1) * S * (2 * PMC_A_WAVES * 64) + threadIdx.x;
        piece_store(part + (size_t)piece * (2 * PMC_A_WAVES * 64), second ? mt : m);
        piece_store(part + (size_t)piece * (2 * PMC_A_WAVES * 64) + PMC_A_WAVES * 64, (second ? st : s) + rowp.value());
        if (!piece_ticket_is_last(a.split_ticket + (blk - a.split_b1), S)) return;
        auto merge = [&](int p0, int p1) -> double {
            double M = -DBL_MAX;
            for (int p = p0; p < p1; ++p) M = max_f64(piece_load(part + (size_t)p * (2 * PMC_A_WAVES * 64)), M);
            double sum = 0.0;
            for (int p = p0; p < p1; ++p) {
                const double mp = piece_load(part + (size_t)p * (2 * PMC_A_WAVES * 64));
                const double sp = piece_load(part + (size_t)p * (2 * PMC_A_WAVES * 64) + PMC_A_WAVES * 64);
                sum = fma(sp, exp_clamped(max_f64(mp - M, -1075.0), EC), sum);
            }
            return log_any(sum) + M;
        };
        lse = merge(0, a.split_s1);
        if (a.pack2 != nullptr) {
            lse_target = merge(a.split_s1, S);
            if (lse != lse) lse_target = lse;             // (k_logpdf: the first mixture's NaN poisons both)
        }
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
    if (a.partials != nullptr) block_scalars_at<5>(sc, a.partials, blk);
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
// Pass 2 of k_resp walks the parked values of components khi-1 ... klo downwards, four loads in flight.
// (One loop over all components with `k < klds ? LDS : global` made the compiler select between the two
// pointers and issue a flat load followed by s_waitcnt 0 in every iteration: a full memory round trip
// per component and wavefront.)
template <class Load, class Step>
__device__ __forceinline__ void descend(int khi, int klo, Load load, Step step)
{
    int k = khi - 1;
    if (k - 3 >= klo) {
        // batches of four, the next batch's loads issued in front of the current batch's arithmetic
        double p0 = load(k), p1 = load(k - 1), p2 = load(k - 2), p3 = load(k - 3);
        for (; k - 7 >= klo; k -= 4) {
            const double q0 = load(k - 4), q1 = load(k - 5), q2 = load(k - 6), q3 = load(k - 7);
            step(k, p0);
            step(k - 1, p1);
            step(k - 2, p2);
            step(k - 3, p3);
            p0 = q0, p1 = q1, p2 = q2, p3 = q3;
        }
        step(k, p0);
        step(k - 1, p1);
        step(k - 2, p2);
        step(k - 3, p3);
        k -= 4;
    }
    for (; k >= klo; --k) step(k, load(k));
}

template <int D, bool PADDED, int KIND>
__global__ __launch_bounds__(PMC_A_WAVES * 64, pmc_min_waves(D, true)) void k_resp(const PmcArgsA a)
{
    const Dims<D> dm(a.dreal);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * PMC_A_WAVES +
                           __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n = tile * 64 + lane;
    const bool valid = n < a.N;
    const bool tile_live = tile * 64 < a.N;               // wave-uniform
    const int K = a.K;

    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    using Engine = MahaEngine<D, PADDED, pmc_use_mfma<D>()>;
    if (!tile_live) {
        Engine::idle(a.pack, K, PMC_RESIDENT_MAX_DIM_RESP);   // keep the workgroup's barriers / staging
    } else {
        Engine engine;
        engine.load(a, tile, lane);
        double *ut = a.u + (size_t)tile * K * 64 + lane;
        // parking place between the passes: LDS for the first klds components (the ones pass 2, which
        // walks downwards, would find evicted from L2), the output buffer itself for the rest
        const int klds = a.klds;
        double *pl = dyn_lds + Engine::LDS_DOUBLES + (size_t)(threadIdx.x >> 6) * klds * 64 + lane;
        double *mt = (KIND == PMC_KIND_STUDENT_T) ? a.scratch + (size_t)tile * K * 64 + lane : nullptr;
        double *vp = (KIND == PMC_KIND_STUDENT_T) ? a.vpartials + (size_t)tile * K * 2 : nullptr;

        // ---- pass 1: a_nk, parked, and the row maximum M (variational.pyx:730-735 / _regularize.pyx:73-77).
        // The soft-max is the reference's own two-pass form; the streaming log-sum-exp of k_logpdf would cost
        // ~16 more vector instructions per pair here (its selects on the running maximum, the rescaled bound
        // term, the product chain that undoes them in the normalisation pass).
        const bool literal = a.log_rho != nullptr;       // wave-uniform: the N x K matrix log_rho is wanted
        double M = a.max_init_zero ? 0.0 : -DBL_MAX;
        // exp_le0 below turns a NaN a_nk into e = 0.  The VB normalisation makes that NaN again (0 / 0); the PMC
        // kinds would get rho = 0 where the reference has NaN, so there the sample's weight carries it
        RowPoison rowp;
        const ExpConst EC;
        cdouble *pk = (cdouble *)a.pack;
        engine.begin(a.pack, K, PMC_RESIDENT_MAX_DIM_RESP);
        for (int k = 0; k < K; ++k, pk += dm.STRIDE) {
            const double maha = engine.eval(pk, k);
            double expo = 0.0;
            const double v = component_value<D, KIND>(maha, pk + dm.DT, expo);
            if constexpr (KIND == PMC_KIND_STUDENT_T) mt[(size_t)k * 64] = maha;
            if constexpr (KIND == PMC_KIND_VB) {
                if (a.exponent != nullptr) {
                    const long long col = ((cint64 *)pk)[dm.DT + 5];
                    if (valid) a.exponent[n * a.ld + col] = expo;
                }
            }
            M = max_f64(v, M);                            // (a NaN value leaves M alone, like `if v > M`)
            if constexpr (KIND != PMC_KIND_VB) rowp.see(v);
            if (k < klds) pl[k * 64] = v;                 // wave-uniform branch
            else ut[(size_t)k * 64] = v;
        }
        const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;
        const double swv = valid ? sw + rowp.value() : 0.0;    // (one select per sample instead of one per pair)
        auto parked_global = [&](int k) { return ut[(size_t)k * 64]; };
        auto parked_lds = [&](int k) { return pl[k * 64]; };

        // ---- pass 2: e = exp(a - M) [times the component weight], its sum, and -- unless the literal pass
        // below needs a_nk again -- e parked in place of a.  Components in DESCENDING order: the values
        // parked last are re-read first, while they are still in L2.
        double s = 0.0, tb = 0.0;
        auto expstep = [&](int k, double v) {
            const double lr = max_f64(v - M, -1075.0);    // variational.pyx:741 (below the clamp e = 0 either way)
            const double e = exp_clamped(lr, EC);         // :742 / _regularize.pyx:79
            if constexpr (KIND == PMC_KIND_VB) {
                tb = fma(e, lr, tb);                      // sum_k e_k (a_k - M): the dominant component adds exactly 0
                s += e;
            } else {
                s += ((cdouble *)a.pack + (size_t)k * dm.STRIDE)[dm.DT + 4] * e;
            }
            if (!literal) {                               // wave-uniform
                if (k < klds) pl[k * 64] = e;
                else ut[(size_t)k * 64] = e;
            }
        };
        descend(K, klds, parked_global, expstep);
        descend(klds, 0, parked_lds, expstep);

        // ---- pass 3: normalisation
        if constexpr (KIND == PMC_KIND_VB) {
            // variational.pyx:748-755: r = exp(log_rho - max) / norm, zeros -> tiny, log_rho += log(1/norm)
            const double norm_inv = 1. / s;
            const double log_norm_inv = log_any(norm_inv);
            double elq;
            if (literal) {
                elq = 0.0;
                pk = (cdouble *)a.pack + (size_t)(K - 1) * dm.STRIDE;
                for (int k = K - 1; k >= 0; --k, pk -= dm.STRIDE) {
                    double lr = (k < klds ? pl[k * 64] : ut[(size_t)k * 64]) - M;
                    double r = exp(lr);
                    r *= norm_inv;
                    if (r == 0.0) r = TINY;
                    lr += log_norm_inv;
                    elq += r * lr;                        // variational.pyx:1003-1013
                    ut[(size_t)k * 64] = swv * r;
                    const long long col = ((cint64 *)pk)[dm.DT + 5];
                    if (valid && a.r != nullptr) a.r[n * a.ld + col] = r;
                    if (valid) a.log_rho[n * a.ld + col] = lr;
                }
            } else {
                auto step = [&](int k, double e) {
                    const double r = zero_to_tiny(e * norm_inv);
                    ut[(size_t)k * 64] = swv * r;
                    if (a.r != nullptr) {
                        const long long col = ((cint64 *)((cdouble *)a.pack + (size_t)k * dm.STRIDE))[dm.DT + 5];
                        if (valid) a.r[n * a.ld + col] = r;
                    }
                };
                descend(K, klds, parked_global, step);
                descend(klds, 0, parked_lds, step);
                // sum_k r_k (a_k - M + log norm_inv) with sum_k r_k = 1   (variational.pyx:1003-1013)
                elq = fma(tb, norm_inv, log_norm_inv);
            }
            sc[0] = swv * elq;
        } else {
            // pmc.pyx:36-41: rho = exp(log q_k) * w_k / (exp(log_denominator) + tiny)
            // product form: exp(log q_k) = e_k exp(M), formed BEFORE anything else touches it, so where the
            // reference's exp(log q_k) underflows (log q_k < -708: denormal, then zero) this product does too.
            const double lse = log_any(s) + M;            // _regularize.pyx:81
            const double denom = exp(lse) + TINY;
            const double em = exp(M);
            const long long lat = (a.mode == PMC_RESP_PMC_LATENT && valid) ? a.latent[n] : -1;
            // rho -> u (and the public matrix, gamma and the dof sums of the Student-t update)
            auto emit = [&](int k, double rho, cdouble *c, long long col) {
                if (valid && a.r != nullptr) a.r[n * a.ld + col] = rho;
                const double wr = swv * rho;
                if constexpr (KIND == PMC_KIND_STUDENT_T) {
                    const double maha = mt[(size_t)k * 64];
                    const double nu = c[3];
                    const double gamma = (nu + (double)a.dreal) / (nu + maha);   // pmc.pyx:610
                    ut[(size_t)k * 64] = wr * gamma;
                    // per-wavefront sums of sample_w*rho and sample_w*rho*log(.5(maha+nu)), the
                    // N-sized parts of pmc.pyx:612 (alpha) and :669 (dof condition)
                    const double s1 = wave_sum(wr);
                    const double s2 = wave_sum(wr * log_pos(.5 * (maha + nu)));
                    if (lane == 0) {
                        vp[2 * k] = s1;
                        vp[2 * k + 1] = s2;
                    }
                } else {
                    ut[(size_t)k * 64] = wr;
                }
            };
            if (a.mode == PMC_RESP_PMC_LATENT || literal) {
                pk = (cdouble *)a.pack + (size_t)(K - 1) * dm.STRIDE;
                for (int k = K - 1; k >= 0; --k, pk -= dm.STRIDE) {
                    cdouble *c = pk + dm.DT;
                    const long long col = ((cint64 *)pk)[dm.DT + 5];
                    double rho;
                    if (a.mode == PMC_RESP_PMC_LATENT) {
                        rho = (lat == col) ? 1. : 0.;         // pmc.pyx:49-50
                    } else {
                        rho = exp(k < klds ? pl[k * 64] : ut[(size_t)k * 64]) * c[4];
                        rho /= denom;
                    }
                    emit(k, rho, c, col);
                }
            } else {
                // one division per sample, a multiplication per pair (a fp64 division is ~14 instructions)
                const double inv_denom = 1. / denom;
                auto step = [&](int k, double e) {
                    cdouble *pkk = (cdouble *)a.pack + (size_t)k * dm.STRIDE;
                    const double rho = ((e * em) * pkk[dm.DT + 4]) * inv_denom;
                    emit(k, rho, pkk + dm.DT, ((cint64 *)pkk)[dm.DT + 5]);
                };
                descend(K, klds, parked_global, step);
                descend(klds, 0, parked_lds, step);
            }
            sc[3] = swv * lse;
        }
    }
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}


// ---------------------------------------------------------------------------------------------
// k_resp_groups: the responsibilities of pmc_estep when the statistics follow as k_stats_gemm -- nothing parked in
// HBM, no normalisation pass.
//
// k_resp above needs a sample's row maximum and row sum before it can write a single u_nk, so every a_nk waits in a
// parking place: LDS for 19 components, the output buffer for the rest -- written, read, written (e), read, written
// (u): five transfers where one is algorithmic (6.0 against 4.2 GB per 1e7 samples at K = 32, 21 against 6.7 GB at
// K = 64, where it costs the clock: 1.83 instead of 1.97 GHz).  Here the components go in groups of 16 -- one row block
// of the statistics kernel's matrix product: a group is parked in LDS (8 KB per wavefront), gets its OWN maximum M_g
// and sum, and its u'_nk = exp(a_nk - M_g) [times the component weight for the PMC kind] leaves at once, written
// exactly once.  What is missing in u' is a factor per (sample, group),
//     VB:   f_ng = w_n exp(M_g - M) / s           PMC:   f_ng = w_n exp(M_g) / (exp(lse) + tiny),
// with M, s, lse combined across the groups as the streaming log-sum-exp does (one exp per GROUP, not per pair).
// The factors go into a.gscale (8 ceil(K / 16) bytes per sample) and k_stats_gemm multiplies its weight operand with
// them (one v_mul_f64 per 16 components x 4 samples).  E[log q(Z)] = sum_k r_k (a_k - M - log s) combines likewise:
// sum_k e_k (a_k - M) = sum_g c_g (tb_g + (M_g - M) s_g), c_g = exp(M_g - M).
// zero -> tiny of _update_r (variational.pyx:752) is applied to u' (a pair that underflows against its group's
// maximum); a pair that only underflows against the row maximum gives 0 where the reference has tiny = 2.2e-308.
// ---------------------------------------------------------------------------------------------
template <int D, bool PADDED, int KIND>
__global__ __launch_bounds__(PMC_A_WAVES * 64, pmc_min_waves(D, true)) void k_resp_groups(const PmcArgsA a)
{
    static_assert(KIND == PMC_KIND_VB || KIND == PMC_KIND_GAUSS, "k_resp_groups: VB and Gaussian Rao-Blackwell PMC");
    constexpr int GS = PMC_RESP_GROUP;
    if (a.blockflag != nullptr && (*a.redo == 0 || a.blockflag[blockIdx.x] == 0)) return;   // behind k_mgemm (see k_logpdf)
    const Dims<D> dm(a.dreal);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * PMC_A_WAVES +
                           __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n = tile * 64 + lane;
    const bool valid = n < a.N;
    const bool tile_live = tile * 64 < a.N;               // wave-uniform
    const int K = a.K, G = (K + GS - 1) / GS;

    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    using Engine = MahaEngine<D, PADDED, pmc_use_mfma<D>()>;
    if (!tile_live) {
        Engine::idle(a.pack, K, PMC_RESIDENT_MAX_DIM_RESP);   // keep the workgroup's barriers / staging
    } else {
        Engine engine;
        engine.load(a, tile, lane);
#ifdef PMC_AB_NO_U_STORE                                  // (A/B switch, timing only: the same instruction stream, but every
        double *ut = a.u + (size_t)(tile & 63) * K * 64 + lane;    //  workgroup's u lands in the same few tiles: L2 absorbs it)
        double *gs = a.gscale + (size_t)(tile & 63) * G * 64 + lane;
#else
        double *ut = a.u + (size_t)tile * K * 64 + lane;
        double *gs = a.gscale + (size_t)tile * G * 64 + lane;
#endif
        double *pl = dyn_lds + Engine::LDS_DOUBLES + (size_t)(threadIdx.x >> 6) * GS * 64 + lane;
        const ExpConst EC;
        RowPoison rowp;                                   // NaN if a component value of the row is NaN / +inf
        double Mrun = -DBL_MAX, srun = 0.0, tbrun = 0.0;
        cdouble *pk = (cdouble *)a.pack;
        engine.begin(a.pack, K, PMC_RESIDENT_MAX_DIM_RESP);
        for (int g = 0; g < G; ++g) {
            const int kb = g * GS, kn = (K - kb < GS) ? K - kb : GS;
            // pass 1 of the group: a_nk -> LDS, the group's maximum
            double Mg = -DBL_MAX;
#ifdef PMC_AB_ONLINE
            // A/B switch, TIMING ONLY (wrong numbers): the instruction stream of an "online" form that never parks a group --
            // u' = exp(a - M_ref) against a reference known BEFORE the group (here: the running maximum), written as soon as
            // a_nk exists, no second pass over LDS.  What such a form could gain at best (verdict r4 #6); a real one would add
            // the rescue of rows whose maximum jumps by more than the exponent range within a group -- which, for samples
            // drawn from the mixture itself, is every row in the group that holds its own component.
            double sg = 0.0, tbg = 0.0;
            const double Mref = Mrun;
            for (int j = 0; j < kn; ++j, pk += dm.STRIDE) {
                const double maha = engine.eval(pk, kb + j);
                double expo = 0.0;
                const double v = component_value<D, KIND>(maha, pk + dm.DT, expo);
                Mg = max_f64(v, Mg);
                rowp.see(v);
                const double lr = max_f64(v - Mref, -1075.0);
                const double e = exp_clamped(lr, EC);
                if constexpr (KIND == PMC_KIND_VB) {
                    tbg = fma(e, lr, tbg);
                    sg += e;
                    store_u(ut + (size_t)(kb + j) * 64, zero_to_tiny(e));
                } else {
                    const double we = (pk + dm.DT)[4] * e;
                    sg += we;
                    store_u(ut + (size_t)(kb + j) * 64, we);
                }
            }
            auto expstep = [&](int, double) {};
            if (false)
#else
            for (int j = 0; j < kn; ++j, pk += dm.STRIDE) {
                const double maha = engine.eval(pk, kb + j);
                double expo = 0.0;
                const double v = component_value<D, KIND>(maha, pk + dm.DT, expo);
                Mg = max_f64(v, Mg);
                rowp.see(v);
                pl[j * 64] = v;
            }
            // pass 2 of the group: u' = exp(a - M_g) [* w_k], written once
            double sg = 0.0, tbg = 0.0;
#endif
#ifndef PMC_AB_ONLINE
            auto expstep = [&](int j, double v) {
                const double lr = max_f64(v - Mg, -1075.0);
                const double e = exp_clamped(lr, EC);
                if constexpr (KIND == PMC_KIND_VB) {
                    tbg = fma(e, lr, tbg);
                    sg += e;
                    store_u(ut + (size_t)(kb + j) * 64, zero_to_tiny(e));
                } else {
                    const double we = ((cdouble *)a.pack + (size_t)(kb + j) * dm.STRIDE)[dm.DT + 4] * e;
                    sg += we;
                    store_u(ut + (size_t)(kb + j) * 64, we);
                }
            };
#endif
            descend(kn, 0, [&](int j) { return pl[j * 64]; }, expstep);
            // the group joins the row's running maximum / sum / bound term; its maximum waits in the factor's place
            const double Mn = max_f64(Mg, Mrun);
            const double cr = exp_clamped(max_f64(Mrun - Mn, -1075.0), EC), cg = exp_clamped(max_f64(Mg - Mn, -1075.0), EC);
            if constexpr (KIND == PMC_KIND_VB)
                tbrun = cr * fma(max_f64(Mrun - Mn, -1075.0), srun, tbrun) + cg * fma(max_f64(Mg - Mn, -1075.0), sg, tbg);
            srun = cr * srun + cg * sg;
            Mrun = Mn;
            gs[(size_t)g * 64] = Mg;
        }
        const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;
        const double swv = valid ? sw + rowp.value() : 0.0;
        if constexpr (KIND == PMC_KIND_VB) {
            // variational.pyx:748-755, :1003-1013
            const double norm_inv = 1. / srun;
            sc[0] = swv * fma(tbrun, norm_inv, log_any(norm_inv));
            const double f = swv * norm_inv;
            for (int g = 0; g < G; ++g)
                gs[(size_t)g * 64] = f * exp_clamped(max_f64(gs[(size_t)g * 64] - Mrun, -1075.0), EC);
        } else {
            // pmc.pyx:36-41; dead components' zeros take part in the row maximum of the reference (max_init_zero): the
            // log-sum-exp is the same number whichever maximum it is taken about
            const double lse = log_any(srun) + Mrun;      // _regularize.pyx:81
            const double f = swv / (exp(lse) + TINY);
            for (int g = 0; g < G; ++g) gs[(size_t)g * 64] = f * exp(gs[(size_t)g * 64]);
            sc[3] = swv * lse;
        }
    }
    if (a.partials != nullptr) block_scalars<5>(sc, a.partials);
}

// ---------------------------------------------------------------------------------------------
// k_resp_groups_split: k_resp_groups with the GROUPS of a sample block split over workgroups (round 6; the layout of
// the launch is k_logpdf_split's).  A group of 16 components is complete in itself -- its own maximum M_g, sum s_g and
// bound term, its u' written once -- so a piece takes split_c1 whole groups, leaves (M_g, s_g, tb_g) per sample and
// group in split_part, and the piece that draws the block's last ticket runs k_resp_groups' own recurrence over the
// groups in ascending order and writes the factors: the SAME operations in the SAME order on the same per-group
// numbers, i.e. the bits of k_resp_groups whatever the number of pieces (tested).
// ---------------------------------------------------------------------------------------------
// (the split form's few extra live values must not cost k_resp_groups<20>'s fourth wavefront per SIMD: 125-127 registers there)
__host__ __device__ constexpr int pmc_min_waves_split(int D) { return (D == 20 || D == 16) ? 4 : pmc_min_waves(D, true); }
template <int D, bool PADDED, int KIND>
__global__ __launch_bounds__(PMC_A_WAVES * 64, pmc_min_waves_split(D)) void k_resp_groups_split(const PmcArgsA a)
{
    static_assert(KIND == PMC_KIND_VB || KIND == PMC_KIND_GAUSS, "k_resp_groups: VB and Gaussian Rao-Blackwell PMC");
    constexpr int GS = PMC_RESP_GROUP, BT = PMC_A_WAVES * 64;
    const Dims<D> dm(a.dreal);
    const int lane = threadIdx.x & 63;
    long long blk = blockIdx.x;
    int piece = -1;                                       // workgroup-uniform
    if ((int)blockIdx.x >= a.split_b1) {
        const int p = (int)blockIdx.x - a.split_b1;
        blk = a.split_b1 + p / a.split_s1;
        piece = p % a.split_s1;
    }
    const long long tile = blk * PMC_A_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n = tile * 64 + lane;
    const bool valid = n < a.N;
    const bool tile_live = tile * 64 < a.N;               // wave-uniform
    const int K = a.K, G = (K + GS - 1) / GS;
    int g0 = 0, g1 = G;
    if (piece >= 0) {
        g0 = piece * a.split_c1;
        g1 = g0 + a.split_c1 < G ? g0 + a.split_c1 : G;
    }
    const int kfirst = g0 * GS, kcount = (g1 * GS < K ? g1 * GS : K) - kfirst;
    const double *mypack = a.pack + (size_t)kfirst * dm.STRIDE;
    // (pointers formed where they are used: the kernel sits at the 128-register boundary of four wavefronts per SIMD)
    auto part_of = [&](int g) { return a.split_part + ((size_t)(blk - a.split_b1) * G + g) * (3 * BT) + threadIdx.x; };
    auto gs_of = [&](int g) { return a.gscale + ((size_t)tile * G + g) * 64 + lane; };

    double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    using Engine = MahaEngine<D, PADDED, pmc_use_mfma<D>()>;
    const ExpConst EC;
    double Mrun = -DBL_MAX, srun = 0.0, tbrun = 0.0;
    RowPoison rowp;                                       // NaN if a component value of the row is NaN / +inf
    if (!tile_live) {
        Engine::idle(mypack, kcount, PMC_RESIDENT_MAX_DIM_RESP);   // keep the workgroup's barriers / staging
    } else {
        Engine engine;
        engine.load(a, tile, lane);
        double *ut = a.u + (size_t)tile * K * 64 + lane;
        double *pl = dyn_lds + Engine::LDS_DOUBLES + (size_t)(threadIdx.x >> 6) * GS * 64 + lane;
        cdouble *pk = (cdouble *)mypack;
        engine.begin(mypack, kcount, PMC_RESIDENT_MAX_DIM_RESP);
        for (int g = g0; g < g1; ++g) {
            const int kb = g * GS, kn = (K - kb < GS) ? K - kb : GS;
            // pass 1 of the group: a_nk -> LDS, the group's maximum
            double Mg = -DBL_MAX;
            for (int j = 0; j < kn; ++j, pk += dm.STRIDE) {
                const double maha = engine.eval(pk, kb + j - kfirst);
                double expo = 0.0;
                const double v = component_value<D, KIND>(maha, pk + dm.DT, expo);
                Mg = max_f64(v, Mg);
                rowp.see(v);
                pl[j * 64] = v;
            }
            // pass 2 of the group: u' = exp(a - M_g) [* w_k], written once
            double sg = 0.0, tbg = 0.0;
            auto expstep = [&](int j, double v) {
                const double lr = max_f64(v - Mg, -1075.0);
                const double e = exp_clamped(lr, EC);
                // (split_complete: another workgroup -- the one that finishes the block -- reads u' back: agent-scope stores)
                if constexpr (KIND == PMC_KIND_VB) {
                    tbg = fma(e, lr, tbg);
                    sg += e;
                    if (a.split_complete) piece_store(ut + (size_t)(kb + j) * 64, zero_to_tiny(e));
                    else store_u(ut + (size_t)(kb + j) * 64, zero_to_tiny(e));
                } else {
                    const double we = ((cdouble *)a.pack + (size_t)(kb + j) * dm.STRIDE)[dm.DT + 4] * e;
                    sg += we;
                    if (a.split_complete) piece_store(ut + (size_t)(kb + j) * 64, we);
                    else store_u(ut + (size_t)(kb + j) * 64, we);
                }
            };
            descend(kn, 0, [&](int j) { return pl[j * 64]; }, expstep);
            if (piece < 0) {
                // the group joins the row's running maximum / sum / bound term; its maximum waits in the factor's place
                const double Mn = max_f64(Mg, Mrun);
                const double cr = exp_clamped(max_f64(Mrun - Mn, -1075.0), EC), cg = exp_clamped(max_f64(Mg - Mn, -1075.0), EC);
                if constexpr (KIND == PMC_KIND_VB)
                    tbrun = cr * fma(max_f64(Mrun - Mn, -1075.0), srun, tbrun) + cg * fma(max_f64(Mg - Mn, -1075.0), sg, tbg);
                srun = cr * srun + cg * sg;
                Mrun = Mn;
                *gs_of(g) = Mg;
            } else {
                double *part = part_of(g);
                piece_store(part, Mg);
                piece_store(part + BT, sg + rowp.value());
                piece_store(part + 2 * BT, tbg);
            }
        }
    }
    double poison = rowp.value();
    if (piece >= 0) {
        if (!piece_ticket_is_last(a.split_ticket + (blk - a.split_b1), a.split_s1)) return;
        if (tile_live) {
            for (int g = 0; g < G; ++g) {
                const double *part = part_of(g);
                const double Mg = piece_load(part), sg = piece_load(part + BT), tbg = piece_load(part + 2 * BT);
                const double Mn = max_f64(Mg, Mrun);
                const double cr = exp_clamped(max_f64(Mrun - Mn, -1075.0), EC), cg = exp_clamped(max_f64(Mg - Mn, -1075.0), EC);
                if constexpr (KIND == PMC_KIND_VB)
                    tbrun = cr * fma(max_f64(Mrun - Mn, -1075.0), srun, tbrun) + cg * fma(max_f64(Mg - Mn, -1075.0), sg, tbg);
                srun = cr * srun + cg * sg;
                Mrun = Mn;
                *gs_of(g) = Mg;
            }
            poison = (srun != srun) ? srun : 0.0;          // (a piece's NaN rides on its sums)
        }
    }
    if (tile_live) {
        const double sw = (a.sample_w != nullptr && valid) ? a.sample_w[n] : 1.0;
        const double swv = valid ? sw + poison : 0.0;
        if constexpr (KIND == PMC_KIND_VB) {
            // variational.pyx:748-755, :1003-1013
            const double norm_inv = 1. / srun;
            sc[0] = swv * fma(tbrun, norm_inv, log_any(norm_inv));
            const double f = swv * norm_inv;
            double *gs = gs_of(0);
            for (int g = 0; g < G; ++g)
                gs[(size_t)g * 64] = f * exp_clamped(max_f64(gs[(size_t)g * 64] - Mrun, -1075.0), EC);
        } else {
            const double lse = log_any(srun) + Mrun;      // _regularize.pyx:81
            const double f = swv / (exp(lse) + TINY);
            double *gs = gs_of(0);
            for (int g = 0; g < G; ++g) gs[(size_t)g * 64] = f * exp(gs[(size_t)g * 64]);
            sc[3] = swv * lse;
        }
    }
    if (a.split_complete && tile_live) {
        // small batches: the statistics that follow run per component (k_stats) and take u as it stands -- the finishing
        // workgroup multiplies every group's factor into its values (written by whichever piece walked the group)
        double *ut = a.u + (size_t)tile * K * 64 + lane;
        const double *gs = gs_of(0);
        for (int g = 0; g < G; ++g) {
            const double fg = gs[(size_t)g * 64];
            const int kb = g * GS, kn = (K - kb < GS) ? K - kb : GS;
            for (int j = 0; j < kn; ++j) {
                double *p = ut + (size_t)(kb + j) * 64;
                *p = piece_load(p) * fg;
            }
        }
    }
    if (a.partials != nullptr) block_scalars_at<5>(sc, a.partials, blk);
}

template <int KIND, int KIND2> hipError_t launch_logpdf_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    constexpr size_t lds = sizeof(double) * MahaEngine<D_, P_, pmc_use_mfma<D_>()>::LDS_DOUBLES;
    hipLaunchKernelGGL((k_logpdf<D_, P_, KIND, KIND2>), dim3(grid), dim3(PMC_A_WAVES * 64), lds, st, a);
    return hipGetLastError();
}
template <int KIND, int KIND2> hipError_t launch_logpdf_split_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
#if PMC_D == 0
    return hipErrorNotSupported;
#else
    constexpr size_t lds = sizeof(double) * MahaEngine<D_, P_, pmc_use_mfma<D_>()>::LDS_DOUBLES;
    hipLaunchKernelGGL((k_logpdf_split<D_, P_, KIND, KIND2>), dim3(grid), dim3(PMC_A_WAVES * 64), lds, st, a);
    return hipGetLastError();
#endif
}
template <int KIND> hipError_t launch_resp_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    const size_t lds = sizeof(double) * (MahaEngine<D_, P_, pmc_use_mfma<D_>()>::LDS_DOUBLES +
                                         (size_t)PMC_A_WAVES * a.klds * 64);
    if (lds > 65536) {
        const hipError_t once = PMC_SET_LDS_PER_DEVICE(
            (&k_resp<D_, P_, KIND>), sizeof(double) * (MahaEngine<D_, P_, pmc_use_mfma<D_>()>::LDS_DOUBLES +
                                                       (size_t)PMC_A_WAVES * pmc_resp_klds(D_) * 64));
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((k_resp<D_, P_, KIND>), dim3(grid), dim3(PMC_A_WAVES * 64), lds, st, a);
    return hipGetLastError();
}

template <int KIND> hipError_t launch_resp_groups_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    constexpr size_t lds = sizeof(double) * (MahaEngine<D_, P_, pmc_use_mfma<D_>()>::LDS_DOUBLES +
                                             (size_t)PMC_A_WAVES * PMC_RESP_GROUP * 64);
    if constexpr (lds > 65536) {
        const hipError_t once = PMC_SET_LDS_PER_DEVICE((&k_resp_groups<D_, P_, KIND>), lds);
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((k_resp_groups<D_, P_, KIND>), dim3(grid), dim3(PMC_A_WAVES * 64), lds, st, a);
    return hipGetLastError();
}

template <int KIND> hipError_t launch_resp_groups_split_k(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
    constexpr size_t lds = sizeof(double) * (MahaEngine<D_, P_, pmc_use_mfma<D_>()>::LDS_DOUBLES +
                                             (size_t)PMC_A_WAVES * PMC_RESP_GROUP * 64);
    if constexpr (lds > 65536) {
        const hipError_t once = PMC_SET_LDS_PER_DEVICE((&k_resp_groups_split<D_, P_, KIND>), lds);
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((k_resp_groups_split<D_, P_, KIND>), dim3(grid), dim3(PMC_A_WAVES * 64), lds, st, a);
    return hipGetLastError();
}

}  // namespace

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_resp_groups_split_d, PMC_D, PMC_PADDED)(int kind, const PmcArgsA &a,
                                                                                        unsigned grid, hipStream_t st)
{
#if PMC_D == 0
    return hipErrorNotSupported;
#else
    switch (kind) {
    case PMC_KIND_GAUSS: return launch_resp_groups_split_k<PMC_KIND_GAUSS>(a, grid, st);
    case PMC_KIND_VB: return launch_resp_groups_split_k<PMC_KIND_VB>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
#endif
}

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_resp_groups_d, PMC_D, PMC_PADDED)(int kind, const PmcArgsA &a,
                                                                                  unsigned grid, hipStream_t st)
{
#if PMC_D == 0
    return hipErrorNotSupported;
#else
    switch (kind) {
    case PMC_KIND_GAUSS: return launch_resp_groups_k<PMC_KIND_GAUSS>(a, grid, st);
    case PMC_KIND_VB: return launch_resp_groups_k<PMC_KIND_VB>(a, grid, st);
    default: return hipErrorInvalidValue;
    }
#endif
}

// kind2: component family of the second (target) mixture of pmc_importance_weights; ignored without one
extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_logpdf_d, PMC_D, PMC_PADDED)(int kind, int kind2, const PmcArgsA &a,
                                                                             unsigned grid, hipStream_t st)
{
    if (a.pack2 == nullptr) kind2 = kind;
    if (kind == PMC_KIND_GAUSS && kind2 == PMC_KIND_GAUSS) return launch_logpdf_k<PMC_KIND_GAUSS, PMC_KIND_GAUSS>(a, grid, st);
    if (kind == PMC_KIND_STUDENT_T && kind2 == PMC_KIND_STUDENT_T)
        return launch_logpdf_k<PMC_KIND_STUDENT_T, PMC_KIND_STUDENT_T>(a, grid, st);
    if (kind == PMC_KIND_GAUSS && kind2 == PMC_KIND_STUDENT_T)
        return launch_logpdf_k<PMC_KIND_GAUSS, PMC_KIND_STUDENT_T>(a, grid, st);
    if (kind == PMC_KIND_STUDENT_T && kind2 == PMC_KIND_GAUSS)
        return launch_logpdf_k<PMC_KIND_STUDENT_T, PMC_KIND_GAUSS>(a, grid, st);
    return hipErrorInvalidValue;
}

// (A/B, -DPMC_TWO_PER_LANE: Gaussian mixtures, the plain pass only; grid = blocks of 512 samples)
extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_logpdf2_d, PMC_D, PMC_PADDED)(const PmcArgsA &a, unsigned grid, hipStream_t st)
{
#if defined(PMC_TWO_PER_LANE) && PMC_D >= 8 && PMC_D <= 30
    hipLaunchKernelGGL((k_logpdf2<D_, P_, PMC_KIND_GAUSS, PMC_KIND_GAUSS>), dim3(grid), dim3(PMC_A_WAVES * 64), 0, st, a);
    return hipGetLastError();
#else
    return hipErrorNotSupported;
#endif
}

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_logpdf_split_d, PMC_D, PMC_PADDED)(int kind, int kind2, const PmcArgsA &a,
                                                                                   unsigned grid, hipStream_t st)
{
    if (a.pack2 == nullptr) kind2 = kind;
    if (kind == PMC_KIND_GAUSS && kind2 == PMC_KIND_GAUSS) return launch_logpdf_split_k<PMC_KIND_GAUSS, PMC_KIND_GAUSS>(a, grid, st);
    if (kind == PMC_KIND_STUDENT_T && kind2 == PMC_KIND_STUDENT_T)
        return launch_logpdf_split_k<PMC_KIND_STUDENT_T, PMC_KIND_STUDENT_T>(a, grid, st);
    if (kind == PMC_KIND_GAUSS && kind2 == PMC_KIND_STUDENT_T)
        return launch_logpdf_split_k<PMC_KIND_GAUSS, PMC_KIND_STUDENT_T>(a, grid, st);
    if (kind == PMC_KIND_STUDENT_T && kind2 == PMC_KIND_GAUSS)
        return launch_logpdf_split_k<PMC_KIND_STUDENT_T, PMC_KIND_GAUSS>(a, grid, st);
    return hipErrorInvalidValue;
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
