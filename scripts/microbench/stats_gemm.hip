// Prototype + calibration harness of the component x monomial statistics kernel (pmc_stats.hip, k_stats_gemm):
//   partial[k][m] = sum_n u[n][k] * z[n][m],   z[n] = (1 | d | d_i d_j, j <= i),  d = x_n - c  (one common shift)
// on v_mfma_f64_16x16x4_f64: A = u (16 components x 4 samples), B = z (4 samples x 16 monomials).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off stats_gemm.hip -o stats_gemm
//   ./stats_gemm [N] [K] [reps]            (D and the tiling are compile-time: -DSG_D=20 -DSG_C=5 ...)
// Checks the kernel against a host loop on every run and prints ms and algorithmic TFLOP/s.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef SG_D
#define SG_D 20
#endif
#ifndef SG_C
#define SG_C 5          // column tiles (16 monomials each) per wavefront
#endif
#ifndef SG_CGW
#define SG_CGW 3        // column groups (wavefronts side by side) per workgroup
#endif
#ifndef SG_SL
#define SG_SL 4         // sample slices: wavefronts that share a column group and split a tile's samples
#endif
#ifndef SG_NS
#define SG_NS 2         // tiles per pipeline step
#endif

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <int I> using ic = std::integral_constant<int, I>;
template <int B, int E, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

struct GArgs {
    const double *x;
    long long N;
    int dreal;
    const double *center;   // dreal doubles
    int K;
    const double *u;        // tile-major ntiles x K x 64
    double *partials;       // [nchunks * SL][K][MSP]
    long long ntiles;
    int nchunks, tiles_per_chunk, ngroups, ncs;
};

template <int D> struct GemmShape {
    static constexpr int NP = (D + 1) / 2;                 // coordinate pairs per sample
    static constexpr int PITCH = (NP + 1) | 1;             // 16-byte slots per LDS row, odd, one spare for the "1"
    static constexpr int ROWD = 2 * PITCH;
    static constexpr int M = (D + 1) * (D + 2) / 2;        // monomials 1 | d | d d^T lower triangle
    static constexpr int NT = (M + 15) / 16;
    static constexpr int MSP = NT * 16;
    static constexpr int XT = 64 * ROWD;                   // doubles per x tile
    static constexpr int UPIECE = 130;                     // doubles per 1-KiB DMA piece (2 components) + 16 B pad
    static constexpr int UT = 16 * UPIECE;                 // doubles per u tile (32 components)
};

__device__ __forceinline__ void dma_barrier()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// row (within a tile, before the lane's 8 g) of MFMA step j of a slice, compile-time part
template <int SL> __host__ __device__ constexpr int ro_imm(int j) { return SL == 1 ? (j & 7) + 32 * (j >> 3) : j; }
// ... and the slice's run-time part
template <int SL> __device__ __forceinline__ int ro_base(int sl)
{
    if constexpr (SL == 1) return 0;
    else if constexpr (SL == 2) return 32 * sl;
    else if constexpr (SL == 4) return 4 * (sl & 1) + 32 * (sl >> 1);
    else if constexpr (SL == 8) return 2 * (sl & 3) + 32 * (sl >> 2);
    else return (sl & 7) + 32 * (sl >> 3);
}

template <int D, bool PADDED, int C, int CGW, int SL, int NS>
__global__ __launch_bounds__(64 * CGW * SL) void k_stats_gemm(const GArgs b)
{
    using SH = GemmShape<D>;
    constexpr int W = CGW * SL, R = 2;
    constexpr int NP = SH::NP, ROWD = SH::ROWD, XT = SH::XT, UT = SH::UT, UPIECE = SH::UPIECE;
    constexpr int BUFX = NS * XT, BUFU = NS * UT;
    constexpr int JN = 16 / SL, NSTEP = NS * JN;
    constexpr int PX = NS * 64 * NP, NPX = (PX + W * 64 - 1) / (W * 64);
    constexpr int PU = NS * 16, NPU = (PU + W - 1) / W;
    extern __shared__ double lds[];                        // 2 x buffers, then 2 u buffers
    double *xs = lds, *us = lds + 2 * BUFX;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % CGW, sl = wave / CGW;
    const int n16 = lane & 15, g = lane >> 4;
    const int dreal = PADDED ? b.dreal : D;
    const int npr = (dreal + 1) / 2, ONE = 2 * npr;
    const int M = (dreal + 1) * (dreal + 2) / 2;
    const long long total = b.N * (long long)dreal;

    // block -> (chunk, component group, column super group); all blocks of a chunk on one XCD
    const int bid = blockIdx.x, qb = bid >> 3;
    const int nsub = b.ngroups * b.ncs;
    const int chunk = (bid & 7) + 8 * (qb / nsub);
    const int sub = qb % nsub, group = sub / b.ncs, cs = sub % b.ncs;
    const int kmin = group * 32;
    const int nrb = (b.K - kmin) > 16 ? 2 : 1;
    const long long t0 = (long long)chunk * b.tiles_per_chunk;
    long long t1 = t0 + b.tiles_per_chunk;
    if (t1 > b.ntiles) t1 = b.ntiles;

    // this lane's two factors of each of the wavefront's column tiles (LDS double offsets incl. the lane's rows)
    const int rowbase = (8 * g + ro_base<SL>(sl)) * ROWD;
    int off1[C], off2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int ct = (cs * CGW + cg) * C + c;
        int m = 16 * ct + n16;
        int i1 = ONE, i2 = ONE;
        if (m >= 1 && m <= dreal) i2 = m - 1;
        else if (m > dreal && m < M) {
            const int t = m - 1 - dreal;
            int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while ((i + 1) * (i + 2) / 2 <= t) ++i;
            while (i * (i + 1) / 2 > t) --i;
            i1 = i;
            i2 = t - i * (i + 1) / 2;
        }
        off1[c] = rowbase + i1;
        off2[c] = rowbase + i2;
    }
    int uoff[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int cc = 16 * r + n16;
        uoff[r] = (cc >> 1) * UPIECE + (cc & 1) * 64 + 8 * g + ro_base<SL>(sl);
    }

    // the "1" of every row, both buffers (never overwritten: the staging writes data slots only)
    for (int row = tid; row < 2 * NS * 64; row += 64 * W) xs[row * ROWD + ONE] = 1.0;

    // x staging: piece -> (row in step, pair), fixed per thread
    int xl[NPX];                                           // LDS double offset (or -1: no piece)
    unsigned xg[NPX];                                      // global double offset relative to the step's first row
    double c0[NPX], c1[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int id = tid + i * W * 64;
        const int n = id / NP, jp = id % NP;
        const bool ok = id < PX && jp < npr;
        xl[i] = ok ? n * ROWD + 2 * jp : -1;
        xg[i] = (unsigned)(n * dreal + 2 * jp);
        c0[i] = ok ? b.center[2 * jp] : 0.0;
        c1[i] = (ok && 2 * jp + 1 < dreal) ? b.center[2 * jp + 1] : 0.0;
    }
    d2 xv[NPX];
    auto xload = [&](long long t) {
        const long long tt = t < t1 ? t : t0;
        const long long base = tt * 64 * dreal;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            long long o = base + xg[i];
            if (o > total - 2) o = total - 2;
            xv[i] = *(const d2u *)(b.x + o);
        }
    };
    auto xstore = [&](long long t, double *xbuf) {
        const long long tt = t < t1 ? t : t0;
        const long long base = tt * 64 * dreal;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            if (xl[i] >= 0) {
                const bool last = base + xg[i] == total - 1;   // odd tail: the pair was fetched one element early
                d2 v;
                v[0] = (last ? xv[i][1] : xv[i][0]) - c0[i];
                v[1] = xv[i][1] - c1[i];
                *(d2 *)(xbuf + xl[i]) = v;
            }
        }
    };
    const long long ulen = b.ntiles * (long long)b.K * 64;
    auto udma = [&](long long t, double *ubuf) {
#pragma unroll
        for (int i = 0; i < NPU; ++i) {
            const int id = wave + i * W;
            if (PU % W == 0 || id < PU) {
                const int q = id / 16, p = id % 16;
                const long long tile = (t + q < t1) ? t + q : t0;
                long long o = (tile * b.K + kmin + 2 * p) * 64 + 2 * lane;
                if (o > ulen - 2) o = ulen - 2;
                __builtin_amdgcn_global_load_lds((gvoid_t *)(b.u + o), (lvoid_t *)(ubuf + q * UT + p * UPIECE), 16, 0, 0);
            }
        }
    };

    d4 acc[R][C];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[r][c] = d4{0.0, 0.0, 0.0, 0.0};

    xload(t0);
    udma(t0, us);
    xstore(t0, xs);
    dma_barrier();
    int buf = 0;
    for (long long t = t0; t < t1; t += NS, buf ^= 1) {
        const double *xb = xs + buf * BUFX;
        const double *ub = us + buf * BUFU;
        xload(t + NS);
        udma(t + NS, us + (buf ^ 1) * BUFU);

        double ac[R], zc[C], an[R], f1n[C], f2n[C];
        auto fetch = [&](auto IDX, double (&a)[R], double (&f1)[C], double (&f2)[C]) {
            constexpr int idx = decltype(IDX)::value, q = idx / JN, j = idx % JN;
            constexpr int XIMM = (q * 64 + ro_imm<SL>(j)) * ROWD, UIMM = q * UT + ro_imm<SL>(j);
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = ub[uoff[r] + UIMM];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                f1[c] = xb[off1[c] + XIMM];
                f2[c] = xb[off2[c] + XIMM];
            }
        };
        fetch(ic<0>{}, ac, f1n, f2n);
#pragma unroll
        for (int c = 0; c < C; ++c) zc[c] = f1n[c] * f2n[c];
        static_for<0, NSTEP>([&](auto IDX) {
            constexpr int idx = decltype(IDX)::value, q = idx / JN;
            if constexpr (idx + 1 < NSTEP) fetch(ic<idx + 1>{}, an, f1n, f2n);
            const bool live = t + q < t1;
            double a0 = live ? ac[0] : 0.0, a1 = live ? ac[1] : 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[0][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, zc[c], acc[0][c], 0, 0, 0);
            if (nrb > 1) {
#pragma unroll
                for (int c = 0; c < C; ++c) acc[1][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, zc[c], acc[1][c], 0, 0, 0);
            }
            if constexpr (idx + 1 < NSTEP) {
#pragma unroll
                for (int r = 0; r < R; ++r) ac[r] = an[r];
#pragma unroll
                for (int c = 0; c < C; ++c) zc[c] = f1n[c] * f2n[c];
            }
        });
        xstore(t + NS, xs + (buf ^ 1) * BUFX);
        dma_barrier();
    }

    // D[row = (lane >> 4) + 4 reg][col = lane & 15]
    double *out = b.partials + ((size_t)(chunk * SL + sl) * b.K) * SH::MSP;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int ct = (cs * CGW + cg) * C + c;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int k = kmin + 16 * r + g + 4 * reg;
                if (k < b.K && ct < SH::NT) out[(size_t)k * SH::MSP + 16 * ct + n16] = acc[r][c][reg];
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv)
{
    constexpr int D = SG_D, C = SG_C, CGW = SG_CGW, SL = SG_SL, NS = SG_NS;
    using SH = GemmShape<D>;
    const long long N = argc > 1 ? atoll(argv[1]) : 4000000;
    const int K = argc > 2 ? atoi(argv[2]) : 32;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const long long ntiles = (N + 63) / 64;
    const int ngroups = (K + 31) / 32, ncs = (SH::NT + C * CGW - 1) / (C * CGW);
    int nchunks = argc > 4 ? atoi(argv[4]) : 256 / (ngroups * ncs);
    if (nchunks < 8) nchunks = 8;
    nchunks = (nchunks + 7) / 8 * 8;
    if (nchunks > ntiles) nchunks = (int)((ntiles + 7) / 8 * 8);
    const int tpc = (int)((ntiles + nchunks - 1) / nchunks);
    printf("D=%d K=%d N=%lld  C=%d CGW=%d SL=%d NS=%d  NT=%d ncs=%d ngroups=%d nchunks=%d tiles/chunk=%d\n", D, K, N, C, CGW,
           SL, NS, SH::NT, ncs, ngroups, nchunks, tpc);

    std::vector<double> hx((size_t)N * D), hu((size_t)ntiles * K * 64, 0.0), hc(D);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0 / 9007199254740992.0); };
    for (auto &v : hx) v = 4.0 * rnd() - 2.0;
    for (int j = 0; j < D; ++j) hc[j] = 0.3 * j - 1.0;
    for (long long n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) hu[((n >> 6) * K + k) * 64 + (n & 63)] = rnd() < 0.3 ? rnd() : 0.0;

    double *dx, *du, *dc, *dp;
    const size_t npart = (size_t)nchunks * SL * K * SH::MSP;
    CK(hipMalloc(&dx, hx.size() * 8));
    CK(hipMalloc(&du, hu.size() * 8));
    CK(hipMalloc(&dc, D * 8));
    CK(hipMalloc(&dp, npart * 8));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(du, hu.data(), hu.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, hc.data(), D * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dp, 0xff, npart * 8));

    GArgs a{dx, N, D, dc, K, du, dp, ntiles, nchunks, tpc, ngroups, ncs};
    auto kern = k_stats_gemm<D, false, C, CGW, SL, NS>;
    const size_t ldsb = sizeof(double) * 2 * NS * (SH::XT + SH::UT);
    printf("LDS %zu bytes, %d wavefronts per workgroup\n", ldsb, CGW * SL);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const unsigned grid = (unsigned)(nchunks * ngroups * ncs);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * CGW * SL), ldsb, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double flops = (double)N * K * (1.0 + 2.0 * D + (double)D * (D + 1));
    printf("k_stats_gemm: %.3f ms   %.1f algorithmic TFLOP/s (%.3f of 78.6)\n", best, flops / best * 1e-9, flops / best * 1e-9 / 78.6);

    // check against the host on a subset of the components (all monomials)
    std::vector<double> hp(npart);
    CK(hipMemcpy(hp.data(), dp, npart * 8, hipMemcpyDeviceToHost));
    const int M = SH::M;
    double worst = 0.0;
    const long long ncheck = N < 300000 ? N : 300000;       // host loop over a prefix only if N is large: then compare chunk 0.. partial sums
    (void)ncheck;
    for (int k : {0, K / 2, K - 1}) {
        std::vector<double> ref(M, 0.0), got(M, 0.0), mag(M, 0.0);
        for (long long n = 0; n < N; ++n) {
            const double uu = hu[((n >> 6) * K + k) * 64 + (n & 63)];
            if (uu == 0.0) continue;
            double d[D];
            for (int j = 0; j < D; ++j) d[j] = hx[n * D + j] - hc[j];
            ref[0] += uu;
            mag[0] += uu;
            for (int j = 0; j < D; ++j) { ref[1 + j] += uu * d[j]; mag[1 + j] += fabs(uu * d[j]); }
            int m = 1 + D;
            for (int i = 0; i < D; ++i)
                for (int j = 0; j <= i; ++j, ++m) { ref[m] += uu * (d[i] * d[j]); mag[m] += fabs(uu * d[i] * d[j]); }
        }
        for (int ce = 0; ce < nchunks * SL; ++ce)
            for (int m = 0; m < M; ++m) got[m] += hp[((size_t)ce * K + k) * SH::MSP + m];
        for (int m = 0; m < M; ++m) {
            const double err = fabs(got[m] - ref[m]) / (mag[m] > 0 ? mag[m] : 1.0);
            if (!(err <= worst)) worst = err;
        }
    }
    printf("max |got - ref| / sum|terms| over 3 components x %d monomials: %.3e  %s\n", M, worst, worst < 1e-13 ? "OK" : "MISMATCH");
    return worst < 1e-13 ? 0 : 1;
}
