// (round 3, second prototype: MG_NCT = 1, 2 or 4 component tiles share every monomial product z -- one v_mul_f64 per
//  MG_NCT matrix instructions instead of one per instruction; everything else as in the first prototype)
// Prototype of the Mahalanobis forms of ALL components as one matrix product (the "quadratic form as monomials" engine):
//   maha[n][k] = sum_m theta[k][m] z[n][m],   z[n] = (d_i d_j, i <= j | d_i | 1),  d = x_n - c  (one common centre)
// on v_mfma_f64_16x16x4_f64: A = theta (16 components x 4 monomials), B = z (4 monomials x 16 samples).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off maha_gemm.hip -o maha_gemm -DMG_D=40
//   ./maha_gemm [N] [K] [reps]
// The four monomials of a step belong to the four lane groups g = lane >> 4; group g sees the coordinates rotated by
// g D / 4 (pi_g(r) = (r + g D / 4) mod D), so that ONE compile-time pair of rows (a, b) per step gives four different
// monomials d_pi(a) d_pi(b): the pairs (a, a + delta), a < D / 4, delta = 0 ... D / 2 cover every unordered pair once
// (delta = D / 2: twice -- groups 2, 3 get coefficient 0).  Then D / 4 steps for the linear terms, one for the constant.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#ifndef MG_D
#define MG_D 40
#endif
#ifndef MG_CH
#define MG_CH 16        // steps per staged chunk of theta
#endif
#ifndef MG_NCT
#define MG_NCT 2        // component tiles (of 16) per pass over the monomials
#endif

typedef double d4 __attribute__((ext_vector_type(4)));
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

template <int D> struct Cfg {
    static_assert(D % 4 == 0, "whole lane groups");
    static constexpr int Q = D / 4;
    static constexpr int ND = 2 * Q + 1;                   // deltas per a
    static constexpr int NQ = Q * ND;                      // quadratic steps
    static constexpr int NSTEP = NQ + Q + 1;
    static constexpr int CH = MG_CH;
    static constexpr int NCH = (NSTEP + CH - 1) / CH;
    static constexpr int NSTEPP = NCH * CH;
    // row stride of the LDS image of d (doubles): >= 64 and Q RS = 16 mod 32, so that the two lane groups of a
    // half-wavefront read 32 banks apart
    static constexpr int RS = D == 20 ? 80 : D == 24 ? 72 : D == 32 ? 66 : D == 40 ? 72 : D == 48 ? 68 : D == 16 ? 68 : 0;
    static_assert(RS >= 64 && (Q * RS) % 32 == 16, "bank spread");
    static_assert(NCH % 2 == 0, "buffer parity per tile");
};

template <int IMM> __device__ __forceinline__ void lds_read64(double &v, unsigned addr)
{
    static_assert(IMM >= 0 && IMM < 65536 && IMM % 8 == 0, "ds_read_b64 offset");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
}
__device__ __forceinline__ void lds_wait5(double &a, double &b, double &c, double &d, double &e)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}
__device__ __forceinline__ void lds_wait4(double &a, double &b, double &c, double &d)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lds_wait1(double &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }

template <int D>
__global__ __launch_bounds__(256) void k_maha(const double *__restrict__ x, long long N, const double *__restrict__ cen,
                                             const double *__restrict__ img, int KT, double *__restrict__ out)
{
    using C = Cfg<D>;
    constexpr int Q = C::Q, RS = C::RS, CH = C::CH, NCH = C::NCH, ND = C::ND, NQ = C::NQ, NSTEP = C::NSTEP;
    extern __shared__ double lds[];
    constexpr int NCT = MG_NCT;
    double *th = lds;                                      // [2][NCT][CH * 64]
    double *dl = lds + 2 * NCT * CH * 64;                  // [4][D][RS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const long long tile = blockIdx.x * 4LL + wave;
    double *dw = dl + wave * D * RS;
    {
        long long n = tile * 64 + lane;
        if (n >= N) n = N - 1;
        for (int j = 0; j < D; ++j) dw[j * RS + lane] = x[n * D + j] - cen[j];
    }
    const unsigned dwa = (unsigned)(uintptr_t)(lvoid_t *)dw, tha = (unsigned)(uintptr_t)(lvoid_t *)th + 8u * lane;
    unsigned base[4];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) base[jq] = dwa + 8u * (unsigned)((((jq + g) * Q) % D) * RS + s16);

    auto stage = [&](int cg) {                             // chunk cg (pass cg / NCH, chunk cg % NCH) of the NCT images -> buffer cg & 1
        const int pass = cg / NCH, ch = cg - pass * NCH;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const double *src = img + ((size_t)(pass * NCT + c) * NCH + ch) * CH * 64;
            double *dst = th + ((cg & 1) * NCT + c) * CH * 64;
#pragma unroll
            for (int p = 0; p < (CH * 64 / 128 + 3) / 4; ++p) {
                const int piece = wave + 4 * p;
                if (piece < CH * 64 / 128)
                    __builtin_amdgcn_global_load_lds((gvoid_t *)(src + piece * 128 + 2 * lane), (lvoid_t *)(dst + piece * 128), 16, 0, 0);
            }
        }
    };
    const int npass = KT / NCT;
    const int nchunks = npass * NCH;
    __syncthreads();
    stage(0);
    for (int kt = 0; kt < npass; ++kt) {
        d4 acc[NCT][4];
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[c][t] = d4{0.0, 0.0, 0.0, 0.0};
        double ar[2][4], bv[2][4], tv[NCT][3], z[2][4];
        // three stages, one step apart: fetch (LDS reads) -> mul (the monomials) -> the matrix instructions
        auto fetch = [&](auto S_) {
            constexpr int s = decltype(S_)::value, set = s & 1;
            constexpr int ch = s / CH, i = s % CH;
            static_for<0, NCT>([&](auto C_) {
                constexpr int c = decltype(C_)::value;
                constexpr int TH = 8 * ((((ch & 1) * NCT + c) * CH + i) * 64);
                lds_read64<TH>(tv[c][s % 3], tha);
            });
            if constexpr (s < NQ) {
                constexpr int a = s / ND, dlt = s % ND, b = a + dlt, jq = b / Q, br = b - jq * Q;
                static_for<0, 4>([&](auto T_) {
                    constexpr int t = decltype(T_)::value;
                    if constexpr (dlt == 0) lds_read64<8 * (a * RS + 16 * t)>(ar[a & 1][t], base[0]);
                    else lds_read64<8 * (br * RS + 16 * t)>(bv[set][t], base[jq]);
                });
            } else if constexpr (s < NQ + Q) {
                constexpr int r = s - NQ;
                static_for<0, 4>([&](auto T_) {
                    constexpr int t = decltype(T_)::value;
                    lds_read64<8 * (r * RS + 16 * t)>(bv[set][t], base[0]);
                });
            }
        };
        auto arrive = [&](auto S_) {
            constexpr int s = decltype(S_)::value, set = s & 1;
            if constexpr (s < NQ && s % ND == 0) {
                constexpr int a = s / ND;
                lds_wait5(tv[0][s % 3], ar[a & 1][0], ar[a & 1][1], ar[a & 1][2], ar[a & 1][3]);
            } else if constexpr (s < NQ + Q) lds_wait5(tv[0][s % 3], bv[set][0], bv[set][1], bv[set][2], bv[set][3]);
            else lds_wait1(tv[0][s % 3]);
            static_for<1, NCT>([&](auto C_) { lds_wait1(tv[decltype(C_)::value][s % 3]); });   // (already there: ties the registers)
        };
        auto mul = [&](auto S_, auto T_) {
            constexpr int s = decltype(S_)::value, set = s & 1, t = decltype(T_)::value;
            if constexpr (s < NQ) {
                constexpr int a = s / ND, dlt = s % ND;
#ifdef MG_NOMUL
                z[set][t] = dlt == 0 ? ar[a & 1][t] : bv[set][t];
#else
                z[set][t] = dlt == 0 ? ar[a & 1][t] * ar[a & 1][t] : ar[a & 1][t] * bv[set][t];
#endif
            } else if constexpr (s < NQ + Q) z[set][t] = bv[set][t];
            else z[set][t] = 1.0;
        };
        auto boundary = [&](auto S_) {                     // in front of the first fetch of a chunk
            constexpr int s = decltype(S_)::value;
            if constexpr (s % CH == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int cg = kt * NCH + s / CH;
                if (cg + 1 < nchunks) stage(cg + 1);
            }
        };
        boundary(ic<0>{});
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
            static_for<0, 4>([&](auto T_) {
                constexpr int t = decltype(T_)::value;
                static_for<0, NCT>([&](auto C_) {
                    constexpr int c = decltype(C_)::value;
                    acc[c][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[c][s % 3], z[s & 1][t], acc[c][t], 0, 0, 0);
                });
                if constexpr (s + 1 < NSTEP) mul(ic<s + 1>{}, T_);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if (tile * 64 < N) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[((size_t)tile * (KT * 16) + (kt * NCT + c) * 16 + g + 4 * r) * 64 + 16 * t + s16] = acc[c][t][r];
        }
    }
}

template <int D> static void build_image(int K, const std::vector<double> &P, const std::vector<double> &mu,
                                        const std::vector<double> &cen, std::vector<double> &img)
{
    using C = Cfg<D>;
    const int KT = K / 16;
    img.assign((size_t)KT * C::NSTEPP * 64, 0.0);
    for (int k = 0; k < K; ++k) {
        const double *Pk = &P[(size_t)k * D * D];
        double dlt[D], Pd[D], cst = 0.0;
        for (int i = 0; i < D; ++i) dlt[i] = mu[(size_t)k * D + i] - cen[i];
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s += Pk[i * D + j] * dlt[j];
            Pd[i] = s;
            cst += s * dlt[i];
        }
        const int kt = k / 16, m = k % 16;
        for (int s = 0; s < C::NSTEP; ++s)
            for (int g = 0; g < 4; ++g) {
                double v;
                if (s < C::NQ) {
                    const int a = s / C::ND, d = s % C::ND;
                    const int i = (a + g * C::Q) % D, j = (a + d + g * C::Q) % D;
                    if (d == 0) v = Pk[i * D + i];
                    else if (d == 2 * C::Q) v = g < 2 ? 2.0 * Pk[i * D + j] : 0.0;
                    else v = 2.0 * Pk[i * D + j];
                } else if (s < C::NQ + C::Q) {
                    const int i = (s - C::NQ + g * C::Q) % D;
                    v = -2.0 * Pd[i];
                } else v = 0.25 * cst;
                img[((size_t)kt * C::NSTEPP + s) * 64 + 16 * g + m] = v;
            }
    }
}

int main(int argc, char **argv)
{
    constexpr int D = MG_D;
    using C = Cfg<D>;
    const long long N = argc > 1 ? atoll(argv[1]) : 2000000;
    const int K = argc > 2 ? atoi(argv[2]) : 128;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    if (K % 16) { printf("K must be a multiple of 16\n"); return 1; }
    srand(1);
    auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
    std::vector<double> mu((size_t)K * D), P((size_t)K * D * D), cen(D, 0.0), x((size_t)N * D);
    for (auto &v : mu) v = 6.0 * rnd();
    for (int k = 0; k < K; ++k) {
        std::vector<double> A(D * D);
        for (auto &v : A) v = rnd();
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double s = i == j ? 0.5 : 0.0;
                for (int l = 0; l < D; ++l) s += A[i * D + l] * A[j * D + l] / D * 4.0;
                P[(size_t)k * D * D + i * D + j] = s;
            }
    }
    for (int j = 0; j < D; ++j) {
        double lo = 1e300, hi = -1e300;
        for (int k = 0; k < K; ++k) { lo = fmin(lo, mu[(size_t)k * D + j]); hi = fmax(hi, mu[(size_t)k * D + j]); }
        cen[j] = 0.5 * (lo + hi);
    }
    for (long long n = 0; n < N; ++n) {
        const int k = rand() % K;
        for (int j = 0; j < D; ++j) x[n * D + j] = mu[(size_t)k * D + j] + 2.0 * rnd();
    }
    std::vector<double> img;
    build_image<D>(K, P, mu, cen, img);
    const long long ntiles = (N + 63) / 64;
    double *dx, *dc, *di, *dout;
    hipMalloc(&dx, x.size() * 8); hipMalloc(&dc, D * 8); hipMalloc(&di, img.size() * 8);
    hipMalloc(&dout, (size_t)ntiles * K * 64 * 8);
    hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, cen.data(), D * 8, hipMemcpyHostToDevice);
    hipMemcpy(di, img.data(), img.size() * 8, hipMemcpyHostToDevice);
    if ((K / 16) % MG_NCT) { printf("K / 16 must be a multiple of MG_NCT\n"); return 1; }
    const size_t ldsb = (2 * MG_NCT * C::CH * 64 + 4 * D * C::RS) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_maha<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const unsigned grid = (unsigned)((ntiles + 3) / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_maha<D>, dim3(grid), dim3(256), ldsb, 0, dx, N, dc, di, K / 16, dout);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<double> out((size_t)ntiles * K * 64);
    hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
    double maxerr = 0.0, maxrel = 0.0;
    for (long long n = 0; n < N; n += (n < 256 ? 1 : 9973)) {
        for (int k = 0; k < K; ++k) {
            long double s = 0.0L;
            for (int i = 0; i < D; ++i) {
                long double r = 0.0L;
                for (int j = 0; j < D; ++j) r += (long double)P[(size_t)k * D * D + i * D + j] * ((long double)x[n * D + j] - mu[(size_t)k * D + j]);
                s += r * ((long double)x[n * D + i] - mu[(size_t)k * D + i]);
            }
            const double got = out[((size_t)(n / 64) * K + k) * 64 + n % 64];
            const double e = fabs(got - (double)s);
            if (e > maxerr) maxerr = e;
            if (e / (fabs((double)s) + 1.0) > maxrel) maxrel = e / (fabs((double)s) + 1.0);
        }
    }
    const double pairs = (double)N * K;
    printf("NCT=%d D=%d K=%d N=%lld steps=%d lds=%zu  %.3f ms  %.2f ps/pair  mfma slots %.1f TFLOP/s  (algorithmic D^2+4D+40: %.1f TFLOP/s)  max abs err %.2e rel %.2e\n",
           MG_NCT, D, K, N, C::NSTEP, ldsb, best, best * 1e9 / pairs, pairs * C::NSTEP * 8.0 / best * 1e-9,
           pairs * (D * D + 4.0 * D + 40.0) / best * 1e-9, maxerr, maxrel);
    return 0;
}
