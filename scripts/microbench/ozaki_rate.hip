// Pricing of an Ozaki-style int8 form of the statistics Uᵀ Z (verdict r5 #4; the accuracy side: ozaki_accuracy.py): the two
// primitive rates on this chip.
//   hipcc --offload-arch=gfx950 -O3 ozaki_rate.hip -o ozaki_rate && ./ozaki_rate
// (a) slicing: z = d_i d_j (a monomial of a sample), scaled by a per-monomial power of two, cut into NS signed 7-bit digits
//     (t *= 128; q = rint(t); t -= q) and packed 8 to a 64-bit word.  Lane = monomial, a run of 16 samples per packed operand
//     register -- the orientation in which v_mfma_i32_16x16x64_i8's k index (the sample) runs inside a lane, so no transpose.
//     Reported: ns per (sample x 861 monomials) if the whole chip did nothing else.
// (b) v_mfma_i32_16x16x64_i8 with its operands read from LDS (ds_read_b128), one read per RE matrix instructions: sustained TOPS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NS = 7;            // digits per value
constexpr int NMON = 861;        // monomials of d (D = 40) up to degree 2
constexpr int D = 40;

__global__ __launch_bounds__(256) void k_slice(const double *__restrict__ dall, int nsamp, unsigned long long *__restrict__ out)
{
    // a workgroup: 256 lanes = 256 monomials (4 passes cover 861), streaming over its samples; d of 16 samples at a time in LDS
    __shared__ double ds[16][D];
    __shared__ unsigned long long sink[256];
    const int lane = threadIdx.x;
    unsigned long long acc = 0;
    for (int s0 = blockIdx.x * 16; s0 + 16 <= nsamp; s0 += gridDim.x * 16) {
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * D; e += 256) ds[e / D][e % D] = dall[(size_t)s0 * D + e];
        __syncthreads();
        for (int pass = 0; pass < 4; ++pass) {
            const int m = pass * 256 + lane;
            if (m >= NMON) break;
            // monomial m -> (i, j): a fixed pseudo-map (the real kernel has a table); scale: a power of two per monomial
            const int i = m % D, j = (m * 7 + 3) % D;
            const double scale = __longlong_as_double((long long)(1023 - 4 - (m & 3)) << 52);
            unsigned digits[NS][4];                              // NS slices x 16 samples x 1 byte
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) digits[s][q] = 0;
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                double t = (ds[n][i] * ds[n][j]) * scale;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    t *= 128.0;
                    const double q = __builtin_rint(t);
                    t -= q;
                    digits[s][n >> 2] |= ((unsigned)(int)q & 0xffu) << (8 * (n & 3));
                }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) acc += ((unsigned long long)(digits[s][0] ^ digits[s][2]) << 32) | (digits[s][1] ^ digits[s][3]);
        }
    }
    sink[lane] = acc;
    out[blockIdx.x * 256 + lane] = sink[lane];
}

typedef int v4i __attribute__((ext_vector_type(4)));
template <int RE>
__global__ __launch_bounds__(256) void k_mfma_i8(int *__restrict__ out, int iters)
{
    __shared__ v4i frag[2][64 * 8];
    for (int e = threadIdx.x; e < 2 * 64 * 8; e += 256) (&frag[0][0])[e] = v4i{e, e * 3, e * 5, e * 7};
    __syncthreads();
    v4i acc[RE];
#pragma unroll
    for (int r = 0; r < RE; ++r) acc[r] = v4i{0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    v4i a[RE];
#pragma unroll
    for (int r = 0; r < RE; ++r) a[r] = frag[0][(lane + 64 * r) & 511];
    for (int it = 0; it < iters; ++it) {
        const v4i b = frag[1][(lane + 64 * (it & 7)) & 511];       // one LDS read per RE matrix instructions
#pragma unroll
        for (int r = 0; r < RE; ++r) acc[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[r], b, acc[r], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int r = 0; r < RE; ++r) s += acc[r].x + acc[r].y + acc[r].z + acc[r].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// (c) do the two overlap?  One workgroup of eight wavefronts -- two per SIMD --: four cut digits (VALU: v_mul_f64, v_rndne_f64, ...),
//     four issue the int8 matrix instructions; `roles` bit 0 / bit 1 switch each half on.  If the pipes run side by side the
//     time of both is the larger of the two, not their sum.
__global__ __launch_bounds__(512) void k_both(const double *__restrict__ dall, int rounds, int iters, int roles,
                                              unsigned long long *__restrict__ out)
{
    __shared__ double ds[16][D];
    __shared__ v4i frag[2][64 * 8];
    for (int e = threadIdx.x; e < 16 * D; e += 512) ds[e / D][e % D] = dall[e];
    for (int e = threadIdx.x; e < 2 * 64 * 8; e += 512) (&frag[0][0])[e] = v4i{e, e * 3, e * 5, e * 7};
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long res = 0;
    if (wave < 4) {
        if (roles & 1) {
            for (int r = 0; r < rounds; ++r) {
                const int m = (r * 256 + threadIdx.x) % NMON;
                const int i = m % D, j = (m * 7 + 3) % D;
                const double scale = __longlong_as_double((long long)(1023 - 4 - (m & 3)) << 52);
                unsigned dg[NS][4];
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q) dg[s][q] = 0;
#pragma unroll
                for (int n = 0; n < 16; ++n) {
                    double t = (ds[n][i] * ds[n][j]) * scale;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        t *= 128.0;
                        const double q = __builtin_rint(t);
                        t -= q;
                        dg[s][n >> 2] |= ((unsigned)(int)q & 0xffu) << (8 * (n & 3));
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) res += ((unsigned long long)(dg[s][0] ^ dg[s][2]) << 32) | (dg[s][1] ^ dg[s][3]);
            }
        }
    } else if (roles & 2) {
        v4i acc[8], a[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[r] = v4i{0, 0, 0, 0};
            a[r] = frag[0][(lane + 64 * r) & 511];
        }
        for (int it = 0; it < iters; ++it) {
            const v4i b = frag[1][(lane + 64 * (it & 7)) & 511];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[r], b, acc[r], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) res += (unsigned)(acc[r].x + acc[r].y + acc[r].z + acc[r].w);
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <class F> double time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const int nsamp = 1 << 20;
    double *d;
    unsigned long long *o;
    int *oi;
    CK(hipMalloc(&d, sizeof(double) * (size_t)nsamp * D));
    CK(hipMalloc(&o, sizeof(unsigned long long) * 4096 * 256));
    CK(hipMalloc(&oi, sizeof(int) * 4096 * 256));
    double *h = (double *)malloc(sizeof(double) * (size_t)nsamp * D);
    for (size_t i = 0; i < (size_t)nsamp * D; ++i) h[i] = (double)rand() / RAND_MAX * 4.0 - 2.0;
    CK(hipMemcpy(d, h, sizeof(double) * (size_t)nsamp * D, hipMemcpyHostToDevice));
    const double ms = time_ms([&] { hipLaunchKernelGGL(k_slice, dim3(2048), dim3(256), 0, 0, d, nsamp, o); }, 5);
    printf("slicing: %d samples x %d monomials x %d digits: %.3f ms = %.3f ns per sample (chip-wide)\n", nsamp, NMON, NS, ms,
           ms * 1e6 / nsamp);
    const int iters = 20000;
    const double m8 = time_ms([&] { hipLaunchKernelGGL((k_mfma_i8<8>), dim3(2048), dim3(256), 0, 0, oi, iters); }, 3);
    const double m4 = time_ms([&] { hipLaunchKernelGGL((k_mfma_i8<4>), dim3(2048), dim3(256), 0, 0, oi, iters); }, 3);
    const double ops = 2.0 * 16 * 16 * 64;
    printf("v_mfma_i32_16x16x64_i8, one ds_read_b128 per 8 instructions: %.0f TOPS;  per 4: %.0f TOPS\n",
           ops * 8 * iters * 2048.0 * 4 / (m8 * 1e-3) * 1e-12, ops * 4 * iters * 2048.0 * 4 / (m4 * 1e-3) * 1e-12);
    // the price: per sample 28 products x 128 x 861 multiply-adds
    const double macs = 28.0 * 128 * 861;
    const double tops8 = ops * 8 * iters * 2048.0 * 4 / (m8 * 1e-3);
    printf("28 slice products x 128 components x 861 monomials per sample at that rate: %.3f ns per sample; k_stats_gemm<40> today: 3.38\n",
           2.0 * macs / tops8 * 1e9);
    // overlap: rounds of slicing (one round = 256 monomials x 16 samples) against iterations of 8 matrix instructions
    const int rounds = 750, it2 = 14000;
    double t[4];
    for (int roles = 1; roles <= 3; ++roles)
        t[roles] = time_ms([&] { hipLaunchKernelGGL(k_both, dim3(1024), dim3(512), 0, 0, d, rounds, it2, roles, o); }, 3);
    printf("two wavefronts per SIMD, one cutting digits, one issuing int8 matrix instructions: digits alone %.3f ms, matrix alone %.3f ms, "
           "both %.3f ms (sum %.3f, max %.3f)\n", t[1], t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
    return 0;
}
