// pmc_big.hip -- the run-time-dimension unit: sample dimensions beyond the compiled ones (PMC_MAX_DIM < D <=
// PMC_BIG_MAX_DIM).  The reference's loops take a vector of any length (pypmc/tools/_linalg.pyx:32-37); the
// compiled units keep a sample's coordinates and a component's accumulators in registers, which ends at D = 64.
// Up there the work per (sample, component) pair is D^2/2 multiply-adds against one exp, so the path is cut where
// it is GEMM-shaped and the rest reuses the per-sample kernels as they are:
//
//   k_big_maha   maha_nk = |R_k (x_n - mu_k)|^2 for all pairs, tile-major -- 16 x 16 x 4 fp64 MFMA: the triangular
//                product Y = R_k D_k for 16 samples per wavefront, R_k's 16-row blocks as the A operand straight
//                from the row-major packed factor (L2 / L1: every wavefront walks the same component at the same
//                time), the samples' coordinates as the B operand from a per-wavefront LDS stage that is filled
//                once and serves all K components;
//   k_logpdf<0>, k_resp<0>   (pmc_persample.hip, engine TILES) read those forms -- every fused output, kind and
//                mode of the compiled units, same code;
//   k_big_stats  sum u | sum u d | sum u d d^T:  one wavefront per (sample chunk, component, 16 x 16 block of the
//                lower triangle), 4 samples per v_mfma_f64_16x16x4_f64, partials in k_stats' layout for the same
//                fixed-order finishing kernel.
//   (k_propose<0>: pmc_propose.hip)
//
// Operand layouts of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md):  A[i][k] lane 16 k + i,  B[k][j] lane
// 16 k + j,  C[i][j] lane 16 (i mod 4) + j, register i / 4.
#include "pmc_device.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

extern __shared__ double big_lds[];

// ---------------------------------------------------------------------------------------------
// k_big_maha
// ---------------------------------------------------------------------------------------------
// One workgroup = 4 wavefronts = NT sub-tiles of 16 samples (NT = 4: one 64-sample tile; 2 / 1 where the LDS holds
// no more).  The samples' coordinates are staged ONCE in LDS, xs[c * P + sample] (P = 16 NT + 1: the staging
// writes -- a lane per coordinate of one row, coalesced in HBM -- land in distinct banks), and serve all K
// components.  Per component the 16-row blocks of R are dealt to the wavefronts in pairs (I, G16 - 1 - I) of equal
// total length, so no two wavefronts load the same rows of R, and every A operand feeds NT instructions (one per
// sub-tile).  A lane fetches 4 consecutive... no: element (r, kk + 4 q + t), t = 0..3 of its row -- the 16 columns
// of a step are assigned to the instructions' k slots as {kk + 4 q + t : q}, the same for A and B -- one step
// ahead of the 4 NT instructions that consume them.  The wavefronts' partial |y|^2 meet in LDS (one barrier per
// component, buffers alternating), summed in wavefront order: bit-reproducible.
template <int NT>
__global__ __launch_bounds__(256) void k_big_maha(const PmcArgsM a)
{
    constexpr int NS = 16 * NT, P = NS + 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.D, D16 = (D + 15) & ~15, G16 = D16 >> 4;
    const int i = lane & 15, q = lane >> 4;
    double *xs = big_lds;                                  // D16 x P
    double *red = big_lds + (size_t)D16 * P;               // 2 x 4 x NS
    const long long g0 = (long long)blockIdx.x * NT;       // first sub-tile
    const long long n0 = g0 * 16;

    // stage: thread -> (sample row, coordinate), coordinates fastest (coalesced); zeros beyond D and beyond N
    for (int idx = threadIdx.x; idx < NS * D16; idx += 256) {
        const int sl = idx / D16, c = idx - sl * D16;
        const long long row = n0 + sl;
        xs[c * P + sl] = (row < a.N && c < D) ? a.x[row * D + c] : 0.0;
    }
    __syncthreads();

    const int npair = (G16 + 1) >> 1;
    for (int k = 0; k < a.K; ++k) {
        const double *pk = a.pack + (size_t)k * a.stride;
        double part[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) part[t] = 0.0;
        for (int p = wave; p < npair; p += 4) {
            for (int half = 0; half < 2; ++half) {
                const int I = half == 0 ? p : G16 - 1 - p;
                if (half == 1 && I == p) break;            // (odd G16: the middle block is its own partner)
                const int r = 16 * I + i;                  // this lane's row of R (A operand: i = lane & 15)
                const bool rv = r < D;
                const int rr = rv ? r : D - 1;
                // element (r, c >= r) of the packed upper triangle sits at rbase[c]
                const double *rbase = pk + D + (long long)rr * D - (long long)rr * (rr - 1) / 2 - rr;
                d4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
                double An[4], mn[4];
                auto fetch = [&](int kk) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int c = kk + 4 * q + t;
                        const int cc = c < rr ? rr : (c < D ? c : D - 1);      // inside the row's storage
                        const double v = rbase[cc];
                        An[t] = (rv && c >= r && c < D) ? v : 0.0;
                        const double m = pk[c < D ? c : D - 1];
                        mn[t] = c < D ? m : 0.0;
                    }
                };
                fetch(16 * I);
                for (int kk = 16 * I; kk < D16; kk += 16) {
                    double A[4], mu[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { A[t] = An[t]; mu[t] = mn[t]; }
                    if (kk + 16 < D16) fetch(kk + 16);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double *xc = xs + (size_t)(kk + 4 * q + t) * P + i;
#pragma unroll
                        for (int sub = 0; sub < NT; ++sub) {
                            const double B = xc[16 * sub] - mu[t];
                            acc[sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t], B, acc[sub], 0, 0, 0);
                        }
                    }
                }
                // C[i][j]: j = sample, i = row of this block -- all 16 rows of a sample are summed below
#pragma unroll
                for (int sub = 0; sub < NT; ++sub) {
                    part[sub] = fma(acc[sub][0], acc[sub][0], part[sub]);
                    part[sub] = fma(acc[sub][1], acc[sub][1], part[sub]);
                    part[sub] = fma(acc[sub][2], acc[sub][2], part[sub]);
                    part[sub] = fma(acc[sub][3], acc[sub][3], part[sub]);
                }
            }
        }
        double *rk = red + (size_t)(k & 1) * 4 * NS;
#pragma unroll
        for (int sub = 0; sub < NT; ++sub) {
            double v = part[sub];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) rk[wave * NS + 16 * sub + i] = v;
        }
        __syncthreads();
        if (wave == (k & 3) && lane < NS) {
            const double tot = ((rk[lane] + rk[NS + lane]) + rk[2 * NS + lane]) + rk[3 * NS + lane];
            const long long g = g0 + (lane >> 4);
            if (g * 16 < ((a.N + 63) >> 6) * 64)
                a.mtile[((size_t)(g >> 2) * a.K + k) * 64 + (size_t)(g & 3) * 16 + (lane & 15)] = tot;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_big_stats
// ---------------------------------------------------------------------------------------------
// Block -> (chunk, task group) as in k_stats (all groups of a sample chunk on one XCD); 4 wavefronts = 4
// consecutive tasks (k, p), p = I (I + 1) / 2 + J the 16 x 16 block (I, J <= I) of the lower triangle.
// A[i][s] = u_s d_s[16 I + i], B[s][j] = d_s[16 J + j], 4 samples per instruction; diagonal tasks also carry the
// first moments of their 16 coordinates, task (0, 0) the sum of the weights.
__global__ __launch_bounds__(256) void k_big_stats(const PmcArgsB b)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = b.dreal;
    const int G16 = (D + 15) >> 4, npairs = G16 * (G16 + 1) / 2;
    const int bid = blockIdx.x;
    const int qd = bid >> 3;
    const int chunk = (bid & 7) + 8 * (qd / b.ngroups);
    const int group = qd % b.ngroups;
    const long long task = (long long)group * 4 + wave;
    const int k = (int)(task / npairs), p = (int)(task % npairs);
    if (k >= b.K) return;
    int I = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > p) --I;
    while ((I + 1) * (I + 2) / 2 <= p) ++I;
    const int J = p - I * (I + 1) / 2;
    const bool diag = I == J;

    const long long t0 = (long long)chunk * b.tiles_per_chunk;
    long long t1 = t0 + b.tiles_per_chunk;
    if (t1 > b.ntiles) t1 = b.ntiles;
    const int stride = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    const double *pk = b.pack + (size_t)k * stride;
    const int i16 = lane & 15, s = lane >> 4;
    const int ci = 16 * I + i16, cj = 16 * J + i16;
    const bool civ = ci < D, cjv = cj < D;
    const double mui = civ ? pk[ci] : 0.0, muj = cjv ? pk[cj] : 0.0;
    const int cic = civ ? ci : 0, cjc = cjv ? cj : 0;

    d4 acc = {0.0, 0.0, 0.0, 0.0};
    double acc1 = 0.0, acc0 = 0.0;
    long long n1 = t1 * 64;
    if (n1 > b.N) n1 = b.N;
#pragma unroll 4
    for (long long n0 = t0 * 64; n0 < n1; n0 += 4) {
        const long long n = n0 + s;
        const bool nv = n < n1;
        const long long nn = nv ? n : n0;
        const double *xr = b.x + nn * D;
        const double u = nv ? b.u[((size_t)(nn >> 6) * b.K + k) * 64 + (nn & 63)] : 0.0;
        const double di = (nv && civ) ? xr[cic] - mui : 0.0;
        const double dj = diag ? di : ((nv && cjv) ? xr[cjc] - muj : 0.0);
        const double A = u * di;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A, dj, acc, 0, 0, 0);
        if (diag) {                                       // wave-uniform
            acc1 += A;
            acc0 += u;
        }
    }

    double *out = b.partials + ((size_t)chunk * b.K + k) * PS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gi = 16 * I + s + 4 * r, gj = 16 * J + i16;     // C: col = lane & 15, row = (lane >> 4) + 4 r
        if (gi < D && gj <= gi) out[1 + D + (size_t)gi * (gi + 1) / 2 + gj] = acc[r];
    }
    if (diag) {
        acc1 += __shfl_xor(acc1, 16, 64);
        acc1 += __shfl_xor(acc1, 32, 64);
        if (s == 0 && civ) out[1 + ci] = acc1;
        if (I == 0) {
            acc0 += __shfl_xor(acc0, 16, 64);
            acc0 += __shfl_xor(acc0, 32, 64);
            if (lane == 0) out[0] = acc0;
        }
    }
}

}  // namespace

template <int NT> static hipError_t launch_maha(const PmcArgsM &a, hipStream_t st)
{
    const int D16 = (a.D + 15) & ~15;
    const size_t lds = sizeof(double) * ((size_t)D16 * (16 * NT + 1) + 2 * 4 * 16 * NT);
    if (lds > 65536) {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_big_maha<NT>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (once != hipSuccess) return once;
    }
    const long long nsub = ((a.N + 63) >> 6) * 4;
    hipLaunchKernelGGL((k_big_maha<NT>), dim3((unsigned)((nsub + NT - 1) / NT)), dim3(256), lds, st, a);
    return hipGetLastError();
}

// sub-tiles per workgroup: as many as fit the LDS next to the reduction buffers (156 KB)
extern "C" hipError_t pmc_launch_big_maha(const PmcArgsM &a, hipStream_t st)
{
    const size_t D16 = (size_t)((a.D + 15) & ~15);
    auto fits = [&](int nt) { return 8 * (D16 * (16 * nt + 1) + 2 * 4 * 16 * nt) <= 156 * 1024; };
    if (fits(4)) return launch_maha<4>(a, st);
    if (fits(2)) return launch_maha<2>(a, st);
    return launch_maha<1>(a, st);
}

extern "C" hipError_t pmc_launch_big_stats(const PmcArgsB &b, unsigned grid, hipStream_t st)
{
    hipLaunchKernelGGL(k_big_stats, dim3(grid), dim3(256), 0, st, b);
    return hipGetLastError();
}

// geometry knobs the dispatcher's stats_geom() uses: tasks per component, wavefronts per workgroup
extern "C" void pmc_big_stats_config(int D, int *nsub, int *waves)
{
    const int G16 = (D + 15) >> 4;
    *nsub = G16 * (G16 + 1) / 2;
    *waves = 4;
}
