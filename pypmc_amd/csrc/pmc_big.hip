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
// blockDim = 64 * subtiles_per_wg; wavefront w of block b owns the 16-sample sub-tile g = b * spw + w, i.e. the
// lanes 16 (g mod 4) ... + 15 of tile g / 4.  LDS: per wavefront D4 x 16 doubles (D4 = D rounded up to 4),
// xs[c * 16 + s] = coordinate c of the sub-tile's sample s (zeros beyond D and beyond N).
__global__ __launch_bounds__(256) void k_big_maha(const PmcArgsM a)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.D, D4 = (D + 3) & ~3;
    const long long g = (long long)blockIdx.x * a.subtiles_per_wg + wave;
    const long long ntiles = (a.N + 63) >> 6;
    if (g >= ntiles * 4) return;                          // (no barriers in this kernel)
    const int s = lane & 15, q = lane >> 4;
    double *xs = big_lds + (size_t)wave * D4 * 16;

    const long long row = g * 16 + s;
    const double *xr = a.x + (row < a.N ? row : 0) * (long long)D;
    for (int c0 = 0; c0 < D4; c0 += 4) {
        const int c = c0 + q;
        xs[c * 16 + s] = (row < a.N && c < D) ? xr[c] : 0.0;
    }
    // (a wavefront reads only what it wrote: no barrier, the compiler's lgkmcnt wait orders it)

    const int G16 = (D + 15) >> 4;
    double *out = a.mtile + ((size_t)(g >> 2) * a.K) * 64 + (size_t)(g & 3) * 16 + s;
    for (int k = 0; k < a.K; ++k) {
        const double *pk = a.pack + (size_t)k * a.stride;
        double acc2 = 0.0;
        for (int I = 0; I < G16; ++I) {
            const int r = 16 * I + s;                     // this lane's row of R (A operand: i = lane & 15)
            const bool rv = r < D;
            const int rr = rv ? r : 0;
            // element (r, c >= r) of the packed upper triangle sits at rbase[c]
            const double *rbase = pk + D + (long long)rr * D - (long long)rr * (rr - 1) / 2 - rr;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int kk = 16 * I; kk < D4; kk += 4) {
                const int c = kk + q;                     // A: k = lane >> 4;  B: k = lane >> 4
                const bool cv = c < D;
                const double A = (rv && cv && c >= r) ? rbase[c] : 0.0;
                const double mu = cv ? pk[c] : 0.0;
                const double B = xs[c * 16 + s] - mu;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, acc, 0, 0, 0);
            }
            // C[i][j]: j = sample, i = row of this block -- all 16 rows of a sample are summed below
            acc2 = fma(acc[0], acc[0], acc2);
            acc2 = fma(acc[1], acc[1], acc2);
            acc2 = fma(acc[2], acc[2], acc2);
            acc2 = fma(acc[3], acc[3], acc2);
        }
        acc2 += __shfl_xor(acc2, 16, 64);
        acc2 += __shfl_xor(acc2, 32, 64);
        if (q == 0) out[(size_t)k * 64] = acc2;
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

extern "C" hipError_t pmc_launch_big_maha(const PmcArgsM &a, hipStream_t st)
{
    const long long nsub = ((a.N + 63) >> 6) * 4;
    const int spw = a.subtiles_per_wg;
    const size_t lds = sizeof(double) * (size_t)spw * ((a.D + 3) & ~3) * 16;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_big_maha),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_big_maha, dim3((unsigned)((nsub + spw - 1) / spw)), dim3(64 * spw), lds, st, a);
    return hipGetLastError();
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
