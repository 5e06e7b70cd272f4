// pmc_stats.hip -- sufficient-statistics kernel of the hot path (see pmc_persample.hip for the
// execution-model notes), compiled once per sample dimension.
//
// Per component k, with d = x_n - mu_k and weight u_nk (tile-major, written by k_resp):
//     sum u | sum u d (D) | sum u d d^T (lower triangle)
//
// Work decomposition.  The D rows are cut into G groups of <= 10 rows.  The lower triangle of
// d d^T then consists of G diagonal blocks (b(b+1)/2 elements) and G(G-1)/2 off-diagonal blocks
// (b x b), each off-diagonal block split in two row halves: G^2 "block tasks" of <= 55 per-lane
// fp64 accumulators that need only the <= 20 coordinates of their own rows and columns.  One
// wavefront owns one (component, block task) and streams over its chunk of samples, one sample
// per lane and step; accumulators live in VGPRs for the whole chunk and are reduced across the
// wavefront once at the end (fixed order => deterministic).
//
// The 64 x D sample tiles are loaded coalesced from HBM once per workgroup, transposed through LDS
// ([coordinate][sample], pitch NS*64+1: conflict-free ds_read/ds_write_b64) and shared by the
// workgroup's wavefronts; global loads of step s+1 are in flight while step s is consumed.
#include "pmc_device.h"

namespace {

// ---------------------------------------------------------------------------------------------
// compile-time blocking of the lower triangle
// ---------------------------------------------------------------------------------------------
template <int D> struct Blocking {
    static constexpr int G = (D + 9) / 10;                // row groups of <= 10
    static constexpr int NSUB = G * G;                    // block tasks per component
    __host__ __device__ static constexpr int start(int g) { return (int)(((long long)g * D) / G); }
};

// task s -> block: s < G: diagonal block (s,s); otherwise off-diagonal (g,h), g > h, row half
template <int D, int S> struct Task {
    using B = Blocking<D>;
    static constexpr bool diag = S < B::G;
    static constexpr int pair = diag ? 0 : (S - B::G) / 2;
    static constexpr int half = diag ? 0 : (S - B::G) % 2;
    __host__ __device__ static constexpr int pair_g()
    {
        int p = pair, g = 1;
        while (p >= g) { p -= g; ++g; }
        return g;
    }
    __host__ __device__ static constexpr int pair_h()
    {
        int p = pair, g = 1;
        while (p >= g) { p -= g; ++g; }
        return p;
    }
    static constexpr int g = diag ? S : pair_g();
    static constexpr int h = diag ? S : pair_h();
    static constexpr int gr0 = B::start(g), gr1 = B::start(g + 1);
    static constexpr int mid = (gr0 + gr1) / 2;
    static constexpr int r0 = diag ? gr0 : (half == 0 ? gr0 : mid);      // rows [r0, r1)
    static constexpr int r1 = diag ? gr1 : (half == 0 ? mid : gr1);
    static constexpr int c0 = B::start(h), c1 = B::start(h + 1);          // columns [c0, c1)
    static constexpr int NR = r1 - r0, NC = c1 - c0;
    // first moments sum u d_i: rows of group 0 by its diagonal block, rows of group g > 0 by the
    // two halves of block (g, 0) -- this evens out the instruction count of the tasks
    static constexpr bool first_moments = diag ? (S == 0) : (h == 0);
    static constexpr bool zeroth = S == B::NSUB - 1;                      // sum u (lightest task)
};

// Samples per pipeline step: NS tiles of 64 (LDS: 2 buffers of D x (NS*64+1) doubles, <= ~84 KB)
template <int D> __host__ __device__ constexpr int stats_ns()
{
    return D <= 20 ? 4 : (D <= 40 ? 2 : 1);
}

template <int D, bool PADDED, int WAVES, int SUB>
__device__ __forceinline__ void stats_run(const PmcArgsB &b, double *xs, int k, long long t0,
                                          long long t1, int chunk)
{
    constexpr int STRIDE = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    constexpr int NT = WAVES * 64;
    constexpr int NS = stats_ns<D>();
    constexpr int LDP = NS * 64 + 1;                      // row pitch: conflict-free b64 access
    constexpr int NLD = (NS * 64 * D + NT - 1) / NT;      // staged doubles per thread and step
    constexpr bool ACTIVE = SUB >= 0;
    using TK = Task<D, ACTIVE ? SUB : 0>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int dreal = PADDED ? b.dreal : D;
    const long long total = b.N * (long long)dreal;

    double acc0 = 0.0;
    double acc1[TK::NR];
    double acc2[TK::NR][TK::NC];
#pragma unroll
    for (int i = 0; i < TK::NR; ++i) {
        acc1[i] = 0.0;
#pragma unroll
        for (int j = 0; j < TK::NC; ++j) acc2[i][j] = 0.0;
    }
    cdouble *pk = (cdouble *)b.pack + (size_t)(ACTIVE ? k : 0) * STRIDE;

    if constexpr (PADDED) {
        for (int e = tid; e < 2 * D * LDP; e += NT) xs[e] = 0.0;
        __syncthreads();
    }

    // software pipeline: while step s is consumed from LDS, the global loads of step s+1 are in
    // flight into registers; they are written (transposed) to the other LDS buffer after the
    // arithmetic, one barrier per step.
    double xn[NLD];
    double un[NS];
    auto fetch = [&](long long t) {
        const long long base = t * 64 * dreal;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            const long long g = base + e;
            xn[i] = (e < NS * 64 * dreal && g < total) ? b.x[g] : 0.0;
        }
        if constexpr (ACTIVE) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const bool in = t + q < t1;
                const size_t uo = ((size_t)(t + q) * b.K + k) * 64 + lane;
                un[q] = in ? b.u[uo] : 0.0;
            }
        }
    };
    auto stage = [&](double *xb) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            if (e < NS * 64 * dreal) {
                const int nloc = PADDED ? e / dreal : e / D;
                const int j = PADDED ? e % dreal : e % D;
                xb[j * LDP + nloc] = xn[i];
            }
        }
    };

    int buf = 0;
    if (t0 < t1) {
        fetch(t0);
        stage(xs);
    }
    __syncthreads();
    for (long long t = t0; t < t1; t += NS, buf ^= 1) {
        const double *xb = xs + buf * (D * LDP);
        double uc[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) uc[q] = un[q];
        const bool more = t + NS < t1;
        if (more) fetch(t + NS);
        if constexpr (ACTIVE) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                // keep the scheduler from overlapping the LDS reads of all NS sub-steps at once
                // (it would need NS x (rows+cols) extra registers and spill)
                __builtin_amdgcn_sched_barrier(0);
                if (t + q < t1) {
                    const double u = uc[q];
                    if constexpr (TK::zeroth) acc0 += u;
                    const double *xl = xb + q * 64 + lane;
                    double dr[TK::NR], dc[TK::NC];
#pragma unroll
                    for (int i = 0; i < TK::NR; ++i) dr[i] = xl[(TK::r0 + i) * LDP] - pk[TK::r0 + i];
                    if constexpr (TK::diag) {
#pragma unroll
                        for (int j = 0; j < TK::NC; ++j) dc[j] = dr[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < TK::NC; ++j) dc[j] = xl[(TK::c0 + j) * LDP] - pk[TK::c0 + j];
                    }
#pragma unroll
                    for (int i = 0; i < TK::NR; ++i) {
                        const double ud = u * dr[i];
                        if constexpr (TK::first_moments) acc1[i] += ud;
#pragma unroll
                        for (int j = 0; j < TK::NC; ++j)
                            if (!TK::diag || j <= i) acc2[i][j] = fma(ud, dc[j], acc2[i][j]);
                    }
                }
            }
        }
        if (more) stage(xs + (buf ^ 1) * (D * LDP));
        __syncthreads();
    }

    if constexpr (ACTIVE) {
        double *out = b.partials + ((size_t)chunk * b.K + k) * PS;
        if constexpr (TK::zeroth) {
            const double s0 = wave_sum(acc0);
            if (lane == 0) out[0] = s0;
        }
#pragma unroll
        for (int i = 0; i < TK::NR; ++i) {
            const int gi = TK::r0 + i;
            if constexpr (TK::first_moments) {
                const double s = wave_sum(acc1[i]);
                if (lane == 0) out[1 + gi] = s;
            }
#pragma unroll
            for (int j = 0; j < TK::NC; ++j) {
                if (!TK::diag || j <= i) {
                    const int gj = TK::c0 + j;
                    const double q = wave_sum(acc2[i][j]);
                    if (lane == 0) out[1 + D + gi * (gi + 1) / 2 + gj] = q;
                }
            }
        }
    }
}

template <int D, bool PADDED, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_stats(const PmcArgsB b)
{
    constexpr int NSUB = Blocking<D>::NSUB;
    extern __shared__ double xs[];                        // 2 * D * (NS*64+1) doubles
    // XCD-aware block -> (chunk, task group): hardware places block i on XCD i % 8; all task
    // groups of one sample chunk get the same residue so they share that XCD's L2 copy of the tile.
    const int bid = blockIdx.x;
    const int q = bid >> 3;
    const int chunk = (bid & 7) + 8 * (q / b.ngroups);
    const int group = q % b.ngroups;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR: uniform k, sub
    const int task = group * WAVES + wave;
    const int k = task / NSUB, sub = task % NSUB;
    const long long t0 = (long long)chunk * b.tiles_per_chunk;
    long long t1 = t0 + b.tiles_per_chunk;
    if (t1 > b.ntiles) t1 = b.ntiles;
    // (an empty chunk still publishes zeros, so the finishing kernel sums every chunk blindly)
    if (k >= b.K) {
        stats_run<D, PADDED, WAVES, -1>(b, xs, 0, t0, t1, chunk);
        return;
    }
    bool done = false;
    static_for<0, NSUB>([&](auto S) {
        constexpr int SUBC = decltype(S)::value;
        if (!done && sub == SUBC) {
            stats_run<D, PADDED, WAVES, SUBC>(b, xs, k, t0, t1, chunk);
            done = true;
        }
    });
}

constexpr int NSUB_ = Blocking<D_>::NSUB;
#ifdef PMC_STATS_WAVES                                     // tuning override (scripts/tune_stats.sh)
constexpr int SW_ = PMC_STATS_WAVES;
#else
constexpr int SW_ = 8;                                   // wavefronts per statistics workgroup
#endif

}  // namespace

// launch geometry knobs the dispatcher needs
extern "C" void PMC_UNIT_NAME_X(pmc_stats_config_d, PMC_D, PMC_PADDED)(int *nsub, int *waves)
{
    *nsub = NSUB_;
    *waves = SW_;
}

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_stats_d, PMC_D, PMC_PADDED)(const PmcArgsB &b, unsigned grid,
                                                                            hipStream_t st)
{
    constexpr size_t lds = sizeof(double) * 2 * D_ * (stats_ns<D_>() * 64 + 1);
    if constexpr (lds > 65536) {
        static const hipError_t once = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&k_stats<D_, P_, SW_>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((k_stats<D_, P_, SW_>), dim3(grid), dim3(SW_ * 64), lds, st, b);
    return hipGetLastError();
}
