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
#ifdef PMC_STATS_NS
    return PMC_STATS_NS;
#else
    return D <= 20 ? 4 : (D <= 40 ? 2 : 1);
#endif
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain the vector
// memory counter too (s_waitcnt vmcnt(0) before every s_barrier), which would stall each step on
// the global prefetch it has just issued; here only this wavefront's LDS operations are waited for
// and the prefetch stays in flight across the barrier (the compiler still places counted vmcnt
// waits before the first use of a loaded register).
typedef double pmc_vec2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double vec_get(double a, int) { return a; }
__device__ __forceinline__ double vec_get(pmc_vec2 a, int v) { return v == 0 ? a.x : a.y; }

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int D, bool PADDED, int WAVES, int SUB>
__device__ __forceinline__ void stats_run(const PmcArgsB &b, double *xs, int k, long long t0,
                                          long long t1, int chunk)
{
    constexpr int STRIDE = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    constexpr int NT = WAVES * 64;
    constexpr int NS = stats_ns<D>();
    constexpr int LDP = NS * 64 + 1;                      // row pitch: conflict-free b64 access
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

    // Software pipeline over steps of NS tiles (256 samples for D <= 20): while step s is consumed
    // from one LDS buffer, the global loads of step s+1 are in flight into registers; they are
    // written (transposed) into the other LDS buffer after the arithmetic; one LDS-only barrier per
    // step.  Loads use "uniform base (SGPR pair) + loop-invariant 32-bit thread offset" addressing
    // and are branch-free: offsets are clamped into the array instead of guarded (a clamped lane
    // reads some other, finite sample, and its weight u is zero), so a load costs ~1 VALU op.
    // Pairs of consecutive doubles (16-byte loads) when the row length is even: one
    // global_load_dwordx4 per two elements (8-byte vector loads run at ~0.6x the rate).
    constexpr int VW = (!PADDED && D % 2 == 0) ? 2 : 1;    // doubles per load
    constexpr int NLV = (NS * 64 * D / VW + NT - 1) / NT;  // loads per thread and step
    typedef double vec2_t __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<VW == 2, vec2_t, double>::type vec_t;
    vec_t xn[NLV];
    double un[NS];
    unsigned xoff[NLV];
    int lds_off[NLV][VW];
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
        const int e = (tid + i * NT) * VW;                 // first element of the pair
        xoff[i] = (unsigned)(tid + i * NT);                // in units of vec_t
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            const int ee = e + v;
            const int nloc = PADDED ? ee / dreal : ee / D;
            const int j = PADDED ? ee % dreal : ee % D;
            lds_off[i][v] = (ee < NS * 64 * dreal) ? j * LDP + nloc : -1;
        }
    }
    // part `part` of `nparts` of the loads of the step starting at tile t: the loads are issued
    // in NS slices, one per sub-step of the arithmetic, so that the texture addresser sees a steady
    // trickle instead of 8 wavefronts x 9 loads right after every barrier
    auto fetch = [&](long long t, int part, int nparts) {
        const long long tt = t < t1 ? t : t0;                                   // keep addresses in range
        const vec_t *__restrict__ xt = (const vec_t *)(b.x + tt * 64 * dreal);  // wave-uniform
        const long long rem = (total - tt * 64 * dreal) / VW - 1;               // >= 0
        const unsigned lim = rem > 0x7ffffff0ll ? 0x7ffffff0u : (unsigned)rem;
#pragma unroll
        for (int i = 0; i < NLV; ++i) {
            if (i % nparts != part) continue;
#ifdef PMC_EXP_NOFETCH
            xn[i] = vec_t((double)(xoff[i] & 1023) * 1e-3);
#else
            xn[i] = xt[xoff[i] < lim ? xoff[i] : lim];
#endif
        }
        if constexpr (ACTIVE) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (q % nparts != part) continue;
                const bool in = t + q < t1;                                     // wave-uniform
                const double *__restrict__ uq = b.u + ((size_t)(in ? t + q : t0) * b.K + k) * 64;
#ifdef PMC_EXP_NOU
                un[q] = in ? (double)(lane & 255) * 1e-3 : 0.0;
#else
                const double v = uq[lane];
                un[q] = in ? v : 0.0;
#endif
            }
        }
    };
    auto stage = [&](double *xb) {
#ifdef PMC_EXP_NOSTAGE
        return;
#endif
#pragma unroll
        for (int i = 0; i < NLV; ++i) {
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                const double val = vec_get(xn[i], v);
                if ((NS * 64 * D) % (NT * VW) == 0 && !PADDED) xb[lds_off[i][v]] = val;
                else if (lds_off[i][v] >= 0) xb[lds_off[i][v]] = val;
            }
        }
    };

    int buf = 0;
    fetch(t0, 0, 1);
    stage(xs);
    lds_barrier();
    for (long long t = t0; t < t1; t += NS, buf ^= 1) {
        const double *xb = xs + buf * (D * LDP);
        double uc[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) uc[q] = un[q];
        if constexpr (!ACTIVE) fetch(t + NS, 0, 1);
        if constexpr (ACTIVE) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                fetch(t + NS, q, NS);
#ifndef PMC_EXP_NOSCHEDBARRIER
                // keep the scheduler from overlapping the LDS reads of all NS sub-steps at once
                // (it would need NS x (rows+cols) extra registers and spill)
                __builtin_amdgcn_sched_barrier(0);
#endif
                const double u = uc[q];                   // zero for tiles beyond the chunk
                if constexpr (TK::zeroth) acc0 += u;
                const double *xl = xb + q * 64 + lane;
                double dr[TK::NR], dc[TK::NC];
#pragma unroll
#ifdef PMC_EXP_NOLDSREAD
                for (int i = 0; i < TK::NR; ++i) dr[i] = (u + (double)i) - pk[TK::r0 + i];
#else
                for (int i = 0; i < TK::NR; ++i) dr[i] = xl[(TK::r0 + i) * LDP] - pk[TK::r0 + i];
#endif
                if constexpr (TK::diag) {
#pragma unroll
                    for (int j = 0; j < TK::NC; ++j) dc[j] = dr[j];
                } else {
#pragma unroll
#ifdef PMC_EXP_NOLDSREAD
                    for (int j = 0; j < TK::NC; ++j) dc[j] = (u - (double)j) - pk[TK::c0 + j];
#else
                    for (int j = 0; j < TK::NC; ++j) dc[j] = xl[(TK::c0 + j) * LDP] - pk[TK::c0 + j];
#endif
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
        stage(xs + (buf ^ 1) * (D * LDP));
        lds_barrier();
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
