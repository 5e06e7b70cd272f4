// pmc_stats.hip -- sufficient-statistics kernel of the hot path (see pmc_persample.hip for the
// execution-model notes), compiled once per sample dimension.
//
// Per component k, with d = x_n - mu_k and weight u_nk (tile-major, written by k_resp):
//     sum u | sum u d (D) | sum u d d^T (lower triangle)
//
// Work decomposition.  The D rows are cut into G groups of <= 10 rows (even boundaries).  The
// lower triangle of d d^T then consists of G diagonal blocks (b(b+1)/2 elements) and G(G-1)/2
// off-diagonal blocks (b x b), each off-diagonal block split in two row halves: G^2 "block tasks"
// of <= 55 per-lane fp64 accumulators that need only the <= 20 coordinates of their own rows and
// columns.  One wavefront owns one (component, block task) and streams over its chunk of samples,
// one sample per lane and sub-step; accumulators live in VGPRs for the whole chunk and are reduced
// across the wavefront once at the end (fixed order => deterministic).
//
// Data path.  The sample tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, no ds_write pass, no address arithmetic in the consumer loop), double buffered, one
// step (NS tiles) ahead of the arithmetic.  LDS image: one row of PITCH 16-byte slots per sample,
// slot p = coordinates (2p, 2p+1); PITCH is odd (pad slot if needed), so the per-lane
// ds_read_b128 of a coordinate pair is bank-conflict free, and every DMA instruction covers 64
// consecutive slots whose global sources are consecutive row pieces (fully coalesced).  The
// workgroup's wavefronts share the tile; all task groups of a sample chunk run on one XCD and
// share its L2 copy (measured HBM traffic = algorithmic).
#include "pmc_device.h"

namespace {

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef double pmc_vec2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// compile-time blocking of the lower triangle (in units of coordinate pairs)
// ---------------------------------------------------------------------------------------------
#ifndef PMC_STATS_BMAX
#define PMC_STATS_BMAX 10
#endif
template <int D> struct Blocking {
    static constexpr int NP = (D + 1) / 2;                // coordinate pairs per sample
    static constexpr int PITCH = NP | 1;                  // 16-byte slots per LDS row, odd
    static constexpr int G = (D + PMC_STATS_BMAX - 1) / PMC_STATS_BMAX;   // row groups of <= BMAX
    static constexpr int NSUB = G * G;                    // block tasks per component
    __host__ __device__ static constexpr int pstart(int g) { return (int)(((long long)g * NP) / G); }
    __host__ __device__ static constexpr int start(int g) { return 2 * pstart(g) < D ? 2 * pstart(g) : D; }
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
    static constexpr int pmid = (B::pstart(g) + B::pstart(g + 1)) / 2;
    static constexpr int mid = 2 * pmid < gr1 ? 2 * pmid : gr1;             // even split point
    static constexpr int r0 = diag ? gr0 : (half == 0 ? gr0 : mid);        // rows [r0, r1), r0 even
    static constexpr int r1 = diag ? gr1 : (half == 0 ? mid : gr1);
    static constexpr int c0 = B::start(h), c1 = B::start(h + 1);            // columns [c0, c1), c0 even
    static constexpr int NR = r1 - r0, NC = c1 - c0;
    // first moments sum u d_i: rows of group 0 by its diagonal block, rows of group g > 0 by the
    // two halves of block (g, 0) -- this evens out the instruction count of the tasks
    static constexpr bool first_moments = diag ? (S == 0) : (h == 0);
    static constexpr bool zeroth = S == B::NSUB - 1;                        // sum u (lightest task)
};

// Samples per pipeline step: NS tiles of 64 (LDS: 2 buffers of NS*64 rows of PITCH*16 bytes)
template <int D> __host__ __device__ constexpr int stats_ns()
{
#ifdef PMC_STATS_NS
    return PMC_STATS_NS;
#else
    return (Blocking<D>::PITCH * 16 * 64 * 4 * 2 <= 96 * 1024)
               ? 4
               : ((Blocking<D>::PITCH * 16 * 64 * 2 * 2 <= 96 * 1024) ? 2 : 1);
#endif
}

template <int D, bool PADDED, int WAVES, int SUB>
__device__ __forceinline__ void stats_run(const PmcArgsB &b, double *xs, int k, int wave, long long t0,
                                          long long t1, int chunk)
{
    using BL = Blocking<D>;
    constexpr int STRIDE = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    constexpr int NS = stats_ns<D>();
    constexpr int PITCH = BL::PITCH;
    constexpr int ROWD = PITCH * 2;                       // doubles per LDS row
    constexpr int BUFD = NS * 64 * ROWD;                  // doubles per LDS buffer
    constexpr int NDMA = NS * PITCH;                      // DMA instructions per step (64 slots each)
    constexpr int DMA_PER_WAVE = (NDMA + WAVES - 1) / WAVES;
#ifdef PMC_STATS_DMA_SLICES
    constexpr int DMA_SLICES = PMC_STATS_DMA_SLICES < NS ? PMC_STATS_DMA_SLICES : NS;
#else
    constexpr int DMA_SLICES = NS >= 4 ? NS / 2 : 1;
#endif
    constexpr bool ACTIVE = SUB >= 0;
    using TK = Task<D, ACTIVE ? SUB : 0>;
    const int lane = threadIdx.x & 63;
    const int dreal = PADDED ? b.dreal : D;
    const int npr = (dreal + 1) / 2;                      // pairs that carry data
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

    // slot -> (sample row, pair) of this lane's DMA pieces: independent of the step
    unsigned dma_off[DMA_PER_WAVE];                       // double offset within the step's samples
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int idx = wave + i * WAVES;                 // DMA instruction index within the step
        const int s = (idx < NDMA ? idx : 0) * 64 + lane; // slot
        const int n = s / PITCH, jp = s % PITCH;
        dma_off[i] = (unsigned)(n * dreal + 2 * (jp < npr ? jp : 0));   // pad slot: any valid piece
    }
    // HBM -> LDS for the NS tiles starting at tile t (clamped into the array: a clamped slot belongs
    // to a sample whose weight is zero, except the odd-D last-row case handled by the consumer)
    // (slice `part` of `nparts`: the instructions are spread over the sub-steps of the arithmetic --
    // 8 wavefronts x 10 vector-memory instructions issued back to back after a barrier fill the
    // memory pipeline's issue queue and stall every wavefront in front of its arithmetic)
    auto dma = [&](long long t, double *buf, int part, int nparts) {
        const long long tt = t < t1 ? t : t0;
        const double *__restrict__ xt = b.x + tt * 64 * dreal;               // wave-uniform
        const long long rem = total - tt * 64 * dreal - 2;                    // last legal pair start
        const unsigned lim = rem < 0 ? 0u : (rem > 0xfffffff0ll ? 0xfffffff0u : (unsigned)rem);
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int idx = wave + i * WAVES;
            if (i % nparts != part) continue;
            if (NDMA % WAVES == 0 || idx < NDMA) {        // wave-uniform
                const unsigned o = dma_off[i] < lim ? dma_off[i] : lim;
                __builtin_amdgcn_global_load_lds((gvoid_t *)(xt + o), (lvoid_t *)(buf + (size_t)idx * 128), 16,
                                                 0, 0);
            }
        }
    };

    // The weights u of the workgroup's components (a contiguous range of k in the tile-major
    // buffer) take the same route: one 1-KiB DMA piece covers two components of one tile, and the
    // block tasks of a component share it instead of each loading its own copy.
    constexpr int NSUBC = BL::NSUB;
    constexpr int UCOMP = (WAVES + NSUBC - 1) / NSUBC + 1;           // components a workgroup can touch
    constexpr int UPIECES = (UCOMP * 64 * 8 + 1023) / 1024;          // 1-KiB pieces per tile
    constexpr int UTILE = UPIECES * 128;                              // doubles per tile in LDS
    constexpr int NUDMA = NS * UPIECES;
    constexpr int UDMA_PER_WAVE = (NUDMA + WAVES - 1) / WAVES;
    double *us = xs + 2 * BUFD;                                       // 2 buffers of NS * UTILE
    const int kmin = (int)(((long long)blockIdx.x >> 3) % b.ngroups) * WAVES / NSUBC;   // first k of the group
    const long long ulen = b.ntiles * (long long)b.K * 64;            // doubles in the u buffer
    auto dma_u = [&](long long t, double *ubuf, int part, int nparts) {
#pragma unroll
        for (int i = 0; i < UDMA_PER_WAVE; ++i) {
            if (i % nparts != part) continue;
            const int idx = wave + i * WAVES;                         // piece index within the step
            if (NUDMA % WAVES == 0 || idx < NUDMA) {                  // wave-uniform
                const int q = idx / UPIECES, piece = idx % UPIECES;
                const long long tile = (t + q < t1) ? t + q : t0;     // stay inside the buffer
                const long long base = (tile * b.K + kmin) * 64 + piece * 128;
                long long o = base + 2 * lane;
                if (o > ulen - 2) o = ulen - 2;
                __builtin_amdgcn_global_load_lds((gvoid_t *)(b.u + o), (lvoid_t *)(ubuf + (size_t)idx * 128), 16,
                                                 0, 0);
            }
        }
    };

    // pipeline: step s consumes LDS buffer s%2 while the DMA of step s+1 fills the other one;
    // __syncthreads() at the end of a step waits for this wavefront's DMA (vmcnt) and orders it
    // against every wavefront's reads.
    dma(t0, xs, 0, 1);
    dma_u(t0, us, 0, 1);
    __syncthreads();
    int buf = 0;
    for (long long t = t0; t < t1; t += NS, buf ^= 1) {
        const double *xb = xs + buf * BUFD;
        const double *ub = us + buf * (NS * UTILE) + (size_t)(ACTIVE ? k - kmin : 0) * 64 + lane;
        if constexpr (!ACTIVE) {
            dma(t + NS, xs + (buf ^ 1) * BUFD, 0, 1);
            dma_u(t + NS, us + (buf ^ 1) * (NS * UTILE), 0, 1);
        }
        if constexpr (ACTIVE) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                // issue the next step's DMA in the first DMA_SLICES sub-steps only: the barrier at the
                // end of the step waits for it (vmcnt(0)), so the last slice needs time to land
                if (q < DMA_SLICES) {
                    dma(t + NS, xs + (buf ^ 1) * BUFD, q, DMA_SLICES);
                    dma_u(t + NS, us + (buf ^ 1) * (NS * UTILE), q, DMA_SLICES);
                }

                const double uraw = ub[q * UTILE];
                double u = (t + q < t1) ? uraw : 0.0;          // zero weight beyond the chunk
                // Fence: the arithmetic of this sub-step depends on `u` and so stays behind this
                // statement, and no LDS read of a LATER sub-step may move above it ("memory").
                // Without it all NS sub-steps' reads are issued up front (NS x (rows+cols) live
                // registers: spills and shuffling moves); with it at most two sub-steps overlap.
                asm volatile("" : "+v"(u) : : "memory");
                if constexpr (TK::zeroth) acc0 += u;
                const pmc_vec2 *xl = (const pmc_vec2 *)(xb + (size_t)(q * 64 + lane) * ROWD);
                // odd D: the pair holding coordinate D-1 of the array's LAST sample was fetched one
                // element early (its second half would lie outside the array)
                const bool lastrow = (D % 2 == 1 || PADDED) && (dreal % 2 == 1) &&
                                     ((t + q) * 64 + lane == b.N - 1);
                auto coord = [&](int j) -> double {       // x_j of this lane's sample
                    const pmc_vec2 v = xl[j / 2];
                    double val = (j % 2 == 0) ? v.x : v.y;
                    if ((D % 2 == 1 || PADDED) && j % 2 == 0) {
                        if (lastrow && j == dreal - 1) val = v.y;
                    }
                    if (PADDED && j >= dreal) val = 0.0;
                    return val;
                };
                double dr[TK::NR], dc[TK::NC];
#pragma unroll
                for (int i = 0; i < TK::NR; ++i) dr[i] = coord(TK::r0 + i) - pk[TK::r0 + i];
                if constexpr (TK::diag) {
#pragma unroll
                    for (int j = 0; j < TK::NC; ++j) dc[j] = dr[j];
                } else {
#pragma unroll
                    for (int j = 0; j < TK::NC; ++j) dc[j] = coord(TK::c0 + j) - pk[TK::c0 + j];
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
    extern __shared__ double xs[];                        // 2 x-buffers of NS*64 rows, then 2 u-buffers
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
        stats_run<D, PADDED, WAVES, -1>(b, xs, 0, wave, t0, t1, chunk);
        return;
    }
    bool done = false;
    static_for<0, NSUB>([&](auto S) {
        constexpr int SUBC = decltype(S)::value;
        if (!done && sub == SUBC) {
            stats_run<D, PADDED, WAVES, SUBC>(b, xs, k, wave, t0, t1, chunk);
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
    constexpr int ucomp = (SW_ + NSUB_ - 1) / NSUB_ + 1;
    constexpr size_t lds = sizeof(double) * 2 * stats_ns<D_>() *
                           (64 * Blocking<D_>::PITCH * 2 + ((ucomp * 64 * 8 + 1023) / 1024) * 128);
    if constexpr (lds > 65536) {
        static const hipError_t once = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&k_stats<D_, P_, SW_>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((k_stats<D_, P_, SW_>), dim3(grid), dim3(SW_ * 64), lds, st, b);
    return hipGetLastError();
}
