// pmc_stats.hip -- sufficient-statistics kernel of the hot path (see pmc_persample.hip for the
// execution-model notes), compiled once per sample dimension.
//
// Per component k, with d = x_n - mu_k and weight u_nk (tile-major, written by k_resp):
//     sum u | sum u d (D) | sum u d d^T (lower triangle)
//
// Arithmetic.  The second moments are rank-1 updates of a small symmetric matrix: the lower
// triangle is cut into 4 x 4 blocks (coordinate groups of 4) and every block is accumulated by
// v_mfma_f64_4x4x4_4b_f64, which multiplies four independent (4 coordinates x 4 samples) by
// (4 samples x 4 coordinates) pairs per instruction, i.e. 16 samples:
//     A[blk][i][s] = u * d_(4I+i)   lane 16 s + 4 blk + i      (layout probed on gfx950:
//     B[blk][s][j] =     d_(4J+j)   lane 16 s + 4 blk + j       scripts/microbench/mfma_probe.hip)
//     C[blk][i][j]                  lane 16 i + 4 blk + j
// A and B share one lane layout, so ONE register per coordinate group holds d for both operands,
// and the 16 accumulators of a block live in ONE register (4 batch blocks x 16 elements over the
// 64 lanes; the batch blocks are summed once at the end).  At D = 20 a component needs 15 block
// accumulators + 5 + 1 for the first moments and sum u = 42 VGPRs, against 462 for one fp64
// accumulator per lane and matrix element -- this is what lets 4 wavefronts per SIMD run where
// the per-lane (VALU) formulation was stuck at 2 and 57 % issue utilisation.  The fp64 matrix pipe
// has the same peak as the fp64 vector pipe on this chip; the gain is register pressure, issue
// slots (1 instruction per 256 FMAs) and LDS traffic, not a higher roof.  87.5 % of the MFMA slots
// carry lower-triangle elements at D = 20 (210 of 240).
//
// Work decomposition.  One wavefront owns one (component, block-row range) -- a single range
// unless D > 40 -- and streams over its chunk of samples, 16 samples per sub-step; the workgroup's
// wavefronts (different components) share the sample tile in LDS.
//
// Data path.  The sample tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, no ds_write pass), double buffered, one step (NS tiles) ahead of the arithmetic.
// LDS image: one row of PITCH 16-byte slots per sample, slot p = coordinates (2p, 2p+1), PITCH odd.
// A sub-step's ds_read_b64 of coordinate group I fetches, per 16 lanes, 4 coordinates of 4 samples
// that are TWO rows apart: 32 * PITCH mod 128 is 32 or 96, so the four 32-byte pieces fall into
// different bank octets (conflict free); the sample <-> lane assignment inside a sub-step is free
// because all 16 (batch block, s) slots are summed in the end.  All task groups of a sample chunk
// run on one XCD and share its L2 copy (measured HBM traffic = algorithmic).
#include "pmc_device.h"

namespace {

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// ---------------------------------------------------------------------------------------------
// compile-time blocking of the lower triangle
// ---------------------------------------------------------------------------------------------
template <int D> struct Blocking {
    static constexpr int NP = (D + 1) / 2;                // coordinate pairs per sample
    static constexpr int PITCH = NP | 1;                  // 16-byte slots per LDS row, odd
    static constexpr int ROWD = 2 * PITCH;                // doubles per LDS row
    static constexpr int G = (D + 3) / 4;                 // coordinate groups of 4
    static constexpr int NBLK = G * (G + 1) / 2;          // 4 x 4 blocks of the lower triangle
    // block-row ranges per component, one wavefront each (accumulator budget: <= ~60 blocks)
#ifndef PMC_STATS_MAXBLK
#define PMC_STATS_MAXBLK 50
#endif
    static constexpr int NSUB = NBLK <= PMC_STATS_MAXBLK ? 1 : (NBLK <= 90 ? 2 : 4);
    // first block row of range s: the row whose running block count is nearest to s/NSUB of all
    __host__ __device__ static constexpr int row0(int s)
    {
        if (s <= 0) return 0;
        if (s >= NSUB) return G;
        const int target = s * NBLK;                      // compare I(I+1)/2 * NSUB with s * NBLK
        int best = 0, bestdiff = target;
        for (int I = 0; I <= G; ++I) {
            int diff = I * (I + 1) / 2 * NSUB - target;
            if (diff < 0) diff = -diff;
            if (diff < bestdiff) { bestdiff = diff; best = I; }
        }
        return best;
    }
};

// Samples per pipeline step: NS tiles of 64
template <int D, int WAVES> __host__ __device__ constexpr int stats_ns()
{
#ifdef PMC_STATS_NS
    return PMC_STATS_NS;
#else
    // x rows + the u values of the workgroup's components, two buffers, within 128 KB
    constexpr int per_tile = Blocking<D>::PITCH * 16 * 64 + (WAVES / Blocking<D>::NSUB) * 64 * 8;
    return (per_tile * 4 * 2 <= 128 * 1024) ? 4 : ((per_tile * 2 * 2 <= 128 * 1024) ? 2 : 1);
#endif
}

template <int D, bool PADDED, int WAVES, int SUB>
__device__ __forceinline__ void stats_run(const PmcArgsB &b, double *xs, int k, int wave, long long t0,
                                          long long t1, int chunk)
{
    using BL = Blocking<D>;
    constexpr int STRIDE = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    constexpr int NS = stats_ns<D, WAVES>();
    constexpr int PITCH = BL::PITCH, ROWD = BL::ROWD, G = BL::G;
    constexpr int BUFD = NS * 64 * ROWD;                  // doubles per LDS buffer
    constexpr int NDMA = NS * PITCH;                      // DMA instructions per step (64 slots each)
    constexpr int DMA_PER_WAVE = (NDMA + WAVES - 1) / WAVES;
#ifdef PMC_STATS_DMA_SLICES
    constexpr int DMA_SLICES = PMC_STATS_DMA_SLICES < NS ? PMC_STATS_DMA_SLICES : NS;
#else
    constexpr int DMA_SLICES = NS >= 4 ? NS / 2 : 1;
#endif
    constexpr bool ACTIVE = SUB >= 0;
    constexpr int S_ = ACTIVE ? SUB : 0;
    constexpr int I0 = BL::row0(S_), I1 = BL::row0(S_ + 1);   // block rows [I0, I1) of this task
    constexpr int NROW = I1 - I0;
    constexpr int NACC = I1 * (I1 + 1) / 2 - I0 * (I0 + 1) / 2;
    static_assert(NROW >= 1, "empty block-row range");
    // coordinate groups that may hold coordinates >= dreal (zero padding) or the array's very last
    // element (odd dreal, see `dma`).  Exact units: the last group if D % 4 != 0.  Padded units serve
    // prev < dreal < D with prev >= 3D/4: every group reaching beyond floor(3D/4).
    constexpr int SPECIAL_FROM = PADDED ? (3 * D / 4) / 4 : (D % 4 != 0 ? G - 1 : G);
    constexpr int NSPECIAL = I1 > SPECIAL_FROM ? I1 - SPECIAL_FROM : 0;
    const int lane = threadIdx.x & 63;
    const int dreal = PADDED ? b.dreal : D;
    const int npr = (dreal + 1) / 2;                      // pairs that carry data
    const long long total = b.N * (long long)dreal;

    // lane -> (sample within the 16-sample sub-step, coordinate within the group)
    const int ci = lane & 3, blk = (lane >> 2) & 3, ks = lane >> 4;
    const int srow = 2 * blk + (ks & 1) + 8 * (ks >> 1);

    double acc0 = 0.0;
    double acc1[NROW];
    double acc2[NACC];
#pragma unroll
    for (int i = 0; i < NROW; ++i) acc1[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc2[i] = 0.0;

    // this lane's slice of the component mean (ordinary global loads, once)
    double mu[I1];
    {
        const double *pk = b.pack + (size_t)(ACTIVE ? k : 0) * STRIDE;
#pragma unroll
        for (int I = 0; I < I1; ++I) {
            const int c = 4 * I + ci;
            mu[I] = (ACTIVE && c < D) ? pk[c < D ? c : 0] : 0.0;
        }
    }
    // special groups: per-lane column (clamped into the LDS row), validity, and the odd-dreal
    // quirk of the array's very last sample: its last coordinate sits one element late in its slot,
    // because that DMA piece was fetched one element early (see `dma`)
    int scol[NSPECIAL > 0 ? NSPECIAL : 1], sfix[NSPECIAL > 0 ? NSPECIAL : 1];
    double svalid[NSPECIAL > 0 ? NSPECIAL : 1];           // 1.0 / 0.0: coordinate < dreal
    const long long lasttile = (b.N - 1) >> 6;            // wave-uniform position of sample N-1
    const int lastss = (int)((b.N - 1) & 63) >> 4, lastrow = (int)((b.N - 1) & 15);
#pragma unroll
    for (int i = 0; i < NSPECIAL; ++i) {
        const int c = 4 * (SPECIAL_FROM + i) + ci;
        scol[i] = c < ROWD - 1 ? c : ROWD - 1;
        svalid[i] = c < dreal ? 1.0 : 0.0;
        sfix[i] = ((dreal % 2 == 1) && c == dreal - 1 && srow == lastrow) ? 1 : 0;
    }

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
    // (slice `part` of `nparts`: the instructions are spread over the tiles of the arithmetic -- a
    // workgroup's vector-memory instructions issued back to back after a barrier fill the memory
    // pipeline's issue queue and stall every wavefront in front of its arithmetic)
    auto dma = [&](long long t, double *buf, int part, int nparts) {
        const long long tt = t < t1 ? t : t0;
        const double *__restrict__ xt = b.x + tt * 64 * dreal;               // wave-uniform
        // last legal pair start relative to xt; -1 if the step begins at the array's very last
        // element (dreal = 1, N = 1 mod 64): that pair then starts one element in FRONT of xt
        // (the launcher guarantees total >= 2)
        const long long rem = total - tt * 64 * dreal - 2;
        const int lim = rem > 0x7ffffff0ll ? 0x7ffffff0 : (int)rem;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int idx = wave + i * WAVES;
            if (i % nparts != part) continue;
            if (NDMA % WAVES == 0 || idx < NDMA) {        // wave-uniform
                const int o = (int)dma_off[i] < lim ? (int)dma_off[i] : lim;
                __builtin_amdgcn_global_load_lds((gvoid_t *)(xt + o), (lvoid_t *)(buf + (size_t)idx * 128), 16,
                                                 0, 0);
            }
        }
    };

    // The weights u of the workgroup's components (a contiguous range of k in the tile-major
    // buffer) take the same route: one 1-KiB DMA piece covers two components of one tile.
    constexpr int NSUBC = BL::NSUB;
    static_assert(WAVES % NSUBC == 0, "block-row ranges must divide the workgroup");
    constexpr int UCOMP = WAVES / NSUBC;                              // components of a workgroup
    constexpr int UPIECES = (UCOMP * 64 * 8 + 1023) / 1024;          // 1-KiB pieces per tile
    constexpr int UTILE = UPIECES * 128;                              // doubles per tile in LDS
    constexpr int NUDMA = NS * UPIECES;
    constexpr int UDMA_PER_WAVE = (NUDMA + WAVES - 1) / WAVES;
    double *us = xs + 2 * BUFD;                                       // 2 buffers of NS * UTILE
    const int kmin = (int)(((long long)blockIdx.x >> 3) % b.ngroups) * UCOMP;   // first k of the group
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
    dma_barrier();
    int buf = 0;
    for (long long t = t0; t < t1; t += NS, buf ^= 1) {
        const double *xb = xs + buf * BUFD + (size_t)srow * ROWD;     // + this lane's sample row
        // special groups: per-lane bases, so that every sub-step's read is base + immediate offset
        const double *xsp[NSPECIAL > 0 ? NSPECIAL : 1], *xspfix[NSPECIAL > 0 ? NSPECIAL : 1];
#pragma unroll
        for (int i = 0; i < NSPECIAL; ++i) {
            xsp[i] = xb + scol[i];
            xspfix[i] = xsp[i] + sfix[i];
        }
        const double *ub = us + buf * (NS * UTILE) + (size_t)(ACTIVE ? k - kmin : 0) * 64 + srow;
        if constexpr (!ACTIVE) {
            dma(t + NS, xs + (buf ^ 1) * BUFD, 0, 1);
            dma_u(t + NS, us + (buf ^ 1) * (NS * UTILE), 0, 1);
        }
        if constexpr (ACTIVE) {
            // Sub-steps of 16 samples; the LDS reads of sub-step s+1 are issued before the arithmetic
            // of sub-step s (register double buffer), so their latency hides behind ~15 MFMAs.  Only
            // the first sub-step after the barrier waits for its operands.
            constexpr int NSUBSTEP = NS * 4;
            double xc[I1], xn[I1], uc, un = 0.0;
            auto fetch = [&](auto S, double (&xx)[I1], double &uu) {
                constexpr int s = decltype(S)::value, q = s / 4, ss = s % 4;
                constexpr int ROW = (q * 64 + ss * 16) * ROWD;
                uu = ub[q * UTILE + ss * 16];
                const bool fixnow = (D % 2 == 1 || PADDED) && (t + q == lasttile) && (ss == lastss);
#pragma unroll
                for (int I = 0; I < I1; ++I) {
                    if (I < SPECIAL_FROM) {
                        xx[I] = xb[ROW + 4 * I + ci];
                    } else {
                        const double *xp = fixnow ? xspfix[I - SPECIAL_FROM] : xsp[I - SPECIAL_FROM];
                        xx[I] = xp[ROW];
                    }
                }
            };
            fetch(std::integral_constant<int, 0>{}, xc, uc);
            static_for<0, NSUBSTEP>([&](auto S) {
                constexpr int s = decltype(S)::value, q = s / 4;
                // issue the next step's DMA in the first DMA_SLICES tiles only: the barrier at the
                // end of the step waits for it (vmcnt(0)), so the last slice needs time to land
                if constexpr (s % 4 == 0 && q < DMA_SLICES) {
                    dma(t + NS, xs + (buf ^ 1) * BUFD, q, DMA_SLICES);
                    dma_u(t + NS, us + (buf ^ 1) * (NS * UTILE), q, DMA_SLICES);
                }
                // Scheduling fences (no instructions).  The first pins this point behind the previous
                // sub-step's arithmetic; the next ones make this sub-step's operands "appear" here, so
                // nothing that consumes them is hoisted to right behind their loads (which would wait
                // for the LDS in front of the previous sub-step's MFMAs); the last one -- on the weight
                // every product below depends on -- keeps the prefetch of sub-step s+1 in front of the
                // arithmetic of sub-step s.
#pragma unroll
                for (int i = 0; i < NACC; ++i) asm volatile("" : "+v"(acc2[i]) : : "memory");
#pragma unroll
                for (int I = 0; I < I1; ++I) asm volatile("" : "+v"(xc[I]));
                if constexpr (s + 1 < NSUBSTEP) fetch(std::integral_constant<int, s + 1>{}, xn, un);
                asm volatile("" : "+v"(uc) : : "memory");       // every MFMA below depends on uc
                const double u = (t + q < t1) ? uc : 0.0;       // tiles beyond the chunk: no weight
                double d[I1];
#pragma unroll
                for (int I = 0; I < I1; ++I) {
                    // padding coordinates: whatever finite value the slot holds, times 0 (a select
                    // here makes the compiler keep every sub-step's operands live)
                    d[I] = I < SPECIAL_FROM ? xc[I] - mu[I] : (xc[I] - mu[I]) * svalid[I - SPECIAL_FROM];
                }
                if constexpr (SUB == 0) acc0 += u;
                int a = 0;
#pragma unroll
                for (int I = I0; I < I1; ++I) {
                    const double ud = u * d[I];
                    acc1[I - I0] += ud;
#pragma unroll
                    for (int J = 0; J <= I; ++J, ++a)
                        acc2[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(ud, d[J], acc2[a], 0, 0, 0);
                }
                if constexpr (s + 1 < NSUBSTEP) {
#pragma unroll
                    for (int I = 0; I < I1; ++I) xc[I] = xn[I];
                    uc = un;
                }
            });
        }
        dma_barrier();
    }

    if constexpr (ACTIVE) {
        // accumulator lane layouts: acc2 -- lane 16 i + 4 blk + j; acc0 / acc1 -- per (sample, ci)
        double *out = b.partials + ((size_t)chunk * b.K + k) * PS;
        if constexpr (SUB == 0) {
            double s0 = acc0;                              // every sample appears in 4 lanes (ci)
            s0 += __shfl_xor(s0, 4, 64);
            s0 += __shfl_xor(s0, 8, 64);
            s0 += __shfl_xor(s0, 16, 64);
            s0 += __shfl_xor(s0, 32, 64);
            if (lane == 0) out[0] = s0;
        }
        int a = 0;
#pragma unroll
        for (int I = I0; I < I1; ++I) {
            double m = acc1[I - I0];
            m += __shfl_xor(m, 4, 64);
            m += __shfl_xor(m, 8, 64);
            m += __shfl_xor(m, 16, 64);
            m += __shfl_xor(m, 32, 64);
            if (lane < 4 && 4 * I + lane < D) out[1 + 4 * I + lane] = m;
#pragma unroll
            for (int J = 0; J <= I; ++J, ++a) {
                double v = acc2[a];
                v += __shfl_xor(v, 4, 64);                 // sum of the 4 batch blocks
                v += __shfl_xor(v, 8, 64);
                const int gi = 4 * I + (lane >> 4), gj = 4 * J + (lane & 3);
                if (blk == 0 && gj <= gi && gi < D) out[1 + D + gi * (gi + 1) / 2 + gj] = v;
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
    if (b.ctl && b.ctl[PMC_CTL_REDO] == 0) return;        // the common-shift form was accurate: nothing to redo
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


// =============================================================================================
// Component x monomial form (k_stats_gemm): the statistics as ONE matrix product per sample block
// =============================================================================================
// With a shift c common to all components the three sums are U^T Z:  U[n][k] = u_nk,  Z[n] = (1 | d | d_i d_j, j <= i),
// d = x_n - c  -- the lower triangle of d~ d~^T for the augmented vector d~ = (d, 1), M = (D+1)(D+2)/2 monomials in
// the order of the statistics vector.  On v_mfma_f64_16x16x4_f64: A = 16 components x 4 samples of u (no product
// u * d any more), B = 4 samples x 16 monomials, formed with ONE v_mul_f64 per lane from two LDS reads; a B operand
// feeds both 16-component row blocks of a 32-component group.  Against the per-component-shift kernel above: no
// per-component x - mu_k, no u * d, no first-moment adds (16 vector operations per 15 MFMAs there), 96-100 % of the
// matrix slots useful (231 of 240 columns at D = 20, 861 of 864 at D = 40; there 87.5 %), and the instruction is the
// one that holds the highest clock.  Measured (scripts/microbench/stats_gemm.hip, the prototype):
// D = 20, K = 32: 57 algorithmic TFLOP/s (0.73 of the fp64 peak) against 45-47; D = 40, K = 128: 67 (0.855) against 51.
//
// The price is numerical: moments about c instead of mu_k lose (|mu_k - c| / sigma_k)^2 leading parts when they are
// re-centred.  The finishing kernels (pmc_api.hip) re-centre on the device, test every component a posteriori
// (|mean - c|^2 <= limit * variance in every coordinate) and raise ctl[PMC_CTL_REDO] if one fails: the kernel above
// then runs as before, otherwise it returns at once.
//
// Work decomposition.  Workgroup = (sample chunk, group of 32 components, column super group); its wavefronts are
// CGW column groups x SL sample slices: a wavefront owns C column tiles (16 monomials each) x 2 row blocks = 2 C
// accumulator tiles (8 registers each) and every SL-th quarter of each sample tile.  LDS holds NS sample tiles, double
// buffered: rows of x - c (the subtraction happens once, on the way from the registers the rows were prefetched into;
// the "1" of d~ sits in a spare slot of each row that the staging never touches) and the u values by LDS-DMA in
// 1-KiB pieces of two components.  A lane (n = lane & 15, g = lane >> 4) works on sample row 8 g + t of a tile in
// MFMA step t: rows 8 apart are 16 doubles apart modulo the 32-double bank cycle (row pitch = odd number of 16-byte
// slots), so the two halves of a wavefront read disjoint banks; the 16 lanes of a row read two of its columns each.
template <int D> struct GemmCfg {
    // column tiles per wavefront | column groups per workgroup | sample slices | tiles per pipeline step
    // (2 C accumulator tiles of 8 registers each have to fit next to ~60 operand / address registers: C = 8 at D = 30
    // and C = 9 at D = 32 / 64 spilled)
#ifdef PMC_GEMM_C                                          // tuning overrides (scripts/tune_unit.sh)
    static constexpr int C = PMC_GEMM_C, CGW = PMC_GEMM_CGW, SL = PMC_GEMM_SL, NS = PMC_GEMM_NS;
#else
    // (D = 24: 325 monomials = 21 column tiles = 3 x 7 -- the 6 x 4 = 24 of rounds 3-4 left an eighth of the matrix slots
    //  idle: 3.19 -> 3.04 ms per 4e6 x 64, profiles/r05_d24_sweep.txt; 7 x 3 needs 112 accumulator registers and is slower)
    static constexpr int C = D <= 8 ? 3 : (D <= 10 ? 5 : (D <= 12 ? 3 : (D <= 20 ? 5 : (D <= 24 ? 3 : (D <= 30 ? 4 :
                             (D <= 32 ? 3 : (D <= 40 ? 7 : (D <= 48 ? 5 : 6))))))));
    static constexpr int CGW = D <= 10 ? 1 : (D <= 16 ? 2 : (D <= 20 ? 3 : (D <= 24 ? 7 : (D <= 30 ? 8 : (D <= 32 ? 12 : 8)))));
    static constexpr int SL = D <= 10 ? 8 : (D <= 20 ? 4 : (D <= 24 ? 2 : 1));
    static constexpr int NS = D <= 32 ? 2 : 1;
#endif
    static constexpr bool ENABLED = D >= 8;
    static constexpr int W = CGW * SL;
    static constexpr int NP = (D + 1) / 2;                 // coordinate pairs per sample
    static constexpr int PITCH = (NP + 1) | 1;             // 16-byte slots per LDS row: odd, one spare for the "1"
    static constexpr int ROWD = 2 * PITCH;
    static constexpr int M = (D + 1) * (D + 2) / 2;
    static constexpr int NT = (M + 15) / 16;
    static constexpr int MSP = NT * 16;                    // doubles per component in the partial sums
    static constexpr int NCS = (NT + C * CGW - 1) / (C * CGW);
    static constexpr int XT = 64 * ROWD;                   // doubles per x tile
    static constexpr int UPIECE = 130;                     // a 1-KiB DMA piece (2 components x 64 samples) + 16 bytes
    static constexpr int UT = 17 * UPIECE;                 // doubles per u tile: 32 components + the piece of the factors
                                                           // k_resp_groups left to be applied (2 row blocks x 64 samples)
    static constexpr size_t LDS_BYTES = sizeof(double) * (2 * NS * (XT + UT) + 64);     // + the common shift
    // workgroups that share a CU (LDS; their wavefronts' registers fit next to each other up to 3 per SIMD): two
    // that run out of step hide each other's barriers and pipeline refills
#ifdef PMC_GEMM_WGS
    static constexpr int WGS_PER_CU = PMC_GEMM_WGS;
#else
    static constexpr int WGS_PER_CU = (LDS_BYTES <= 80 * 1024 && W <= 6) ? 2 : 1;
#endif
};

// sample row (within a tile, before the lane's 8 g) of MFMA step j of a slice: compile-time part ...
template <int SL> __host__ __device__ constexpr int gemm_row_imm(int j) { return SL == 1 ? (j & 7) + 32 * (j >> 3) : j; }
// ... and the slice's run-time part
template <int SL> __device__ __forceinline__ int gemm_row_base(int sl)
{
    if constexpr (SL == 1) return 0;
    else if constexpr (SL == 2) return 32 * sl;
    else if constexpr (SL == 4) return 4 * (sl & 1) + 32 * (sl >> 1);
    else if constexpr (SL == 8) return 2 * (sl & 3) + 32 * (sl >> 2);
    else return (sl & 7) + 32 * (sl >> 3);
}

// LDS reads of the inner loop, written out: with an explicit 16-bit offset (the compiler pairs ordinary reads into
// ds_read2_b64, whose 8-bit offsets need a base register every 2 KB -- dozens, spilled), waited for by lds_wait,
// which hands the values on so that nothing consumes them earlier.
template <int IMM> __device__ __forceinline__ void lds_read64(double &v, unsigned addr)
{
    static_assert(IMM >= 0 && IMM < 65536 && IMM % 8 == 0, "ds_read_b64 offset");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
}
template <int N> __device__ __forceinline__ void lds_wait(double (&v)[N])
{
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]));
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]));
    else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    else if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]));
    else if constexpr (N == 6)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
    else if constexpr (N == 7)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]));
    else {
        static_assert(N == 8, "lds_wait: up to 8 values");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    }
}

typedef double gd4 __attribute__((ext_vector_type(4)));
typedef double gd2 __attribute__((ext_vector_type(2)));
typedef double gd2u __attribute__((ext_vector_type(2), aligned(8)));

// NRB = row blocks of 16 components that hold any (the last group of a K that is not a multiple of 32 may have one)
// SCALED: u comes from k_resp_groups and is still to be multiplied by a factor per (sample, row block)
template <int D, bool PADDED, int NRB, bool SCALED>
__device__ __forceinline__ void stats_gemm_run(const PmcArgsG &b, double *xs)
{
    using CF = GemmCfg<D>;
    constexpr int C = CF::C, CGW = CF::CGW, SL = CF::SL, NS = CF::NS, W = CF::W, R = NRB;
    constexpr int NP = CF::NP, ROWD = CF::ROWD, XT = CF::XT, UT = CF::UT, UPIECE = CF::UPIECE;
    constexpr int BUFX = NS * XT, BUFU = NS * UT;
    constexpr int JN = 16 / SL, NSTEP = NS * JN;
    constexpr int PX = NS * 64 * NP, NPX = (PX + W * 64 - 1) / (W * 64);
    constexpr int PU = NS * 16, NPU = (PU + W - 1) / W;
    double *us = xs + 2 * BUFX;                            // xs: 2 x buffers, then 2 u buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % CGW, sl = wave / CGW;
    const int n16 = lane & 15, g = lane >> 4;
    const int dreal = PADDED ? b.dreal : D;
    const int npr = (dreal + 1) / 2, ONE = 2 * npr;        // column of the "1": first spare slot behind the data
    const int M = (dreal + 1) * (dreal + 2) / 2;
    const long long total = b.N * (long long)dreal;

    // block -> (chunk, component group, column super group); all blocks of a chunk on one XCD (one L2 copy of x)
    const int bid = blockIdx.x, qb = bid >> 3;
    const int nsub = b.ngroups * b.ncs;
    const int chunk = (bid & 7) + 8 * (qb / nsub);
    const int sub = qb % nsub, group = sub / b.ncs, cs = sub % b.ncs;
    const int kmin = group * 32;
    const long long t0 = (long long)chunk * b.tiles_per_chunk;
    long long t1 = t0 + b.tiles_per_chunk;
    if (t1 > b.ntiles) t1 = b.ntiles;

    // The plan, made by every workgroup for itself (K x D numbers out of L2; no launch of its own): the common
    // shift c = midrange of the component means per coordinate (it minimises the largest |mu_k - c|), and the
    // a-priori test with the pack's own scale when the caller knows what the pack describes (kind >= 0): with
    // precision = R^T R, 1 / R_ii^2 <= Sigma_ii, so (mu_ki - c_i)^2 R_ii^2 [* nu_k for the VB kind, whose W is the
    // precision / nu] > limit says "too far apart" -- conservatively; whoever passes is tested again a posteriori,
    // on the data, by k_gemm_convert.  Workgroup 0 publishes c and the decision for the finishing kernels.
    double *cen = us + 2 * BUFU;                           // 64 doubles behind the buffers; scratch: the u buffers
#ifdef PMC_AB_NOPLAN
    // A/B switch, TIMING ONLY (wrong numbers unless the means straddle 0; scripts/stats_plan_ab.sh): no plan -- what the
    // K x D numbers every workgroup reads and reduces before its first tile cost per launch
    if (tid < 64) cen[tid] = 0.0;
    if (blockIdx.x == 0 && tid == 0) { b.ctl[PMC_CTL_GO] = 1; b.ctl[PMC_CTL_REDO] = 0; }
    if (blockIdx.x == 0 && tid < dreal) b.center[tid] = 0.0;
    __syncthreads();
#else
    {
        constexpr int STRIDE = pmc_pack_stride_c(D);
        double *lo = us, *hi = us + W * 64;
        int *farflag = (int *)(us + 2 * W * 64);
        if (tid == 0) *farflag = 0;
        const int j = lane < dreal ? lane : 0;
        double l = b.spack[(size_t)(wave < b.K ? wave : 0) * STRIDE + j], h = l;
        for (int k = wave + W; k < b.K; k += W) {
            const double m = b.spack[(size_t)k * STRIDE + j];
            l = m < l ? m : l;
            h = m > h ? m : h;
        }
        lo[wave * 64 + lane] = l;
        hi[wave * 64 + lane] = h;
        __syncthreads();
        if (tid < 64) {
            const int wmax = b.K < W ? b.K : W;
            for (int w = 1; w < wmax; ++w) {
                l = lo[w * 64 + lane] < l ? lo[w * 64 + lane] : l;
                h = hi[w * 64 + lane] > h ? hi[w * 64 + lane] : h;
            }
            const double c = 0.5 * l + 0.5 * h;
            cen[lane] = (c == c && fabs(c) <= 1.7976931348623157e308) ? c : 0.0;
        }
        __syncthreads();
        if (b.kind >= 0 && pmc_engine(D) != PMC_ENG_DPP) {
            // R upper triangular, packed row-major over the compiled dimension: R_ii at D + i D - i (i - 1) / 2
            bool far = false;
            for (int idx = tid; idx < b.K * dreal; idx += 64 * W) {
                const int k = idx / dreal, i = idx - k * dreal;
                const double *pk = b.pack + (size_t)k * STRIDE;
                const double rii = pk[D + i * D - i * (i - 1) / 2];
                double s = rii * rii;
                if (b.kind == PMC_KIND_VB) s *= pk[D + pmc_tri(D) + 1];          // c1 = nu_k
                const double dlt = b.spack[(size_t)k * STRIDE + i] - cen[i];
                far = far || dlt * dlt * s > b.limit_prior;
            }
            if (far) *farflag = 1;                         // (benign race: every writer stores 1)
        }
        __syncthreads();
        const int isfar = *farflag;
        if (blockIdx.x == 0) {
            if (tid < dreal) b.center[tid] = cen[tid];
            if (tid == 0) {
                b.ctl[PMC_CTL_GO] = isfar ? 0 : 1;
                b.ctl[PMC_CTL_REDO] = isfar ? 1 : 0;
            }
        }
        if (isfar) return;
    }
#endif

    // this lane's two factors of each of the wavefront's column tiles: LDS offsets (doubles) incl. the lane's rows.
    // Monomial m in the order of the statistics vector: 0 -> 1 * 1, 1 + j -> 1 * d_j, 1 + D + i(i+1)/2 + j -> d_i * d_j.
    const int rowbase = (8 * g + gemm_row_base<SL>(sl)) * ROWD;
    int off1[C], off2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int ct = (cs * CGW + cg) * C + c;
        const int m = 16 * ct + n16;
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
#ifdef PMC_AB_NOCONFLICT
        // A/B switch, TIMING ONLY (wrong numbers; scripts/stats_conflict_ab.sh, verdict r5 #3): another address pattern of the
        // B-operand reads -- sixteen different coordinates per column tile instead of the monomials' factors.  It was meant to
        // remove the bank conflicts and turned out to have MORE (conflict fraction 0.21 -> 0.33, the rows of the four lane
        // groups collide) -- at the same kernel time, 2.62 -> 2.61 ms: the LDS reads hide behind the matrix pipe (84 % busy,
        // LDS 20-24 %); the conflicts are off the critical path (profiles/r06_stats_conflict_ab.txt).
        i1 = n16 % (dreal > 0 ? dreal : 1);
        i2 = (n16 + 7) % (dreal > 0 ? dreal : 1);
#endif
        off1[c] = rowbase + i1;
        off2[c] = rowbase + i2;
    }
    // ... as LDS byte addresses (buffer 0; the buffer and the step go into the reads' immediates)
    const unsigned xs_addr = (unsigned)(uintptr_t)(lvoid_t *)xs, us_addr = (unsigned)(uintptr_t)(lvoid_t *)us;
    unsigned a1[C], a2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        a1[c] = xs_addr + 8u * (unsigned)off1[c];
        a2[c] = xs_addr + 8u * (unsigned)off2[c];
    }
    int uoff[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int cc = 16 * r + n16;
        // (the four 8-row groups of a 32-row half sit in the order 0, 2, 1, 3 in the LDS image of u -- see udma --
        // so that the two groups a half-wavefront reads together are 32 banks apart)
        uoff[r] = (cc >> 1) * UPIECE + (cc & 1) * 64 + 8 * (((g & 1) << 1) | (g >> 1)) + gemm_row_base<SL>(sl);
    }
    unsigned au[R];
#pragma unroll
    for (int r = 0; r < R; ++r) au[r] = us_addr + 8u * (unsigned)uoff[r];
    // the factor of (row block r, this lane's sample): piece 16 of a tile, natural sample order; the 16 lanes of a
    // sample read one address
    const unsigned af = us_addr + 8u * (unsigned)(16 * UPIECE + 8 * g + gemm_row_base<SL>(sl));

    // the "1" of every row, both buffers (never overwritten: the staging writes data slots only)
    for (int row = tid; row < 2 * NS * 64; row += 64 * W) xs[row * ROWD + ONE] = 1.0;

    // x staging: piece = one coordinate pair of one row, fixed per thread; global -> registers one step ahead,
    // minus c on the way into LDS
    // (row and pair of a piece are recomputed where they are needed and c comes from LDS again: the registers are
    // better spent on accumulators)
    if (tid == 0 && (dreal & 1)) cen[dreal] = 0.0;         // second half of an odd dimension's last pair
    __syncthreads();                                       // the plan's scratch is free: the u buffers may be filled
    auto piece = [&](int i, int &n, int &jp) {
        const int id = tid + i * W * 64;
        n = id / NP;
        jp = id - n * NP;
        return id < PX && jp < npr;
    };
    gd2 xv[NPX];
    // rows beyond the chunk or the array are clamped into it (their weights are zero); the array's very last
    // element of an odd-sized array is fetched one element early and picked from the pair's second half
    auto xload = [&](long long t) {
        // (uniform 64-bit base + 32-bit lane offset: as 64-bit per-lane addresses these cost register pairs that spilled)
        const long long tt = t < t1 ? t : t0;
        const double *__restrict__ xt = b.x + tt * 64 * dreal;                // wave-uniform
        const long long rem = total - tt * 64 * dreal - 2;                   // last legal pair start relative to xt
        const int lim = rem > 0x7ffffff0ll ? 0x7ffffff0 : (int)rem;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int n, jp;
            piece(i, n, jp);
            const int o = n * dreal + 2 * jp;
            xv[i] = *(const gd2u *)(xt + (o < lim ? o : lim));
        }
    };
    auto xstore = [&](long long t, double *xbuf) {
        const long long tt = t < t1 ? t : t0;
        const long long base = tt * 64 * dreal;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int n, jp;
            if (piece(i, n, jp)) {
                const bool last = base + (n * dreal + 2 * jp) == total - 1;
                const gd2 cc = *(const gd2 *)(cen + 2 * jp);
                gd2 v;
                v[0] = (last ? xv[i][1] : xv[i][0]) - cc[0];
                v[1] = xv[i][1] - cc[1];
                *(gd2 *)(xbuf + n * ROWD + 2 * jp) = v;
            }
        }
    };
    const long long ulen = b.ntiles * (long long)b.K * 64;
    const int gtot = (b.K + PMC_RESP_GROUP - 1) / PMC_RESP_GROUP;
    const long long glen = b.ntiles * (long long)gtot * 64;
    auto fdma = [&](long long t, double *ubuf) {           // the factors of the step's tiles: wavefront q takes tile q
        if constexpr (SCALED) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (wave != q % W) continue;               // wave-uniform
                if (t + q >= t1) {
                    *(gd2 *)(ubuf + q * UT + 16 * UPIECE + 2 * lane) = gd2{0.0, 0.0};
                    continue;
                }
#ifdef PMC_AB_CONST_U
                long long o = (0 * gtot + 2 * group) * 64;
#else
                long long o = ((t + q) * gtot + 2 * group) * 64;                     // wave-uniform
#endif
                // the array's very last group has no successor: its second half re-reads the group itself (rows discarded)
                const int gl = o >= glen - 64 ? (2 * lane) & 63 : 2 * lane;
                if (o > glen - 64) o = glen - 64;
                __builtin_amdgcn_global_load_lds((gvoid_t *)(b.gscale + o + gl), (lvoid_t *)(ubuf + q * UT + 16 * UPIECE), 16, 0, 0);
            }
        }
    };
    const int ulane = (lane >> 5) * 64 + 2 * (((lane & 31) & 0x13) | ((lane & 4) << 1) | ((lane & 8) >> 1));
    auto udma = [&](long long t, double *ubuf) {
        fdma(t, ubuf);
#pragma unroll
        for (int i = 0; i < NPU; ++i) {
            const int id = wave + i * W;
            if (PU % W == 0 || id < PU) {                  // wave-uniform
                const int q = id / 16, p = id % 16;
                if (t + q >= t1) {                         // tile beyond the chunk: no weight (wave-uniform, last step only)
                    *(gd2 *)(ubuf + q * UT + p * UPIECE + 2 * lane) = gd2{0.0, 0.0};
                    continue;
                }
#ifdef PMC_AB_CONST_U                                     // (A/B switch, timing only: u from ONE tile, i.e. out of L2)
                const long long tile = 0;
#else
                const long long tile = t + q;
#endif
                // lane -> 16-byte chunk (component 2 p + (lane >> 5), sample pair lane & 31 with bits 2 and 3 swapped)
                long long o = (tile * b.K + kmin + 2 * p) * 64;                      // wave-uniform
                // components beyond K read the next tile's (finite values, rows discarded); past the array's very
                // last component the pair's second half re-reads that component itself
                const int ul = o >= ulen - 64 ? ulane & 63 : ulane;
                if (o > ulen - 64) o = ulen - 64;
                __builtin_amdgcn_global_load_lds((gvoid_t *)(b.u + o + ul), (lvoid_t *)(ubuf + q * UT + p * UPIECE), 16, 0, 0);
            }
        }
    };

    gd4 acc[R][C];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[r][c] = gd4{0.0, 0.0, 0.0, 0.0};

    xload(t0);
    udma(t0, us);
    xstore(t0, xs);
    dma_barrier();
    // The per-lane address registers are moved from one LDS buffer to the other IN PLACE at the end of a step (as
    // ordinary pointers into a run-time buffer the compiler kept both sets and spilled accumulators).
    int buf = 0;
    unsigned dx = 8u * (unsigned)BUFX, du = 8u * (unsigned)BUFU;
    unsigned af_cur = af;
    for (long long t = t0; t < t1; t += NS, buf ^= 1) {
        xload(t + NS);
        udma(t + NS, us + (buf ^ 1) * BUFU);

        double ac[R], zc[C], an[R], f1n[C], f2n[C], fn[SCALED ? R : 1];
        auto fetch = [&](auto IDX) {
            constexpr int idx = decltype(IDX)::value, q = idx / JN, j = idx % JN;
            constexpr int XIMM = 8 * ((q * 64 + gemm_row_imm<SL>(j)) * ROWD);
            constexpr int UIMM = 8 * (q * UT + gemm_row_imm<SL>(j));
            static_for<0, R>([&](auto RR) { lds_read64<UIMM>(an[decltype(RR)::value], au[decltype(RR)::value]); });
            if constexpr (SCALED)
                static_for<0, R>([&](auto RR) { lds_read64<UIMM + 8 * 64 * decltype(RR)::value>(fn[decltype(RR)::value], af_cur); });
            static_for<0, C>([&](auto CC) {
                lds_read64<XIMM>(f1n[decltype(CC)::value], a1[decltype(CC)::value]);
                lds_read64<XIMM>(f2n[decltype(CC)::value], a2[decltype(CC)::value]);
            });
        };
        auto arrive = [&]() {                              // the operands fetch() asked for: wait, form B
            lds_wait(an);
            lds_wait(f1n);
            lds_wait(f2n);
            if constexpr (SCALED) lds_wait(fn);
#pragma unroll
            for (int r = 0; r < R; ++r) ac[r] = SCALED ? an[r] * fn[r] : an[r];
#pragma unroll
            for (int c = 0; c < C; ++c) zc[c] = f1n[c] * f2n[c];
        };
        fetch(ic<0>{});
        arrive();
        static_for<0, NSTEP>([&](auto IDX) {
            constexpr int idx = decltype(IDX)::value;
            // operands of step idx + 1 are read in front of the multiplies of step idx
            if constexpr (idx + 1 < NSTEP) fetch(ic<idx + 1>{});
            __builtin_amdgcn_sched_barrier(0);             // (... really in front: into registers of their own)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < C; ++c) acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[r], zc[c], acc[r][c], 0, 0, 0);
            // (the wait for the next operands stays behind this step's multiplies, and nothing of the step after
            // next is hoisted into this one)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (idx + 1 < NSTEP) arrive();
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int c = 0; c < C; ++c) {
            a1[c] += dx;
            a2[c] += dx;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) au[r] += du;
        af_cur += du;
        dx = 0u - dx;
        du = 0u - du;
        xstore(t + NS, xs + (buf ^ 1) * BUFX);
        dma_barrier();
    }

    // The SL sample slices of a column group add their accumulators through LDS (the tile buffers are free: the
    // loop's last barrier is behind every read and every copy), slice by slice in ascending order -- a fixed order,
    // so the sums stay bit-reproducible -- and slice 0 publishes ONE partial vector per chunk.
    if constexpr (SL > 1) {
        static_assert((size_t)CGW * 2 * C * 256 * sizeof(double) <= CF::LDS_BYTES, "slice reduction does not fit the LDS");
        double *red = xs + (size_t)cg * (R * C * 256) + lane;
        for (int s = 1; s < SL; ++s) {
            if (sl == s) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) red[((r * C + c) * 4 + reg) * 64] = acc[r][c][reg];
            }
            __syncthreads();
            if (sl == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) acc[r][c][reg] += red[((r * C + c) * 4 + reg) * 64];
            }
            __syncthreads();
        }
        if (sl != 0) return;
    }
    // accumulator layout of v_mfma_f64_16x16x4_f64: D[row = (lane >> 4) + 4 reg][col = lane & 15]
    double *out = b.partials + ((size_t)chunk * b.K) * CF::MSP;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int ct = (cs * CGW + cg) * C + c;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int k = kmin + 16 * r + g + 4 * reg;
                if (k < b.K && ct < CF::NT) out[(size_t)k * CF::MSP + 16 * ct + n16] = acc[r][c][reg];
            }
        }
    }
}

template <int D, bool PADDED>
__global__ __launch_bounds__(64 * GemmCfg<D>::W) void k_stats_gemm(const PmcArgsG b)
{
    extern __shared__ double xs[];
    const int nsub = b.ngroups * b.ncs;
    const int group = ((blockIdx.x >> 3) % nsub) / b.ncs;
    if (b.gscale != nullptr) {
        if (b.K - group * 32 > 16) stats_gemm_run<D, PADDED, 2, true>(b, xs);
        else stats_gemm_run<D, PADDED, 1, true>(b, xs);
    } else {
        if (b.K - group * 32 > 16) stats_gemm_run<D, PADDED, 2, false>(b, xs);
        else stats_gemm_run<D, PADDED, 1, false>(b, xs);
    }
}

constexpr int NSUB_ = Blocking<D_>::NSUB;
#ifdef PMC_STATS_WAVES                                     // tuning override (scripts/tune_stats.sh)
constexpr int SW_ = PMC_STATS_WAVES;
#else
// wavefronts per statistics workgroup: 16 (4 per SIMD) while a task fits 128 VGPRs under that
// launch bound, else 8 -- two such workgroups still share a CU when registers and LDS allow
// (measured at D = 24: 8 -> 1.68 ms, 12 -> 2.68 ms, 16 (spills) -> 2.47 ms per 4e6 samples, K = 32)
constexpr int SW_ = Blocking<D_>::G <= 5 ? 16 : 8;
#endif

}  // namespace

// launch geometry knobs the dispatcher needs
extern "C" void PMC_UNIT_NAME_X(pmc_stats_config_d, PMC_D, PMC_PADDED)(int *nsub, int *waves)
{
    *nsub = NSUB_;
    *waves = SW_;
}

// geometry of the component x monomial form: monomial tiles per workgroup (0: this dimension has no such kernel),
// sample slices (partial statistics vectors per chunk), doubles per component in a partial vector
extern "C" void PMC_UNIT_NAME_X(pmc_stats_gemm_config_d, PMC_D, PMC_PADDED)(int *cols_per_wg, int *slices, int *msp,
                                                                           int *wgs_per_cu)
{
    using CF = GemmCfg<D_>;
    *cols_per_wg = CF::ENABLED ? CF::C * CF::CGW : 0;
    *slices = 1;                                           // (the sample slices are summed inside the kernel)
    *msp = CF::MSP;
    *wgs_per_cu = CF::WGS_PER_CU;
}

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_stats_gemm_d, PMC_D, PMC_PADDED)(const PmcArgsG &b, unsigned grid,
                                                                                 hipStream_t st)
{
    using CF = GemmCfg<D_>;
    if constexpr (!CF::ENABLED) return hipErrorInvalidValue;
    else {
        static_assert(CF::LDS_BYTES <= 160 * 1024, "k_stats_gemm tile buffers exceed the LDS");
        static_assert(16 % CF::SL == 0 && CF::W <= 16, "k_stats_gemm slicing");
        const hipError_t once = PMC_SET_LDS_PER_DEVICE((&k_stats_gemm<D_, P_>), CF::LDS_BYTES);
        if (once != hipSuccess) return once;
        hipLaunchKernelGGL((k_stats_gemm<D_, P_>), dim3(grid), dim3(64 * CF::W), CF::LDS_BYTES, st, b);
        return hipGetLastError();
    }
}

extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_stats_d, PMC_D, PMC_PADDED)(const PmcArgsB &b, unsigned grid,
                                                                            hipStream_t st)
{
    constexpr int ucomp = SW_ / NSUB_;
    constexpr size_t lds = sizeof(double) * 2 * stats_ns<D_, SW_>() *
                           (64 * Blocking<D_>::PITCH * 2 + ((ucomp * 64 * 8 + 1023) / 1024) * 128);
    static_assert(lds <= 160 * 1024, "statistics tile buffers exceed the LDS");
    if constexpr (lds > 65536) {
        const hipError_t once = PMC_SET_LDS_PER_DEVICE((&k_stats<D_, P_, SW_>), lds);
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((k_stats<D_, P_, SW_>), dim3(grid), dim3(SW_ * 64), lds, st, b);
    return hipGetLastError();
}
