// pmc_big.hip -- the run-time-dimension unit: sample dimensions beyond the compiled ones (PMC_MAX_DIM < D <=
// PMC_BIG_MAX_DIM).  The reference's loops take a vector of any length (pypmc/tools/_linalg.pyx:32-37); the
// compiled units keep a sample's coordinates and a component's accumulators in registers, which ends at D = 64.
// Up there the work per (sample, component) pair is D^2/2 multiply-adds against one exp, so the path is cut where
// it is GEMM-shaped and the rest reuses the per-sample kernels as they are:
//
//   k_big_maha   maha_nk = |R_k (x_n - mu_k)|^2 for all pairs, tile-major -- 16 x 16 x 4 fp64 MFMA: the triangular
//                product Y = R_k D_k, R_k's 16-row blocks as the A operand straight from the row-major packed factor
//                (dealt to the workgroup's wavefronts, each operand shared by up to 4 sub-tiles of 16 samples), the
//                samples' coordinates as the B operand from an LDS stage that is filled once per workgroup and
//                serves all K components;
//   k_logpdf<0>, k_resp<0>   (pmc_persample.hip, engine TILES) read those forms -- every fused output, kind and
//                mode of the compiled units, same code;
//   k_big_stats  sum u | sum u d | sum u d d^T:  one wavefront per (sample chunk, component, block of 48 x 48 or
//                64 x 64 coordinates of the lower triangle), 4 samples per v_mfma_f64_16x16x4_f64, the samples staged
//                in LDS for the workgroup's 8 tasks; partials in k_stats' layout for the same fixed-order finishing
//                kernel.
//   (k_propose_big: pmc_propose.hip)
//
// Measured (profiles/r02_big_dims.txt, 1e6 samples x 32 components): D = 128  log-pdf 13.2 ms = 41 algorithmic
// TFLOP/s, statistics 12.7 ms = 42; D = 256  44.5 ms = 48 and 56.4 ms = 38 -- against 41-51 for the compiled D = 64 unit.
//
// Operand layouts of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md):  A[i][k] lane 16 k + i,  B[k][j] lane
// 16 k + j,  C[i][j] lane 16 (i mod 4) + j, register i / 4.
#include "pmc_device.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));        // 16 bytes at 8-byte alignment: one dwordx4 load

extern __shared__ double big_lds[];

// ---------------------------------------------------------------------------------------------
// k_big_maha
// ---------------------------------------------------------------------------------------------
// One workgroup = NW wavefronts (4; 8 where the LDS admits a single workgroup per CU) = NT sub-tiles of 16 samples
// (NT = 4: one 64-sample tile; 2 / 1 where the LDS holds no more).  The samples' coordinates are staged ONCE in
// LDS, xs[c * P + sample] (P = 16 NT + 1: the staging writes -- a lane per coordinate of one row, coalesced in HBM
// -- land in distinct banks), and serve all K components.  Per component the 16-row blocks of R are dealt to the
// wavefronts (longest first, to the least loaded), so no two wavefronts load the same rows of R, and every A
// operand feeds NT instructions (one per sub-tile).  A lane fetches the elements (r, kk + 4 q + t), t = 0..3 of its
// row -- the 16 columns of a step are assigned to the instructions' k slots as {kk + 4 q + t : q}, the same for A
// and B -- one step ahead of the 4 NT instructions that consume them.  The wavefronts' partial |y|^2 meet in LDS
// (one barrier per component, buffers alternating), summed in wavefront order: bit-reproducible.
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void k_big_maha(const PmcArgsM a)
{
    constexpr int NS = 16 * NT, P = NS + 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.D, D16 = (D + 15) & ~15, G16 = D16 >> 4;
    const int i = lane & 15, q = lane >> 4;
    double *xs = big_lds;                                  // D16 x P
    double *red = big_lds + (size_t)D16 * P;               // 2 x NW x NS
    const long long g0 = (long long)blockIdx.x * NT;       // first sub-tile
    const long long n0 = g0 * 16;

    // stage: thread -> (sample row, coordinate), coordinates fastest (coalesced); zeros beyond D and beyond N
    for (int idx = threadIdx.x; idx < NS * D16; idx += 64 * NW) {
        const int sl = idx / D16, c = idx - sl * D16;
        const long long row = n0 + sl;
        xs[c * P + sl] = (row < a.N && c < D) ? a.x[row * D + c] : 0.0;
    }
    __syncthreads();

    // this wavefront's row blocks: block I costs G16 - I steps; longest first, each to the least loaded wavefront
    // (the same tiny scalar loop in every wavefront; pairs (I, G16 - 1 - I) would leave a wavefront idle for
    // G16 = 5, 6: D = 65 ... 96)
    unsigned long long mine = 0;
    {
        int load[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) load[w] = 0;
        for (int I = 0; I < G16; ++I) {
            int best = 0;
#pragma unroll
            for (int w = 1; w < NW; ++w) best = load[w] < load[best] ? w : best;
#pragma unroll
            for (int w = 0; w < NW; ++w) load[w] += (w == best) ? G16 - I : 0;
            if (best == wave) mine |= 1ull << I;
        }
    }
    for (int k = 0; k < a.K; ++k) {
        const double *pk = a.pack + (size_t)k * a.stride;
        double part[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) part[t] = 0.0;
        {
            for (unsigned long long todo = mine; todo != 0; todo &= todo - 1) {
                const int I = __builtin_ctzll(todo);
                const int r = 16 * I + i;                  // this lane's row of R (A operand: i = lane & 15)
                const bool rv = r < D;
                const int rr = rv ? r : D - 1;
                // element (r, c >= r) of the packed upper triangle sits at pk[roff + c]: offsets from the component's
                // (uniform) base, so a load is base + 32-bit lane offset + immediate
                const unsigned roff = (unsigned)(D + rr * D - rr * (rr - 1) / 2 - rr);
                const unsigned lane_a = roff + 4u * q, lane_m = 4u * q;
                const double *xl = xs + (size_t)(4 * q) * P + i;              // + kk * P: this lane's B operands
                d4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
                // Operands of the NEXT step are fetched raw (unconditional loads) in front of this step's multiplies
                // and turned into A / mu -- zeros below the diagonal and beyond D -- only when their step comes: a
                // select right behind its load makes the compiler wait for the load there, and a conditional fetch
                // is a branch behind which it can no longer count the loads in flight.  Only a block's first step
                // (the diagonal) and a last step that reaches beyond D need clamped addresses and selects; all steps
                // in between -- "interior": every column beyond every row of the block and inside D -- are plain
                // loads at base + kk (the counters had shown 11.5 vector instructions per matrix instruction:
                // 38 % of the issue slots for address clamps and selects).
                double An[4], mn[4];
                auto interior = [&](int kk) { return kk > 16 * I && kk + 16 <= D; };         // uniform
                auto fetch = [&](int kk) {
                    if (interior(kk)) {
                        // a lane's four columns are adjacent: two 16-byte loads instead of four 8-byte ones -- the
                        // CU's one vector-memory pipeline walks each of the instruction's 16 rows (cache lines)
                        // once per instruction, and it, not the matrix pipe, was what the kernel waited for
                        const d2 a0 = *(const d2 *)(pk + lane_a + (unsigned)kk), a1 = *(const d2 *)(pk + lane_a + (unsigned)kk + 2);
                        const d2 m0 = *(const d2 *)(pk + lane_m + (unsigned)kk), m1 = *(const d2 *)(pk + lane_m + (unsigned)kk + 2);
                        An[0] = a0[0], An[1] = a0[1], An[2] = a1[0], An[3] = a1[1];
                        mn[0] = m0[0], mn[1] = m0[1], mn[2] = m1[0], mn[3] = m1[1];
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int c = kk + 4 * q + t;
                            An[t] = pk[roff + (unsigned)(c < rr ? rr : (c < D ? c : D - 1))];
                            mn[t] = pk[c < D ? c : D - 1];
                        }
                    }
                };
                fetch(16 * I);
                for (int kk = 16 * I; kk < D16; kk += 16) {
                    double A[4], mu[4];
                    if (interior(kk)) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            A[t] = An[t];
                            mu[t] = mn[t];
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int c = kk + 4 * q + t;
                            A[t] = (rv && c >= r && c < D) ? An[t] : 0.0;
                            mu[t] = c < D ? mn[t] : 0.0;
                        }
                    }
                    // this step's B operands: all 4 NT LDS reads issued up front (left to itself the compiler reads each
                    // pair right in front of its two multiplies and waits for it there: an LDS round trip per pair)
                    const double *xk = xl + (size_t)kk * P;
                    double Bv[4][NT];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int sub = 0; sub < NT; ++sub) Bv[t][sub] = xk[t * P + 16 * sub];
                    fetch(kk + 16);                        // (beyond the last step: clamped, unused)
                    __builtin_amdgcn_sched_barrier(0);     // ... and issued HERE, in front of this step's multiplies
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int sub = 0; sub < NT; ++sub) {
                            const double B = Bv[t][sub] - mu[t];
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
        double *rk = red + (size_t)(k & 1) * NW * NS;
#pragma unroll
        for (int sub = 0; sub < NT; ++sub) {
            double v = part[sub];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) rk[wave * NS + 16 * sub + i] = v;
        }
        __syncthreads();
        if (wave == k % NW && lane < NS) {
            double tot = rk[lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) tot += rk[w * NS + lane];
            const long long g = g0 + (lane >> 4);
            if (g * 16 < ((a.N + 63) >> 6) * 64)
                a.mtile[((size_t)(g >> 2) * a.K + k) * 64 + (size_t)(g & 3) * 16 + (lane & 15)] = tot;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_big_stats
// ---------------------------------------------------------------------------------------------
// Block -> (chunk, task group) as in k_stats (all groups of a sample chunk on one XCD).  A workgroup = 8
// wavefronts = 8 consecutive tasks (k, p); p = BI (BI + 1) / 2 + BJ is a block of 16 BT x 16 BT coordinates
// (BI, BJ <= BI) of the lower triangle, i.e. BT x BT instruction tiles whose 2 BT operands are read once per 4
// samples: A[i][s] = u_s d_s[16 I + i], B[s][j] = d_s[16 J + j] (a diagonal block skips its upper tiles; it also
// carries the first moments of its coordinates, block (0, 0) the sum of the weights).  The samples go through LDS
// in steps of S rows (row-major, DB doubles each, zeros beyond D and beyond the chunk) that all 8 tasks share; the
// next step's rows are fetched into registers while the matrix pipe works on the current one.
// BT = 4 (64 x 64 coordinates, 16 accumulator tiles = 128 VGPRs) or 3, whichever pads less (stats_bt below): every
// staged byte feeds BT^2 / 4 times the arithmetic of 2 x 2 tiles, whose staging traffic -- K x blocks / 8 passes over
// the samples: 40 at D = 128, K = 32 against 12 -- and short steps made them slower.
#ifndef PMC_BIG_STATS_WAVES
#define PMC_BIG_STATS_WAVES 2
#endif
template <int BT, int NPRE, int NJ>
__global__ __launch_bounds__(512, PMC_BIG_STATS_WAVES) void k_big_stats(const PmcArgsB b, const int S)
{
    constexpr int BC = 16 * BT;                           // coordinates per block
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = b.dreal, DB = (D + BC - 1) / BC * BC;
    const int GB = DB / BC, nb = GB * (GB + 1) / 2;
    const int bid = blockIdx.x;
    const int qd = bid >> 3;
    const int chunk = (bid & 7) + 8 * (qd / b.ngroups);
    const int group = qd % b.ngroups;
    const long long task = (long long)group * 8 + wave;
    const int k = (int)(task / nb), p = (int)(task % nb);
    const bool active = k < b.K;                          // wave-uniform (an idle wavefront still stages and syncs)
    int BI = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while (BI * (BI + 1) / 2 > p) --BI;
    while ((BI + 1) * (BI + 2) / 2 <= p) ++BI;
    const int BJ = p - BI * (BI + 1) / 2;
    const bool diag = BI == BJ;

    const long long t0 = (long long)chunk * b.tiles_per_chunk;
    long long t1 = t0 + b.tiles_per_chunk;
    if (t1 > b.ntiles) t1 = b.ntiles;
    long long n1 = t1 * 64;
    if (n1 > b.N) n1 = b.N;
    const int stride = pmc_pack_stride_c(D), PS = pmc_stats_stride_c(D);
    const int kc = active ? k : 0;
    const double *pk = b.pack + (size_t)kc * stride;
    const int i16 = lane & 15, s = lane >> 4;
    double mi[BT], mj[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        const int ci = BC * BI + 16 * t + i16, cj = BC * BJ + 16 * t + i16;
        mi[t] = ci < D ? pk[ci] : 0.0;
        mj[t] = cj < D ? pk[cj] : 0.0;
    }

    d4 acc[BT][BT];
    double f[BT], su = 0.0;
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        f[t] = 0.0;
#pragma unroll
        for (int v = 0; v < BT; ++v) acc[t][v] = d4{0.0, 0.0, 0.0, 0.0};
    }
    // staging: element e of a thread is idx = threadIdx.x + 512 e of the step's S x DB image; its global value is
    // fetched into a register one step ahead
    // NPRE staging registers per thread and NJ sub-steps per step: S * DB <= 512 NPRE elements, S <= 4 NJ rows.  The 16
    // accumulator tiles of BT = 4 are 128 VGPRs, so it stages less per step (4 x 4 up to D = 512, 8 x 2 beyond)
    const int nimg = S * DB;
    const int step_row = 512 / DB, step_col = 512 % DB;
    double pre[NPRE];
    // (every load below is unconditional, from a clamped address, and the value is selected afterwards: a
    //  conditional load is a branch, and behind branches the compiler can no longer count the loads in flight --
    //  it waited for ALL of them, the prefetch included, in front of the first multiply)
    // Element e of a thread sits at (row soff[e] of the step, column): offset eoff[e] from the step's first sample,
    // fixed for the whole chunk (-1: not part of the image or a padding column -> 0).  The address arithmetic per
    // step is one uniform base; rows beyond the chunk's end (its last step only) fall back to the chunk's last row.
    int eoff[NPRE], esl[NPRE];
    {
        int sl = threadIdx.x / DB, c = threadIdx.x % DB;
#pragma unroll
        for (int e = 0; e < NPRE; ++e) {
            const bool in = (int)threadIdx.x + 512 * e < nimg && c < D;
            eoff[e] = in ? sl * D + c : -1;
            esl[e] = sl;
            sl += step_row;
            c += step_col;
            if (c >= DB) { c -= DB; ++sl; }
        }
    }
    unsigned premask = 0;                                 // bit e: pre[e] is an element of the image (else 0)
    auto gload = [&](long long nstart) {
        const double *xb = b.x + nstart * D;              // uniform
        const long long left = n1 - nstart;               // rows of this step inside the chunk (uniform)
        premask = 0;
        if (left >= S) {
#pragma unroll
            for (int e = 0; e < NPRE; ++e) {
                pre[e] = xb[eoff[e] < 0 ? 0 : eoff[e]];
                premask |= eoff[e] >= 0 ? (1u << e) : 0u;
            }
        } else {
            const double *xl = b.x + (n1 - 1) * D;        // a row that exists
#pragma unroll
            for (int e = 0; e < NPRE; ++e) {
                const bool in = eoff[e] >= 0 && esl[e] < left;
                pre[e] = in ? xb[eoff[e]] : xl[0];
                premask |= in ? (1u << e) : 0u;
            }
        }
    };
    const long long nbeg = t0 * 64;
    gload(nbeg);                                          // (addresses are clamped: harmless for an empty chunk)
    for (long long n0 = nbeg; n0 < n1; n0 += S) {
        // this step's weights: in flight across the two barriers
        const double *ub = b.u + ((size_t)(n0 >> 6) * b.K + kc) * 64;        // uniform
        double uu[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // (a step lies inside one 64-sample tile, whose K x 64 block of u exists in full: no clamp)
            const double v = ub[(int)(n0 & 63) + 4 * j + s];
            uu[j] = (active && 4 * j < S && n0 + 4 * j + s < n1) ? v : 0.0;
        }
        __syncthreads();                                  // the previous step's reads are done
#pragma unroll
        for (int e = 0; e < NPRE; ++e)                    // (the image is 512 NPRE long: every thread writes all its slots)
            big_lds[threadIdx.x + 512 * e] = ((premask >> e) & 1u) ? pre[e] : 0.0;
        __syncthreads();
        gload(n0 + S);                                    // unconditional: a branch here hides the load count from the compiler
        if (!active) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (4 * j >= S) break;                        // uniform
            // (keeps the unrolled sub-steps apart: left alone the scheduler hoists all their LDS reads -- spills)
            __builtin_amdgcn_sched_barrier(0);
            const double u = uu[j];
            const double *xr = big_lds + (size_t)(4 * j + s) * DB + i16;
            double A[BT], B[BT];
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const double a = xr[BC * BI + 16 * t] - mi[t];
                B[t] = diag ? a : xr[BC * BJ + 16 * t] - mj[t];
                A[t] = u * a;
            }
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int v = 0; v < BT; ++v)
                    if (v <= t || !diag)                  // wave-uniform: a diagonal block skips its upper tiles
                        acc[t][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t], B[v], acc[t][v], 0, 0, 0);
            if (diag) {
#pragma unroll
                for (int t = 0; t < BT; ++t) f[t] += A[t];
                su += u;
            }
        }
    }
    if (!active) return;

    double *out = b.partials + ((size_t)chunk * b.K + k) * PS;
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int v = 0; v < BT; ++v) {
            if (diag && v > t) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                 // C: col = lane & 15, row = (lane >> 4) + 4 r
                const int gi = BC * BI + 16 * t + s + 4 * r, gj = BC * BJ + 16 * v + i16;
                if (gi < D && gj <= gi) out[1 + D + (size_t)gi * (gi + 1) / 2 + gj] = acc[t][v][r];
            }
        }
    if (diag) {
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            double m = f[t];
            m += __shfl_xor(m, 16, 64);
            m += __shfl_xor(m, 32, 64);
            const int c = BC * BI + 16 * t + i16;
            if (s == 0 && c < D) out[1 + c] = m;
        }
        if (BI == 0) {
            su += __shfl_xor(su, 16, 64);
            su += __shfl_xor(su, 32, 64);
            if (lane == 0) out[0] = su;
        }
    }
}

}  // namespace

template <int NT, int NW> static hipError_t launch_maha(const PmcArgsM &a, hipStream_t st)
{
    const int D16 = (a.D + 15) & ~15;
    const size_t lds = sizeof(double) * ((size_t)D16 * (16 * NT + 1) + 2 * NW * 16 * NT);
    if (lds > 65536) {
        const hipError_t once = PMC_SET_LDS_PER_DEVICE((&k_big_maha<NT, NW>), 160 * 1024);
        if (once != hipSuccess) return once;
    }
    const long long nsub = ((a.N + 63) >> 6) * 4;
    hipLaunchKernelGGL((k_big_maha<NT, NW>), dim3((unsigned)((nsub + NT - 1) / NT)), dim3(64 * NW), lds, st, a);
    return hipGetLastError();
}

// sub-tiles per workgroup: as many as fit the LDS next to the reduction buffers (156 KB) -- sharing every A operand
// among 4 sub-tiles beats a second workgroup per CU (D = 256: 26 against 23 TFLOP/s with 2 sub-tiles, 78 KB).
// Where only one workgroup fits a CU (beyond 78 KB) it has 8 wavefronts, two per SIMD, to hide the operand fetches.
extern "C" hipError_t pmc_launch_big_maha(const PmcArgsM &a, hipStream_t st)
{
    const size_t D16 = (size_t)((a.D + 15) & ~15);
    auto bytes = [&](int nt) { return 8 * (D16 * (16 * nt + 1) + 2 * 8 * 16 * nt); };
    const int nt = bytes(4) <= 156 * 1024 ? 4 : (bytes(2) <= 156 * 1024 ? 2 : 1);
    const bool wide = bytes(nt) > 78 * 1024;
    if (nt == 4) return wide ? launch_maha<4, 8>(a, st) : launch_maha<4, 4>(a, st);
    if (nt == 2) return wide ? launch_maha<2, 8>(a, st) : launch_maha<2, 4>(a, st);
    return wide ? launch_maha<1, 8>(a, st) : launch_maha<1, 4>(a, st);
}

// instruction tiles per block side of the statistics kernel: 4 (64 coordinates) or 3 (48), whichever pads the
// lower triangle with fewer tiles -- D = 65 ... 96: 21 tiles with 3 against 36 with 4; D = 97 ... 128: 36 with 4
// against 45.  (2 -- 32 coordinates -- wastes least of all and is slower: its steps are too short for their
// barriers and staging, 13.6 against 12.6 ms at D = 72.)
static int stats_tiles(int D, int bt)
{
    const int bc = 16 * bt, gb = (D + bc - 1) / bc;
    return gb * (bt * (bt + 1) / 2) + gb * (gb - 1) / 2 * bt * bt;
}
// (3 x 3 only where it saves a fifth of the tiles: its 9 tiles per 6 operand reads run at a lower rate than 16 per 8 --
//  D = 200: 120 tiles in 56 ms against 136 tiles in 53 ms)
static int stats_bt(int D) { return 5 * stats_tiles(D, 3) < 4 * stats_tiles(D, 4) ? 3 : 4; }

extern "C" hipError_t pmc_launch_big_stats(const PmcArgsB &b, unsigned grid, hipStream_t st)
{
    // rows per LDS step: what the staging registers hold (512 threads x NPRE doubles), at most 4 NJ
    const int bt = stats_bt(b.dreal), bc = 16 * bt;
    const int DB = (b.dreal + bc - 1) / bc * bc;
    const bool wide = DB > 512;
    const int npre = wide ? 8 : 4, smax = wide ? 8 : 16;
    const int cap = 512 * npre / DB;
    const int S = cap >= smax ? smax : (cap >= 16 ? 16 : (cap >= 8 ? 8 : 4));
    // LDS image: one element per (thread, staging register), S * DB of them used
    const size_t lds = sizeof(double) * 512 * npre;
    if (bt == 3 && wide) hipLaunchKernelGGL((k_big_stats<3, 8, 2>), dim3(grid), dim3(512), lds, st, b, S);
    else if (bt == 3) hipLaunchKernelGGL((k_big_stats<3, 4, 4>), dim3(grid), dim3(512), lds, st, b, S);
    else if (wide) hipLaunchKernelGGL((k_big_stats<4, 8, 2>), dim3(grid), dim3(512), lds, st, b, S);
    else hipLaunchKernelGGL((k_big_stats<4, 4, 4>), dim3(grid), dim3(512), lds, st, b, S);
    return hipGetLastError();
}

// geometry knobs the dispatcher's stats_geom() uses: tasks per component, wavefronts per workgroup
extern "C" void pmc_big_stats_config(int D, int *nsub, int *waves)
{
    const int bc = 16 * stats_bt(D);
    const int GB = (D + bc - 1) / bc;
    *nsub = GB * (GB + 1) / 2;
    *waves = 8;
}
