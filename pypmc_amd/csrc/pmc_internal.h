// Internal (non-ABI) declarations shared by the per-dimension kernel units and the dispatcher.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int PMC_TILE = 64;       // samples per tile = one wavefront
constexpr int PMC_NSCALARS = 8;    // per-launch scalar reductions
// (overridable for A/B builds of the whole library.  With the hand-scheduled scalar loads of round 4, ms at the headline
//  for k_logpdf / k_resp_groups: 2 wavefronts 3.69 / 3.39, 4: 3.20 / 2.97, 8: 3.47 / 3.33 -- four share a scalar-cache
//  fill behind one barrier per component; config 3: 9.2 / 6.7 / 9.0)
#ifndef PMC_A_WAVES_N
#define PMC_A_WAVES_N 4
#endif
constexpr int PMC_A_WAVES = PMC_A_WAVES_N;     // wavefronts (tiles) per workgroup in the per-sample kernels
// k_resp parks its first components in LDS between its passes (pmc_resp_klds below) and the rest in the output buffer
// fused small-D E-step (pmc_fused.hip): wavefronts per workgroup, components per wavefront in its
// responsibility phase (so K <= PMC_F_WAVES / 2 * PMC_F_KQMAX = 32), largest compiled dimension
#define PMC_F_WAVES 8
#define PMC_F_KQMAX 8
#ifndef PMC_F_UNROLL_A
#define PMC_F_UNROLL_A 2
#endif
#define PMC_FUSED_MAX_DIM 7
#define PMC_FUSED_MAX_K 32
// ... and its register-resident form for D <= PMC_F_REG_MAX_DIM (k_estep_reg): pmc_freg_kqmax(D) components per
// wavefront -- their per-lane moments, 1 + D + D(D+1)/2 each, are what the registers have to hold -- so
// K <= PMC_F_WAVES * pmc_freg_kqmax(D).  Measured per 4e6 samples (ms; LDS form | 4 | 8 components per wavefront):
//   D = 1:  K = 32  0.356 | 0.346 | 0.288        D = 2:  K = 8  0.129 | 0.130 | 0.143,  K = 16  - | 0.221 | 0.225,
//   K = 32  0.428 | 0.406 | 0.403;     D = 3: the LDS form with its matrix-pipe statistics wins (K = 32: 0.493 | 0.517).
#ifndef PMC_F_REG_MAX_DIM
#define PMC_F_REG_MAX_DIM 2
#endif
__host__ __device__ constexpr int pmc_freg_kqmax(int D)
{
    return D > PMC_F_REG_MAX_DIM ? 0 : (D == 1 ? 8 : 4);
}
// Below this many components the one-kernel form loses to the two kernels from D = 5 on (K = 8, 4e6 samples:
// D = 5  0.277 against 0.258 ms, D = 7  0.308 against 0.285; D = 3  0.156 against 0.209 the other way round)
#define PMC_FUSED_MIN_K_FROM_D5 9

// Mahalanobis engines of the per-sample kernels (pmc_persample.hip) by compiled dimension, and with
// them the layout of the triangular factor in the parameter pack (pmc_pack_components):
//   SGPR  D <  PMC_DPP_FROM                      R row-major (i, j >= i), read through the scalar cache
//   DPP   PMC_DPP_FROM <= D < PMC_MFMA_FROM      unit-diagonal rows in pairs, U_ij = R_ij / R_ii, s_i = R_ii^2,
//                                                streamed through VGPRs (row broadcast)
//   MFMA  D >= PMC_MFMA_FROM, D % 4 == 0         R row-major, staged in LDS
// The DPP engine is a build-time alternative that is OFF by default: it removes every scalar-cache stall
// (91 % instead of 86 % of the issue slots busy at D = 20) -- and the chip answers with a lower clock
// (1.77 instead of 1.93 GHz, same box, profiles/r02_dpp_engine_ab.txt): these kernels are power-bound.
#ifndef PMC_DPP_FROM
#define PMC_DPP_FROM 1000
#endif
#ifndef PMC_MFMA_FROM
#define PMC_MFMA_FROM 32
#endif
//   TILES compiled "dimension" 0 = any sample dimension at run time (the unit behind D > PMC_MAX_DIM): the
//         Mahalanobis forms come from a kernel of their own (k_big_maha, pmc_big.hip: 16 x 16 x 4 fp64 MFMA over
//         row-major R) as tile-major maha_nk, and the per-sample kernels only read them
enum { PMC_ENG_SGPR = 0, PMC_ENG_DPP = 1, PMC_ENG_MFMA = 2, PMC_ENG_TILES = 3 };
__host__ __device__ constexpr int pmc_engine(int D)
{
    return D == 0 ? PMC_ENG_TILES
                  : ((D >= PMC_MFMA_FROM && D % 4 == 0) ? PMC_ENG_MFMA : (D >= PMC_DPP_FROM ? PMC_ENG_DPP : PMC_ENG_SGPR));
}
// largest sample dimension of the run-time-dimension unit (its kernels stage 16 samples x D doubles per wavefront
// in LDS: 128 KB at D = 1024)
#define PMC_BIG_MAX_DIM 1024

// Components k_resp parks in LDS: 19 x 512 B per wavefront is what 16 wavefronts per CU (4 per SIMD, the
// register-limited occupancy) leave room for in 160 KB (19: 3.35-3.38 ms, 16: 3.40-3.44 ms at N = 1e7, K = 32,
// D = 20); 16 where the MFMA engine needs LDS for its parameter buffers too.
// Below D = 12 the registers allow a fifth wavefront per SIMD, which 16 leaves room for (D = 8: 0.56 against 0.62 ms
// per 4e6 samples).
__host__ __device__ constexpr int pmc_resp_klds(int D)
{
    return (pmc_engine(D) == PMC_ENG_MFMA || D < 12) ? 16 : 19;
}

// hipFuncSetAttribute is a per-DEVICE setting.  One process may drive several devices (pmc_init_devices: a host thread per
// device): a function-local `static const hipError_t once = hipFuncSetAttribute(...)` -- the form of rounds 1-4, when a
// process had one device -- would raise a kernel's dynamic LDS limit on the FIRST device only and let the launches on the
// others fail.  Here: one bit per device ordinal and call site, set once the call succeeded on that device.
#include <atomic>
inline hipError_t pmc_lds_attr_dev(std::atomic<unsigned long long> *done, const void *fn, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<unsigned long long> &word = done[(dev >> 6) & 3];
    const unsigned long long bit = 1ull << (dev & 63);
    if (word.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) word.fetch_or(bit, std::memory_order_release);
    return e;
}
#define PMC_SET_LDS_PER_DEVICE(fnptr, bytes)                                                          \
    ([&]() -> hipError_t {                                                                            \
        static std::atomic<unsigned long long> done_[4];                                              \
        return pmc_lds_attr_dev(done_, reinterpret_cast<const void *>(fnptr), (int)(bytes));          \
    }())

__host__ __device__ constexpr int pmc_tri(int D) { return D * (D + 1) / 2; }
__host__ __device__ constexpr int pmc_pack_stride_c(int D) { return (D + pmc_tri(D) + 6 + 7) & ~7; }
__host__ __device__ constexpr int pmc_stats_stride_c(int D) { return 1 + D + pmc_tri(D); }

// per-sample kernels (log-pdf, responsibilities)
struct PmcArgsA {
    const double *x;
    long long N;
    int dreal;            // real sample dimension (== compiled D unless padded)
    const double *pack;
    int K;
    const double *pack2;  // k_logpdf: target mixture of the importance weights (or NULL)
    int K2;
    double *log_target_out;
    int max_init_zero;
    int mode;             // pmc_resp_mode (responsibility kernels)
    int klds;             // responsibility kernels: components parked in LDS between the passes
    long long ld;
    double *out;
    double *individual;
    const double *log_target;
    double *weights;
    const double *sample_w;
    const long long *latent;
    double *atile;        // k_logpdf: tile-major maha_nk of the FIRST mixture, kept for pmc_estep_from_tiles (or NULL)
    const double *mtile;  // run-time-dimension unit: tile-major maha_nk of `pack` (k_big_maha made them), else NULL
    const double *mtile2; //   ... and of `pack2`
    double *u;            // tile-major responsibilities (output)
    int ku;               // emitting passes: columns of u -- the first ku components of `pack` (0: all K).  The components behind
                          // them have no weight (pruned components of a PMC run, sorted to the end by the caller: they take
                          // part in log q's row maximum and leave no responsibilities)
    double *scratch;      // tile-major scratch (Student-t: maha between the two passes)
    double *vpartials;    // Student-t: ntiles * K * 2 per-wavefront sums of v1, v2
    double *r, *log_rho, *exponent;
    double *partials;     // gridDim.x * PMC_NSCALARS
    double *gscale;       // k_resp_groups: ntiles x ceil(K / 16) x 64 per-(sample, group of 16 components) factors
                          // (k_logpdf's emitting epilogue writes ones there when it stands in for k_mgemm)
    // the exact kernels as fall-back behind k_mgemm (pmc_mgemm.hip): a workgroup returns at once unless *redo != 0
    // and blockflag[blockIdx.x] != 0 (both NULL: an ordinary launch)
    const int *blockflag;
    const int *redo;
    // Components of a sample block split over workgroups (k_logpdf_split / k_resp_groups_split, round 6): workgroups
    // [0, split_b1) of the launch are whole blocks of PMC_A_WAVES tiles as ever; from there on every block is walked by
    // split_s1 + split_s2 PIECES -- workgroups that take split_c1 components of `pack` each (pieces 0 .. s1 - 1; the
    // responsibility kernel: split_c1 GROUPS of 16) or split_c2 of `pack2` (pieces s1 .. s1 + s2 - 1) -- and leave their
    // (maximum, sum [, bound term]) per sample in split_part; the piece that draws the block's last ticket combines them
    // in piece order and does what follows the component loop.
    int split_b1, split_s1, split_c1, split_s2, split_c2;
    int split_complete;       // k_resp_groups_split: the finishing workgroup multiplies the factors into u itself (u is complete,
                              // nothing is left to the statistics kernel): small batches, whose statistics run per component
    double *split_part;       // [block - split_b1][piece][2 or 3][256]
    unsigned *split_ticket;   // [block - split_b1], zero between launches (the last ticket wraps it)
};
// pieces of one launch / blocks that are walked in pieces (the library's ticket counters; the caller's workspace holds
// PMC_SPLIT_MAX_PIECES x 3 x 256 doubles at most)
constexpr int PMC_SPLIT_MAX_BLOCKS = 4096;
constexpr int PMC_SPLIT_MAX_PIECES = 32768;

// the Mahalanobis forms as one matrix product + the fused per-sample epilogues (pmc_mgemm.hip)
struct PmcArgsQ {
    PmcArgsA a;           // samples, outputs (out / weights with log_target / partials), sample_w; u + gscale: the grouped
                          // responsibilities (k_resp_groups' form, one group per pass)
    int kind;             // pmc_kind of the components
    int npass;            // passes of 16 NCT components (K padded up to a multiple)
    const double *img;    // [npass * NCT][NSTEPP][64] coefficient image (k_theta_build)
    const double *ctab;   // [npass * NCT * 16][4] c0, c1, c2, c3-or-weight
    const double *center; // [D] the common centre
    const double *guard;  // Theta_1, Theta_2, Theta_3
    double eps_tol;       // tolerance / eps_g: a sample with Theta_1 |d|^2 + Theta_2 |d| + Theta_3 beyond it flags its workgroup
    int *blockflag;       // [gridDim.x] (output)
    int *redo;            // set to 1 if any workgroup was flagged (zeroed by the caller)
};
constexpr int PMC_RESP_GROUP = 16;   // components per group of k_resp_groups = one row block of k_stats_gemm

// responsibilities from kept Mahalanobis forms (pmc_tiles.hip, one unit for all dimensions)
struct PmcArgsT {
    const double *mtile;  // ntiles x ld x 64: maha_nk as k_logpdf kept them, column = position in ITS pack
    long long N;
    int ld;               // components of the mixture the tiles were made with
    int dreal;            // sample dimension (Student-t: gamma = (nu + D) / (nu + maha))
    const double *pack;   // this call's components
    int K;
    int stride;           // doubles per component in the pack
    int coff;             // offset of c0..c3 | weight | column in a component's block
    int max_init_zero;
    const double *sample_w;
    double *u;            // ntiles x K x 64, tile-major (output)
    double *vpartials;    // Student-t: ntiles x K x 2 per-wavefront sums (or NULL)
    double *partials;     // gridDim.x * PMC_NSCALARS (or NULL)
};

// the two sums of the Student-t degree-of-freedom condition from emitted responsibilities (k_dof_sums, pmc_tiles.hip)
struct PmcArgsV {
    const double *u;      // ntiles x K x 64, tile-major: u' of the emitting pass
    const double *gscale; // ntiles x ceil(K / 16) x 64: the factors u' is still to be multiplied with
    const double *weights;// N: the importance weights of the pass
    const double *lse;    // N: log q of the pass
    long long N;
    int K;                // columns of u = live components, the first K of the pack
    int dreal;
    const double *pack;
    int stride, coff;     // doubles per component in the pack, offset of c0..c3 | weight | column
    long long ntiles;
    int tiles_per_chunk;
    double *vpartials;    // gridDim.y x K x 2 (output)
};

// statistics kernel
struct PmcArgsB {
    const double *x;
    long long N;
    int dreal;
    const double *pack;
    int K;
    const double *u;
    double *partials;     // nchunks * K * pmc_stats_stride_c(Dcompiled)
    long long ntiles;
    int nchunks;          // multiple of 8 (XCD count)
    int tiles_per_chunk;
    int ngroups;
    const int *ctl;       // NULL, or the control block of a call that tried the common-shift form first: the kernel
                          // returns at once unless ctl[PMC_CTL_REDO] != 0
};

// statistics kernel, component x monomial form (k_stats_gemm): moments about ONE common shift `center`
struct PmcArgsG {
    const double *x;
    long long N;
    int dreal;
    const double *pack;   // the components: their triangular factors serve the a-priori test (kind >= 0)
    const double *spack;  // the pack whose means are the components' own shifts (= pack unless the caller gave others):
                          // they give the common shift c (midrange per coordinate) and the re-centring
    int kind;             // pmc_kind of `pack`, or -1: no a-priori test
    double limit_prior;
    double *center;       // dreal doubles (device, output of workgroup 0): the common shift c
    int K;
    const double *u;      // tile-major ntiles x K x 64
    const double *gscale; // NULL, or per-(sample, 16 components) factors u is still to be multiplied with (k_resp_groups)
    double *partials;     // [nchunks * slices][K][msp]: monomials 1 | d | d d^T lower triangle (row-major i, j <= i), d = x - c
    long long ntiles;
    int nchunks;          // multiple of 8 (XCD count)
    int tiles_per_chunk;
    int ngroups;          // groups of 32 components
    int ncs;              // column super groups (workgroups that split the monomials)
    int *ctl;             // control block (device, output of workgroup 0): {go, redo}
};
// control block of a statistics call that may use the common-shift form (device memory, in the workspace)
enum { PMC_CTL_GO = 0, PMC_CTL_REDO = 1, PMC_CTL_INTS = 4 };

// fused E-step kernel
struct PmcArgsF {
    const double *x;
    long long N;
    int dreal;
    const double *pack;
    int K;
    int max_init_zero;
    int qs;               // wavefronts that share a tile in the responsibility phase (1, 2, 4)
    int kq;               // components per wavefront in the responsibility phase: ceil(K / qs)
    int cw;               // wavefronts that own different components in the statistics phase (1, 2, 4, 8)
    const double *sample_w;
    double *partials;     // gridDim.x * (PMC_F_WAVES / cw) * K * pmc_stats_stride_c(Dcompiled)
    double *spartials;    // gridDim.x * PMC_NSCALARS
    double *vpartials;    // Student-t: gridDim.x * (PMC_F_WAVES / qs) * K * 2 per-(workgroup, tile slot) sums of v1, v2
    const double *shift_pack;   // NULL, or a pack whose means are the points the moments are taken about (else: pack's)
    long long ntiles;
    int rounds_per_wg;
    int reg;              // 1: register-resident form (k_estep_reg): qs wavefronts x kq components per tile
};

// run-time-dimension unit (pmc_big.hip): Mahalanobis forms of N samples x K components, tile-major
struct PmcArgsM {
    const double *x;
    long long N;
    int D;                // sample dimension = dimension the pack was made for
    const double *pack;
    int K;
    int stride;           // doubles per component in the pack
    double *mtile;        // ntiles x K x 64 (output)
};

// propose kernel
struct PmcArgsP {
    const double *mu;             // K x dreal
    const double *chol;           // K x dreal x dreal, lower triangular
    const double *dof;            // K or NULL
    const long long *offsets;     // K + 1 exclusive prefix sums of the component counts
    int K, dreal;
    long long N, first_sample;
    unsigned long long seed;
    double *x;
    long long *origin;
};

struct PmcKernelSet {
    int dim;              // compiled dimension (run-time-dimension unit: the dimension itself)
    int padded;           // 1: accepts dreal <= dim;  2: the run-time-dimension unit (D > PMC_MAX_DIM)
    int stats_nsub;       // row subsets per component in the statistics kernel
    int stats_waves;      // wavefronts per workgroup in the statistics kernel
    hipError_t (*logpdf)(int kind, int kind2, const PmcArgsA &, unsigned grid, hipStream_t);
    hipError_t (*resp)(int kind, const PmcArgsA &, unsigned grid, hipStream_t);
    hipError_t (*resp_groups)(int kind, const PmcArgsA &, unsigned grid, hipStream_t);   // NULL: none (run-time-dimension unit)
    hipError_t (*stats)(const PmcArgsB &, unsigned grid, hipStream_t);
    void (*config)(int *nsub, int *waves);
    hipError_t (*propose)(const PmcArgsP &, unsigned grid, hipStream_t);
    hipError_t (*fused)(int kind, int qs, const PmcArgsF &, unsigned grid, hipStream_t);
    int (*fused_lds_bytes)(int qs, int K);
    // component x monomial statistics (NULL / cols_per_wg == 0: this dimension has none)
    hipError_t (*stats_gemm)(const PmcArgsG &, unsigned grid, hipStream_t);
    void (*gemm_config)(int *cols_per_wg, int *slices, int *msp, int *wgs_per_cu);
    int gemm_cols, gemm_slices, gemm_msp;   // monomial tiles (of 16) per workgroup, sample slices, partial row length
    int gemm_wgs;                           // workgroups of it that share a CU
    // Mahalanobis forms as one matrix product (NULL / mg_nstepp == 0: this dimension has none)
    hipError_t (*mgemm)(int nct, const PmcArgsQ &, unsigned grid, hipStream_t);
    hipError_t (*theta)(const double *pack, int K, int Kpad, int kind, double *img, double *ctab, double *center,
                        unsigned long long *guard, hipStream_t);
    void (*mgemm_config)(int *nstepp, int *nct_max);
    int mg_nstepp, mg_nct_max;
    // the per-sample kernels with the components of a block split over workgroups (NULL: the run-time-dimension unit)
    hipError_t (*logpdf_split)(int kind, int kind2, const PmcArgsA &, unsigned grid, hipStream_t);
    hipError_t (*resp_groups_split)(int kind, const PmcArgsA &, unsigned grid, hipStream_t);
    hipError_t (*logpdf2)(const PmcArgsA &, unsigned grid, hipStream_t);      // A/B builds only (-DPMC_TWO_PER_LANE)
};
