// pmc_api.hip -- C ABI (include/pmc_hip.h) of libpmc_hip.so: argument checking, host-side
// parameter packing, dispatch to the per-dimension kernel units, and the small fixed-order
// finishing kernels.  Never throws; every failure becomes a status code + pmc_last_error().
#include "../../include/pmc_hip.h"
#include "pmc_dims.h"
#include "pmc_internal.h"
#include "pmc_convert.h"

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

// ---------------------------------------------------------------------------------------------
// kernel-set registry (launchers of the per-dimension units pmc_persample / pmc_stats / pmc_propose)
// ---------------------------------------------------------------------------------------------
#define PMC_DECL_UNIT(d, p) \
    extern "C" hipError_t pmc_launch_logpdf_d##d##_p##p(int, int, const PmcArgsA &, unsigned, hipStream_t); \
    extern "C" hipError_t pmc_launch_resp_d##d##_p##p(int, const PmcArgsA &, unsigned, hipStream_t);   \
    extern "C" hipError_t pmc_launch_resp_groups_d##d##_p##p(int, const PmcArgsA &, unsigned, hipStream_t); \
    extern "C" hipError_t pmc_launch_logpdf_split_d##d##_p##p(int, int, const PmcArgsA &, unsigned, hipStream_t); \
    extern "C" hipError_t pmc_launch_logpdf2_d##d##_p##p(const PmcArgsA &, unsigned, hipStream_t); \
    extern "C" hipError_t pmc_launch_resp_groups_split_d##d##_p##p(int, const PmcArgsA &, unsigned, hipStream_t); \
    extern "C" hipError_t pmc_launch_stats_d##d##_p##p(const PmcArgsB &, unsigned, hipStream_t);       \
    extern "C" void pmc_stats_config_d##d##_p##p(int *, int *);                                        \
    extern "C" hipError_t pmc_launch_propose_d##d##_p##p(const PmcArgsP &, unsigned, hipStream_t);         \
    extern "C" hipError_t pmc_launch_fused_d##d##_p##p(int, int, const PmcArgsF &, unsigned, hipStream_t); \
    extern "C" int pmc_fused_lds_bytes_d##d##_p##p(int, int);                                          \
    extern "C" hipError_t pmc_launch_stats_gemm_d##d##_p##p(const PmcArgsG &, unsigned, hipStream_t);   \
    extern "C" void pmc_stats_gemm_config_d##d##_p##p(int *, int *, int *, int *);          \
    extern "C" hipError_t pmc_launch_mgemm_d##d##_p##p(int, const PmcArgsQ &, unsigned, hipStream_t);   \
    extern "C" hipError_t pmc_launch_theta_d##d##_p##p(const double *, int, int, int, double *, double *, double *, \
                                                       unsigned long long *, hipStream_t);               \
    extern "C" void pmc_mgemm_config_d##d##_p##p(int *, int *);
extern "C" hipError_t pmc_launch_resp_tiles(int, const PmcArgsT &, unsigned, hipStream_t);
extern "C" hipError_t pmc_launch_dof_sums(const PmcArgsV &, unsigned, unsigned, hipStream_t);
// the run-time-dimension unit (pmc_big.hip, pmc_persample.hip / pmc_propose.hip compiled with PMC_D = 0)
extern "C" hipError_t pmc_launch_logpdf_d0_p0(int, int, const PmcArgsA &, unsigned, hipStream_t);
extern "C" hipError_t pmc_launch_resp_d0_p0(int, const PmcArgsA &, unsigned, hipStream_t);
extern "C" hipError_t pmc_launch_big_maha(const PmcArgsM &, hipStream_t);
extern "C" hipError_t pmc_launch_big_stats(const PmcArgsB &, unsigned, hipStream_t);
extern "C" void pmc_big_stats_config(int, int *, int *);
extern "C" hipError_t pmc_launch_propose_big(const PmcArgsP &, unsigned, hipStream_t);
#define PMC_DECL_X(d) PMC_DECL_UNIT(d, 0)
#define PMC_DECL_XP(d) PMC_DECL_UNIT(d, 0) PMC_DECL_UNIT(d, 1)
PMC_DIM_LIST(PMC_DECL_X, PMC_DECL_XP)

namespace {

#define PMC_SET(d, p) \
    {d, p, 0, 0, &pmc_launch_logpdf_d##d##_p##p, &pmc_launch_resp_d##d##_p##p, &pmc_launch_resp_groups_d##d##_p##p, \
     &pmc_launch_stats_d##d##_p##p, &pmc_stats_config_d##d##_p##p, &pmc_launch_propose_d##d##_p##p, \
     &pmc_launch_fused_d##d##_p##p, &pmc_fused_lds_bytes_d##d##_p##p, &pmc_launch_stats_gemm_d##d##_p##p, \
     &pmc_stats_gemm_config_d##d##_p##p, 0, 0, 0, 0, &pmc_launch_mgemm_d##d##_p0, &pmc_launch_theta_d##d##_p0, \
     &pmc_mgemm_config_d##d##_p0, 0, 0, &pmc_launch_logpdf_split_d##d##_p##p, &pmc_launch_resp_groups_split_d##d##_p##p, \
     &pmc_launch_logpdf2_d##d##_p##p}
struct DimEntry {
    int dim;
    bool has_padded;
    PmcKernelSet exact, padded;
};
#define PMC_ENT_X(d) {d, false, PMC_SET(d, 0), PMC_SET(d, 0)},
#define PMC_ENT_XP(d) {d, true, PMC_SET(d, 0), PMC_SET(d, 1)},
DimEntry g_dims[] = {PMC_DIM_LIST(PMC_ENT_X, PMC_ENT_XP)};
constexpr int g_ndims = sizeof(g_dims) / sizeof(g_dims[0]);

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    // A failed runtime call (a hipMalloc beyond the device's memory, ...) leaves its code behind as the thread's "last
    // error", and the hipGetLastError behind the NEXT kernel launch would report it as that launch's: the error has been
    // handed to the caller here, so it is taken off the runtime's slate (every failure of the library, the handle layer
    // and the exchange passes through this function).
    if (code == PMC_EHIP) (void)hipGetLastError();
    return code;
}

int hipfail(hipError_t e, const char *what)
{
    return fail(PMC_EHIP, "%s: %s", what, hipGetErrorString(e));
}

// ---------------------------------------------------------------------------------------------
// kernel timing through the ABI (pmc_timing_enable / pmc_get_timings): every launch of a hot kernel is
// bracketed by HIP events on the caller's stream while timing is on
// ---------------------------------------------------------------------------------------------
enum { T_LOGPDF = 0, T_RESP, T_STATS, T_FUSED, T_PROPOSE, T_FINISH, T_COUNT };
const char *const g_timing_names[T_COUNT] = {"k_logpdf", "k_resp", "k_stats", "k_estep_fused", "k_propose",
                                             "finishing reductions"};
struct TimingRec {
    hipStream_t st;       // the stream the launch went to (the handle layer reads its own context's records)
    int dev;              // ... and its device: an event belongs to the device it was created on
    int id;
    int calls;            // 1, or 0 for a bracket that continues the previous launch of the same kernel
    hipEvent_t a, b;
    double flops, bytes;
};
std::mutex g_timing_mutex;
bool g_timing_on = false;
std::vector<TimingRec> g_timing_recs;
std::vector<hipStream_t> g_timing_streams;                  // streams whose launches are timed for their context only
bool timing_stream_on(hipStream_t st)                      // (g_timing_mutex held)
{
    for (hipStream_t q : g_timing_streams)
        if (q == st) return true;
    return false;
}
// events wait for their next use per DEVICE (an event can only be recorded on a stream of the device it was created on,
// and one process may drive several devices: pmc_init_devices)
constexpr int PMC_TIMING_MAX_DEVICES = 64;
std::vector<hipEvent_t> g_timing_pool[PMC_TIMING_MAX_DEVICES];
constexpr size_t PMC_TIMING_MAX_RECORDS = 1 << 16;

hipEvent_t timing_event(int dev)                            // (g_timing_mutex held; the calling thread's device is `dev`)
{
    std::vector<hipEvent_t> &pool = g_timing_pool[dev];
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// RAII bracket: records the start event now and the stop event when it goes out of scope
struct Timed {
    TimingRec rec;
    hipStream_t st;
    bool on;
    Timed(int id, hipStream_t st_, double flops, double bytes, int calls = 1) : st(st_), on(false)
    {
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        if (!(g_timing_on || timing_stream_on(st)) || g_timing_recs.size() >= PMC_TIMING_MAX_RECORDS) return;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PMC_TIMING_MAX_DEVICES) return;
        rec.st = st;
        rec.dev = dev;
        rec.id = id; rec.calls = calls; rec.flops = flops; rec.bytes = bytes;
        rec.a = timing_event(dev);
        rec.b = timing_event(dev);
        if (!rec.a || !rec.b) return;
        on = hipEventRecord(rec.a, st) == hipSuccess;
    }
    ~Timed()
    {
        if (!on) return;
        const bool ok = hipEventRecord(rec.b, st) == hipSuccess;
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        if (ok) {
            g_timing_recs.push_back(rec);
        } else {                                           // (no stop event, no record: the pair goes back to its pool)
            (void)hipGetLastError();
            g_timing_pool[rec.dev].push_back(rec.a);
            g_timing_pool[rec.dev].push_back(rec.b);
        }
    }
};

// algorithmic work per launch
double flops_pairs(double N, int K, int D) { return N * K * ((double)D * D + 4.0 * D + 40.0); }
double flops_stats(double N, int K, int D) { return N * K * (1.0 + 2.0 * D + (double)D * (D + 1)); }

// kernel set (and with it the compiled dimension) for a sample dimension D
// kernel sets of the run-time-dimension unit, one per dimension asked for (they differ in `dim` and the
// statistics geometry only); entries are never removed, so the pointers handed out stay valid
const PmcKernelSet *big_kernels_for(int D)
{
    static std::mutex m;
    static std::vector<PmcKernelSet *> sets;
    std::lock_guard<std::mutex> lock(m);
    for (const PmcKernelSet *ks : sets)
        if (ks->dim == D) return ks;
    PmcKernelSet *ks = new PmcKernelSet();
    ks->dim = D;
    ks->padded = 2;
    pmc_big_stats_config(D, &ks->stats_nsub, &ks->stats_waves);
    ks->logpdf = &pmc_launch_logpdf_d0_p0;
    ks->resp = &pmc_launch_resp_d0_p0;
    ks->resp_groups = nullptr;
    ks->stats = &pmc_launch_big_stats;
    ks->config = nullptr;
    ks->propose = &pmc_launch_propose_big;
    ks->fused = nullptr;
    ks->fused_lds_bytes = nullptr;
    ks->stats_gemm = nullptr;
    ks->gemm_config = nullptr;
    ks->gemm_cols = ks->gemm_slices = ks->gemm_msp = ks->gemm_wgs = 0;
    ks->mgemm = nullptr;
    ks->theta = nullptr;
    ks->mgemm_config = nullptr;
    ks->mg_nstepp = ks->mg_nct_max = 0;
    ks->logpdf_split = nullptr;
    ks->resp_groups_split = nullptr;
    ks->logpdf2 = nullptr;
    sets.push_back(ks);
    return ks;
}

const PmcKernelSet *kernels_for(int D)
{
    if (D > PMC_MAX_DIM) return D <= PMC_BIG_MAX_DIM ? big_kernels_for(D) : nullptr;
    for (int i = 0; i < g_ndims; ++i) {
        PmcKernelSet *ks = nullptr;
        if (g_dims[i].dim == D) ks = &g_dims[i].exact;
        else if (g_dims[i].dim > D) ks = g_dims[i].has_padded ? &g_dims[i].padded : nullptr;
        else continue;
        if (ks && ks->stats_nsub == 0) {                                                // idempotent
            ks->gemm_config(&ks->gemm_cols, &ks->gemm_slices, &ks->gemm_msp, &ks->gemm_wgs);
            ks->mgemm_config(&ks->mg_nstepp, &ks->mg_nct_max);
            ks->config(&ks->stats_nsub, &ks->stats_waves);
        }
        return ks;
    }
    return nullptr;
}

// ---------------------------------------------------------------------------------------------
// finishing kernels (single workgroup, fixed summation order)
// ---------------------------------------------------------------------------------------------
// scalars[i] = sum_b partials[b*PMC_NSCALARS + i] in a fixed order, in one launch:
// workgroup g sums its contiguous slice of the rows (thread t: scalar t % 8, rows t / 8, t / 8 + 32, ... --
// every wavefront load covers 8 whole 64-byte rows; fixed LDS tree) into slices[g][8]; the workgroup that
// takes the last ticket adds the slices in ascending g.  The ticket counter wraps to 0 by itself
// (atomicInc), cross-workgroup visibility follows MI355X_MICROARCH.md: release fence + drained vmcnt before
// the ticket, acquire fence after it.
constexpr int FIN_GROUPS = 64;
__global__ __launch_bounds__(256) void k_finish_scalars(const double *__restrict__ partials,
                                                        long long nblocks, double *slices,
                                                        unsigned *counter, double *__restrict__ scalars)
{
    __shared__ double red[256];
    __shared__ unsigned ticket;
    const int i = threadIdx.x & (PMC_NSCALARS - 1), r = threadIdx.x >> 3;
    const long long per = (nblocks + FIN_GROUPS - 1) / FIN_GROUPS;
    const long long b0 = (long long)blockIdx.x * per;
    long long b1 = b0 + per;
    if (b1 > nblocks) b1 = nblocks;
    double v = 0.0;
    for (long long b = b0 + r; b < b1; b += 256 / PMC_NSCALARS) v += partials[b * PMC_NSCALARS + i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s >= PMC_NSCALARS; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < PMC_NSCALARS) __hip_atomic_store(&slices[blockIdx.x * PMC_NSCALARS + threadIdx.x], red[threadIdx.x],
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticket = atomicInc(counter, FIN_GROUPS - 1);
    }
    __syncthreads();
    if (ticket != FIN_GROUPS - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // the 64 x 8 slices: fetched by all threads at once (two each) into LDS, then summed per scalar in the same ascending
    // order as before (same bits) -- one thread per scalar walking 64 L2 round trips was most of this kernel's 10 us
    __shared__ double sl[FIN_GROUPS * PMC_NSCALARS];
    for (int q = threadIdx.x; q < FIN_GROUPS * PMC_NSCALARS; q += 256)
        sl[q] = __hip_atomic_load(&slices[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x < PMC_NSCALARS) {
        double t = 0.0;
        for (int g = 0; g < FIN_GROUPS; ++g) t += sl[g * PMC_NSCALARS + threadIdx.x];
        scalars[threadIdx.x] = t;
    }
}

// stats[k][p] (real dimension D) = sum_chunk partials[chunk][k][p'] (compiled dimension Dc).
// One wavefront per output element: lane l sums the chunks l, l + 64, ... in ascending order, then a
// fixed shuffle tree -- the same order on every launch (bit-reproducible), and the nchunks loads of an
// element are 64 independent streams instead of one dependent chain.
__global__ __launch_bounds__(256) void k_finish_stats(const double *__restrict__ partials,
                                                      int nchunks, int K, int D, int Dc,
                                                      double *__restrict__ stats, const int *__restrict__ ctl)
{
    if (ctl && ctl[PMC_CTL_REDO] == 0) return;            // the common-shift form stands
    const int PS = pmc_stats_stride_c(D), PSc = pmc_stats_stride_c(Dc);
    const long long idx = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= (long long)K * PS) return;
    const int k = (int)(idx / PS), p = (int)(idx % PS);
    int pc;
    if (p < 1 + D) pc = p;                                // sum u, first moments
    else pc = p - (1 + D) + (1 + Dc);                     // lower triangle: i(i+1)/2+j is D-free
    double v = 0.0;
    for (int c = lane; c < nchunks; c += 64) v += partials[((size_t)c * K + k) * PSc + pc];
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) stats[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// the common-shift (component x monomial) form of the statistics: plan, reduce, re-centre + check
// ---------------------------------------------------------------------------------------------
// Library options (pmc_configure): the form is tried for K >= g_gemm_min_k when the compiled dimension has the
// kernel; a component whose weighted mean lies further than sqrt(limit) of its own standard deviations (in some
// coordinate) from the common shift sends the call back to the per-component-shift kernel.
// The options live in one struct: the process-wide instance pmc_configure() writes, and -- for the handle layer,
// include/pmc_ctx.h -- one instance per context (pmc_ctx_configure), installed for the calling thread while a call of
// that context runs.  Every public entry point takes ONE snapshot when it starts (TuneScope) and everything below it
// reads that snapshot: a pmc_configure() from another thread cannot change the selection between the two halves of a
// call (advice r3).
struct PmcTuning {
    int gemm_min_k = 17;
    double gemm_limit = 1000.0;
    long long gemm_min_n = 524288;
    double gemm_min_fill = 0.63;
    int resp_groups = 1;
    size_t big_scratch_bytes = 256u << 20;
    double mgemm_tol = 5e-11;
    long long mgemm_min_n = 49152;
    // components of a sample block split over workgroups (k_logpdf_split / k_resp_groups_split; split_plan below)
    int split = 1;                 // 0: never
    int split_min_comps = 0;       // components per piece at least (0: by Mahalanobis engine)
    int split_tail_pieces = 4;     // pieces per block of a launch that fills the chip (its last round only)
    double split_max_rounds = 24;  // launches of more rounds than this are not split at all
    // a small launch is split until it has this many workgroups per slot of the chip -- but into at most split_max_pieces
    // pieces per mixture: every piece pays the load of its samples and a share of the merge, and from ~16 pieces on that is
    // more than the shorter walk saves (scripts/split_small_sweep.py, profiles/r06_split_small_sweep.txt: D = 40, K = 128,
    // N = 4096: 64 pieces 66.7 us, 32: 43.3, 16: 37.3; D = 20, K = 128: 32 pieces 31.4, 16: 23.1)
    double split_fill = 1.0;
    int split_max_pieces = 16;
    // rounds of the chip, in front of the last (partial) one, that are walked in pieces too.  Measured (scripts/split_tail_sweep.py,
    // profiles/r06_split_tail_sweep.txt): 0 -- the remainder alone -- is best or within 1 % of the best at every shape but
    // config 2's (K = 16: 0.5 by 2.6 %)
    double split_tail_rounds = 0.25;
    int small_grouped = 1;            // pmc_estep of a small batch: grouped responsibilities in pieces, completed in place
    int split_tail_min_comps = 0;     // components per piece at least in such a launch (0: by Mahalanobis engine)
};
PmcTuning g_tuning;
std::mutex g_tuning_mutex;
thread_local const PmcTuning *t_tuning = nullptr;
PmcTuning tun()
{
    if (t_tuning) return *t_tuning;
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    return g_tuning;
}
struct TuneScope {
    PmcTuning snap;
    bool mine;
    TuneScope() : mine(t_tuning == nullptr)
    {
        if (mine) {
            snap = tun();
            t_tuning = &snap;
        }
    }
    ~TuneScope()
    {
        if (mine) t_tuning = nullptr;
    }
};
#define g_gemm_min_k (tun().gemm_min_k)
#define g_gemm_limit (tun().gemm_limit)
// ... and from g_gemm_min_n samples per 32 components on: the form costs three launches more (reduce, re-centre, the
// skipped fallback pair) and its own prologue, ~35 us that pay back at 6.7e-8 ms per sample and 32 components
// (scripts/gemm_crossover.py, D = 20: N = 262144, K = 32: 0.133 against 0.110 ms; N = 4e6: 1.09 against 1.32)
#define g_gemm_min_n (tun().gemm_min_n)
// ... and only if the groups of 32 components are filled well enough: a group costs the same whether it holds 1 or 32
// (one row block is no cheaper than two: the B operands dominate then), so K = 33 ... 40 runs 4-11 % slower than the
// per-component kernel at D = 20 and K = 41 36 % faster (scripts/gemm_crossover.py ksweep, profiles/r03_gemm_crossover.txt)
#define g_gemm_min_fill (tun().gemm_min_fill)
// pmc_estep's responsibilities in groups of 16 with their factors left to k_stats_gemm (k_resp_groups): 0 never,
// 1 where it pays, 2 wherever the common-shift statistics run.  Where it pays (scripts/resp_groups_ab.py matrix,
// profiles/r03_resp_groups.txt; responsibilities + statistics, per 2e6 samples): compiled D <= 16 at any K (-2 ... -19 %),
// D = 20, 32, 40 from K = 64 on (-2 ... -6 %: the parked traffic of k_resp costs clock there); round 3 saw it lose at D = 24,
// 30, 48, 64 (+2 ... +14 %: 25-40 more registers than k_resp) and neutral at D = 20, K = 32 (round 5: see below).
#define g_resp_groups (tun().resp_groups)
bool resp_groups_pays(int dim, int K)
{
    if (g_resp_groups != 1) return g_resp_groups == 2;
    // measured per (D, K) with scripts/resp_groups_ab.py matrix (profiles/r03_resp_groups.txt; re-measured in round 5,
    // profiles/r05_resp_groups_matrix.txt: since the hand-scheduled scalar loads of round 4 the grouped form also wins at
    // D = 24 from K = 32 on (-2 / -5 / -9 % of the pair at K = 32 / 64 / 128) and at D = 30 from K = 64 on (-3 / -6.5 %);
    // it still loses at D = 64 (+8 %); D = 32 ... 64 run k_mgemm at these batch sizes either way)
    return dim <= 16 || dim == 20 || dim == 24 || (dim == 30 && K >= 64) || dim == 32 || dim == 40;
}


// totals[k][m] = sum over the nce partial vectors, in a fixed order: thread = monomial (coalesced rows of 64),
// wavefront w of the workgroup takes the partial vectors w, w + 4, ... in ascending order, the four are added in
// wavefront order.
__global__ __launch_bounds__(256) void k_gemm_reduce(const double *__restrict__ partials, int nce, int K, int msp,
                                                     double *__restrict__ totals, const int *__restrict__ ctl)
{
    if (ctl[PMC_CTL_GO] == 0) return;
    __shared__ double red[4][64];
    const int k = blockIdx.y, m = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    double v = 0.0;
    if (m < msp) {
        // eight loads in flight, added in the same ascending order as one by one (same bits): the launch is a chain of
        // dependent L2 round trips otherwise (20.6 us whatever N, a third of the statistics' finishing time at an
        // 8-GPU shard size)
        const double *p = partials + (size_t)k * msp + m;
        const size_t step = (size_t)K * msp;
        int ce = w;
        for (; ce + 28 < nce; ce += 32) {
            double a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = p[(size_t)(ce + 4 * q) * step];
#pragma unroll
            for (int q = 0; q < 8; ++q) v += a[q];
        }
        for (; ce < nce; ce += 4) v += p[(size_t)ce * step];
    }
    red[w][threadIdx.x & 63] = v;
    __syncthreads();
    if (w == 0 && m < msp) totals[(size_t)k * msp + m] = ((red[0][m & 63] + red[1][m & 63]) + red[2][m & 63]) + red[3][m & 63];
}

// One workgroup per component: re-centre the moments from the common shift c to the component's own shift mu_k
// (what pmc_sufficient_stats returns), and test a posteriori whether the common shift was near enough:
//   dlt = mu_k - c;   S0' = S0;   M1'_i = M1_i - S0 dlt_i;   M2'_ij = M2_ij - M1_i dlt_j - dlt_i M1_j + S0 dlt_i dlt_j
//   far  <=>  (M1_i / S0)^2 > limit * var_i,  var_i = M2_ii / S0 - (M1_i / S0)^2,  for a component that holds more
//             than a millionth of the total weight (the rule of mix_adapt._stats.shift_is_far)
__global__ __launch_bounds__(256) void k_gemm_convert(const double *__restrict__ totals, int K, int D, int msp,
                                                      const double *__restrict__ pack, int stride,
                                                      const double *__restrict__ center, double limit,
                                                      double *__restrict__ stats, int *__restrict__ ctl)
{
    if (ctl[PMC_CTL_GO] == 0) return;
    const int k = blockIdx.x, PS = pmc_stats_stride_c(D);
    const double *tk = totals + (size_t)k * msp;
    const double *mu = pack + (size_t)k * stride;
    double *out = stats + (size_t)k * PS;
    const double S0 = tk[0];
    __shared__ double wsum;
    __shared__ double s0s[1024];
    // the K weights: loaded side by side, summed by one thread in ascending order (same bits as a serial walk over global
    // memory, without its K dependent round trips)
    for (int q = threadIdx.x; q < K && q < 1024; q += 256) s0s[q] = totals[(size_t)q * msp];
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < K; ++q) {
            const double s = q < 1024 ? s0s[q] : totals[(size_t)q * msp];
            if (s == s && fabs(s) <= 1.7976931348623157e308) t += s;
        }
        wsum = t;
    }
    __syncthreads();
    const bool counts = S0 > 1e-200 && S0 > 1e-6 * wsum;
    bool far = false;
    for (int p = threadIdx.x; p < PS; p += 256) {
        double v;
        if (p == 0) v = S0;
        else if (p <= D) {
            const int i = p - 1;
            v = tk[p] - S0 * (mu[i] - center[i]);
        } else {
            const int t = p - 1 - D;
            int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while ((i + 1) * (i + 2) / 2 <= t) ++i;
            while (i * (i + 1) / 2 > t) --i;
            const int j = t - i * (i + 1) / 2;
            const double di = mu[i] - center[i], dj = mu[j] - center[j];
            const double m1i = tk[1 + i], m1j = tk[1 + j];
            v = ((tk[p] - m1i * dj) - di * m1j) + S0 * di * dj;
            if (i == j && counts) {
                const double dbar = m1i / S0, raw = tk[p] / S0;
                double var = raw - dbar * dbar;
                if (!(var > 1e-14 * raw)) var = 1e-14 * raw;
                if (dbar * dbar > limit * var) far = true;
            }
        }
        out[p] = v;
    }
    if (far) ctl[PMC_CTL_REDO] = 1;                       // (benign race: every writer stores 1)
}

// The common-shift form was refused and u still lacks the factors k_resp_groups left for k_stats_gemm to apply:
// u_nk *= f_n,group(k), in place, so that the per-component-shift kernel finds what k_resp would have written.
__global__ __launch_bounds__(256) void k_apply_scale(double *__restrict__ u, const double *__restrict__ gscale,
                                                     long long ntiles, int K, const int *__restrict__ ctl)
{
    if (ctl && ctl[PMC_CTL_REDO] == 0) return;            // (a small grid: the launch is there in every call)
    const int G = (K + PMC_RESP_GROUP - 1) / PMC_RESP_GROUP;
    const long long total = ntiles * K * 64;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long tk = idx >> 6;                     // tile * K + k
        const long long tile = tk / K;
        const int k = (int)(tk - tile * K);
        u[idx] *= gscale[(tile * G + k / PMC_RESP_GROUP) * 64 + (idx & 63)];
    }
}

// ... and the factors themselves become ones (a launch of its own: every factor serves 16 components' threads above), so
// that a caller-owned pair (u, factors) -- pmc_importance_weights_emit_grouped -- still means the same u afterwards
__global__ __launch_bounds__(256) void k_reset_scale(double *__restrict__ gscale, long long len, const int *__restrict__ ctl)
{
    if (ctl && ctl[PMC_CTL_REDO] == 0) return;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < len; idx += (long long)gridDim.x * 256) gscale[idx] = 1.0;
}

// N = 1, D = 1: stats[k] = (u_k, u_k d, u_k d^2), d = x - mu_k   (u tile-major: one tile, lane 0)
__global__ __launch_bounds__(256) void k_stats_single(const double *__restrict__ x,
                                                      const double *__restrict__ pack, int stride, int K,
                                                      const double *__restrict__ u,
                                                      double *__restrict__ stats)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const double uk = u[(size_t)k * 64], d = x[0] - pack[(size_t)k * stride];
    const double ud = uk * d;
    stats[3 * k] = uk;
    stats[3 * k + 1] = ud;
    stats[3 * k + 2] = fma(ud, d, 0.0);
}

// vsums[k*2+c] = sum_tile vpartials[(tile*K + k)*2 + c]   (one workgroup per output, fixed order)
__global__ __launch_bounds__(256) void k_finish_vsums(const double *__restrict__ vpartials,
                                                      long long ntiles, int K,
                                                      double *__restrict__ vsums)
{
    __shared__ double red[256];
    const int o = blockIdx.x;                             // k*2 + c
    double v = 0.0;
    for (long long t = threadIdx.x; t < ntiles; t += 256) v += vpartials[(size_t)t * K * 2 + o];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) vsums[o] = red[0];
}

// importance-weight sums over an existing weight vector (convergence.py:31-39, :67-72)
__global__ __launch_bounds__(256) void k_weight_sums(const double *__restrict__ w, long long N,
                                                     double *__restrict__ partials)
{
    __shared__ double red[4][PMC_NSCALARS];
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    double sc[3] = {0.0, 0.0, 0.0};
    if (n < N) {
        const double v = w[n];
        sc[0] = v;
        sc[1] = (v != 0.0) ? v * log(v) : 0.0;
        sc[2] = v * v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = 0; i < 3; ++i) {
        double v = sc[i];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < PMC_NSCALARS) {
        double v = 0.0;
        if (threadIdx.x < 3)
            for (int q = 0; q < 4; ++q) v += red[q][threadIdx.x];
        partials[(size_t)blockIdx.x * PMC_NSCALARS + threadIdx.x] = v;
    }
}

// logsumexp2D (pypmc/tools/_regularize.pyx:57-84) of an existing row-major N x K matrix:
// out[n] = max_k a[n,k] + log sum_k w_k exp(a[n,k] - max), max initialised to -DBL_MAX.
__global__ __launch_bounds__(256) void k_logsumexp2d(const double *__restrict__ a,
                                                     const double *__restrict__ w, long long N,
                                                     int K, double *__restrict__ out)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double *row = a + n * K;
    double mx = -1.7976931348623157e308;
    for (int k = 0; k < K; ++k)
        if (row[k] > mx) mx = row[k];
    double res = 0.0;
    for (int k = 0; k < K; ++k) res += w[k] * exp(row[k] - mx);
    out[n] = log(res) + mx;
}

// Deterministic-mixture weights of one run (pypmc/sampler/importance_sampling.py:313-365); the
// operation order of both branches follows the reference statement by statement.
__global__ __launch_bounds__(256) void k_combine_weights(const double *__restrict__ q, long long N, int T,
                                                         const double *__restrict__ counts, int t,
                                                         const double *__restrict__ omega, double n_total,
                                                         int log_scale, double *__restrict__ out,
                                                         double *__restrict__ flag)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double *col = q + n;                           // q[l, n] = col[l * N]
    double w;
    if (log_scale) {
        double lw = log(omega[n]);                       // :349-351
        lw += col[t * N];
        lw += log(n_total);
        double mx = -1.7976931348623157e308;             // logsumexp2D(q, N), _regularize.pyx:57-84
        for (int l = 0; l < T; ++l)
            if (col[l * N] > mx) mx = col[l * N];
        double res = 0.0;
        for (int l = 0; l < T; ++l) res += counts[l] * exp(col[l * N] - mx);
        lw -= log(res) + mx;
        w = exp(lw);                                     // :365
    } else {
        double den = 0.0;                                // :320-323
        for (int l = 0; l < T; ++l) den += counts[l] * exp(col[l * N]);
        den /= n_total;
        w = exp(col[t * N]) * omega[n] / den;            // :325-327
    }
    out[n] = w;
    if (flag && !(fabs(w) <= 1.7976931348623157e308)) atomicAdd(flag, 1.0);
}

// ---------------------------------------------------------------------------------------------
// K-sized work on the device (round 5): the parameter pack, the shift pack and the conversion of the statistics into
// the reference's conventions -- what the host did between the M-step and the E-step's first launch and behind its last
// (verdict r4 #3: 15 % of an E-step at one GPU's share of eight).  Same operations in the same order as the host
// functions they stand in for (pmc_pack_components, pmc_pack_means, pmc_host_convert_stats; -ffp-contract=off here as
// there, sqrt and division correctly rounded on both sides): the SAME BITS, which the tests hold them to.
// ---------------------------------------------------------------------------------------------
// One wavefront per component, lane j = column j of the upper factor (D <= 64); the matrix and its factor in LDS (the
// matrix is read from global memory ONCE, coalesced: a load per row of the factorisation would put a memory latency in
// front of each of its D dependent steps).  status = [1 + failing pivot, or 0 (K) | its value (K)]: nothing to initialise,
// every component writes its two slots.
// blocks K ... 2K-1 (when means != NULL): the pack of the shifts of a statistics pass (pmc_pack_means) in the same launch.
// vbx.on: the constants of a VB posterior's pack (enum pmc_kind, PMC_KIND_VB) formed here from the state's fields and the
// caller's psi parts -- k_vb_expect's work (pmc_vbstate.hip) without its launch: c0 = D / beta, c1 = nu, c2 = E[ln pi],
// c3 = E[ln|Lambda|] - D ln 2 pi with E[ln|Lambda|] = (sum psi + D ln 2) + ln|W|; the two expectations are stored as well.
struct PmcVbExpect {
    int on;
    const double *beta, *nu, *log_det_W, *parts;
    double *ln_lambda, *ln_pi;
    double d_ln_2pi;
};

__global__ __launch_bounds__(64) void k_pack_build(const double *mu, const double *prec, const double *c0, const double *c1,
                                                  const double *c2, const double *c3, const double *weight, const int *column,
                                                  int K, int D, int Dp, int stride, double *pack, double *status,
                                                  const double *means, double *mpack, PmcVbExpect vbx)
{
    extern __shared__ double lds[];
    const int j = threadIdx.x;
    if ((int)blockIdx.x >= K) {
        const int k = blockIdx.x - K;
        double *pk = mpack + (size_t)k * stride;
        for (int idx = j; idx < stride; idx += 64) pk[idx] = 0.0;
        __syncthreads();
        if (j < D) pk[j] = means[(size_t)k * D + j];
        if (j == 0) {
            double *c = pk + Dp + Dp * (Dp + 1) / 2;
            c[4] = 1.0;
            ((long long *)c)[5] = (long long)k;
        }
        return;
    }
    const int k = blockIdx.x;
    double *Rl = lds, *Al = lds + D * D;
    double *pk = pack + (size_t)k * stride;
    const double *A = prec + (size_t)k * D * D;
    for (int idx = j; idx < D * D; idx += 64) Al[idx] = A[idx];
    for (int idx = j; idx < stride; idx += 64) pk[idx] = 0.0;
    __syncthreads();
    if (j < D) pk[j] = mu[(size_t)k * D + j];
    int bad_i = -1;
    double bad_s = 0.0;
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
        if (j >= i && j < D) {
            // (the plain loop's subtractions in its order; the LDS reads issued eight at a time -- a loop that waits for two
            // reads per term is 128 cycles per term, and this kernel is latency from end to end: 22 -> 12 us at K = 64, D = 20)
            s = Al[i * D + j];
            int l = 0;
            for (; l + 8 <= i; l += 8) {
                double x[8], y[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x[q] = Rl[(l + q) * D + i];
                    y[q] = Rl[(l + q) * D + j];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) s -= x[q] * y[q];
            }
            for (; l < i; ++l) s -= Rl[l * D + i] * Rl[l * D + j];
        }
        const double sii = __shfl(s, i, 64);                          // the pivot, to every lane
        if (!(sii > 0.0) || !isfinite(sii)) {
            bad_i = i;
            bad_s = sii;
            break;                                                    // (wave-uniform)
        }
        const double rii = sqrt(sii);
        if (j == i) Rl[i * D + i] = rii;
        else if (j > i && j < D) Rl[i * D + j] = s / rii;
        __syncthreads();
    }
    if (j == 0) {
        status[k] = (double)(bad_i + 1);                              // 0 = factorised, else 1 + the failing pivot
        status[K + k] = bad_s;
    }
    if (bad_i >= 0) return;
    // packed row-major (i, j >= i) for the compiled dimension; padding rows / columns stay zero
    for (int i = 0; i < D; ++i)
        if (j >= i && j < D) pk[Dp + i * Dp - i * (i - 1) / 2 + (j - i)] = Rl[i * D + j];
    if (j == 0) {
        double *c = pk + Dp + Dp * (Dp + 1) / 2;
        if (vbx.on) {
            const double lam = vbx.parts[K + k] + vbx.log_det_W[k];
            vbx.ln_lambda[k] = lam;
            vbx.ln_pi[k] = vbx.parts[k];
            c[0] = D / vbx.beta[k];
            c[1] = vbx.nu[k];
            c[2] = vbx.parts[k];
            c[3] = lam - vbx.d_ln_2pi;
        } else {
            c[0] = c0 ? c0[k] : 0.0;
            c[1] = c1 ? c1[k] : 0.0;
            c[2] = c2 ? c2[k] : 0.0;
            c[3] = c3 ? c3[k] : 0.0;
        }
        c[4] = weight ? weight[k] : 1.0;
        ((long long *)c)[5] = column ? (long long)column[k] : (long long)k;
    }
}

// pmc_host_convert_stats on the device (pmc_convert.h: shared with the VB state's fused conversion).
__global__ __launch_bounds__(256) void k_convert_stats(const double *stats, const double *shift, const double *ncov, int K, int D,
                                                      const double *scalars, double *out)
{
    pmc_convert_stats_block(stats, shift, ncov, K, D, scalars, out);
}

inline long long ceil_div(long long a, long long b) { return (a + b - 1) / b; }

// statistics launch geometry
struct StatsGeom {
    int ngroups, nchunks, tiles_per_chunk;
    long long ntiles;
    unsigned grid;
};
StatsGeom stats_geom(long long N, int K, const PmcKernelSet *ks)
{
    StatsGeom g;
    g.ntiles = ceil_div(N, PMC_TILE);
    g.ngroups = (int)ceil_div((long long)K * ks->stats_nsub, ks->stats_waves);
    // aim at ~8 resident wavefronts per SIMD-set: 256 CUs * 16 wavefronts
    long long want = ceil_div(256LL * 16, (long long)g.ngroups * ks->stats_waves);
    if (want > g.ntiles) want = g.ntiles;
    if (want < 1) want = 1;
    g.nchunks = (int)(ceil_div(want, 8) * 8);              // multiple of the XCD count
    if (ks->padded == 2) {
        // run-time-dimension unit: ONE 8-wavefront workgroup per CU (its registers) and long-running workgroups, so
        // the number of workgroups should fill whole rounds of 256: the chunk count (multiple of 8, partials within
        // 256 MB) with the fewest idle CUs in the last round, the larger one among equals
        const double per_chunk = 8.0 * K * pmc_stats_stride_c(ks->dim);
        long long cmax = (long long)(256.0 * 1024 * 1024 / per_chunk) / 8 * 8;
        if (cmax > 128) cmax = 128;
        if (cmax > ceil_div(g.ntiles, 8) * 8) cmax = ceil_div(g.ntiles, 8) * 8;
        if (cmax < 8) cmax = 8;
        double best = -1.0;
        for (long long c = 8; c <= cmax; c += 8) {
            const long long wgs = c * g.ngroups, rounds = ceil_div(wgs, 256);
            const double eff = (double)wgs / (256.0 * rounds);
            if (eff >= best - 1e-9) { best = eff > best ? eff : best; g.nchunks = (int)c; }
        }
    }
    g.tiles_per_chunk = (int)ceil_div(g.ntiles, g.nchunks);
    if (g.tiles_per_chunk < 1) g.tiles_per_chunk = 1;
    g.grid = (unsigned)((long long)g.nchunks * g.ngroups);
    return g;
}

// geometry of the component x monomial form (k_stats_gemm): workgroup = (chunk, group of 32 components, column
// super group); one workgroup per CU is resident (LDS), so about one round of 256
struct GemmGeom {
    int ngroups, ncs, nchunks, tiles_per_chunk, nce;
    long long ntiles;
    unsigned grid;
};
bool gemm_available(const PmcKernelSet *ks) { return ks->stats_gemm != nullptr && ks->gemm_cols > 0; }
GemmGeom gemm_geom(long long N, int K, const PmcKernelSet *ks)
{
    GemmGeom g;
    g.ntiles = ceil_div(N, PMC_TILE);
    g.ngroups = (int)ceil_div(K, 32);
    g.ncs = (int)ceil_div(ks->gemm_msp / 16, ks->gemm_cols);
    const long long nsub = (long long)g.ngroups * g.ncs;
    long long c = 256LL * (ks->gemm_wgs > 0 ? ks->gemm_wgs : 1) / nsub / 8 * 8;
    if (c < 8) c = 8;
    const long long cmax = ceil_div(ceil_div(g.ntiles, 4), 8) * 8;          // at least ~4 tiles per chunk
    if (c > cmax) c = cmax;
    g.nchunks = (int)c;
    g.tiles_per_chunk = (int)ceil_div(g.ntiles, g.nchunks);
    if (g.tiles_per_chunk < 1) g.tiles_per_chunk = 1;
    g.nce = g.nchunks * ks->gemm_slices;
    g.grid = (unsigned)(g.nchunks * nsub);
    return g;
}
// does pmc_estep take the common-shift statistics for this call?  (pmc_configure's knobs)
bool gemm_selected(const PmcKernelSet *ks, long long N, int K, int kind)
{
    return kind >= 0 && gemm_available(ks) && K >= g_gemm_min_k && g_gemm_limit > 0.0 && N >= 16384 &&
           N * ceil_div(K, 32) >= g_gemm_min_n && (double)K >= g_gemm_min_fill * 32.0 * (double)ceil_div(K, 32);
}
// workspace of a statistics call: [region shared by the two forms' partial sums | centre (Dc doubles) | control block |
// factors of k_resp_groups: ceil(N / 64) x ceil(K / 16) x 64 doubles]
size_t stats_region_bytes(long long N, int K, const PmcKernelSet *ks)
{
    const StatsGeom g = stats_geom(N > 0 ? N : 1, K, ks);
    size_t bytes = (size_t)g.nchunks * K * pmc_stats_stride_c(ks->dim) * sizeof(double);
    if (gemm_available(ks)) {
        const GemmGeom gg = gemm_geom(N > 0 ? N : 1, K, ks);
        const size_t gb = ((size_t)gg.nce + 1) * K * ks->gemm_msp * sizeof(double);      // partial vectors + totals
        if (gb > bytes) bytes = gb;
    }
    return (bytes + 255) & ~(size_t)255;
}
size_t stats_tail_bytes(const PmcKernelSet *ks)
{
    return gemm_available(ks) ? (((size_t)ks->dim * sizeof(double) + 255) & ~(size_t)255) + 256 : 0;
}
size_t gscale_bytes(long long N, int K, const PmcKernelSet *ks)
{
    if (!gemm_available(ks) || !ks->resp_groups) return 0;
    return (size_t)ceil_div(N > 0 ? N : 1, PMC_TILE) * (size_t)ceil_div(K, PMC_RESP_GROUP) * 64 * sizeof(double);
}

// fused E-step launch geometry (pmc_fused.hip)
struct FusedGeom {
    int qs, kq, cw, tpr, rounds_per_wg, reg;
    long long ntiles, nrounds;
    unsigned grid;
    long long nchunks;        // partial statistics vectors: grid * (PMC_F_WAVES / cw), register form: grid * tpr
};
bool fused_reg(int dim, int K) { return pmc_freg_kqmax(dim) > 0 && K <= PMC_F_WAVES * pmc_freg_kqmax(dim); }
bool fused_eligible(const PmcKernelSet *ks, int K, int kind, int mode)
{
    if (ks->dim > PMC_FUSED_MAX_DIM || (K > PMC_FUSED_MAX_K && !fused_reg(ks->dim, K))) return false;
    if (ks->dim >= 5 && K < PMC_FUSED_MIN_K_FROM_D5) return false;
    if (mode == PMC_RESP_VB) return kind == PMC_KIND_VB;
    // (Student-t: the LDS form only, compiled dimensions 3 ... 7)
    return mode == PMC_RESP_PMC_RB && (kind == PMC_KIND_GAUSS || (kind == PMC_KIND_STUDENT_T && !fused_reg(ks->dim, K) &&
                                                                  pmc_freg_kqmax(ks->dim) == 0));
}
FusedGeom fused_geom(long long N, int K, int dim)
{
    FusedGeom g;
    g.reg = fused_reg(dim, K) ? 1 : 0;
    long long wgs_per_cu = 3;
    if (g.reg) {
        // wavefronts per tile x components per wavefront: the split with the fewest idle component slots,
        // the fewest wavefronts per tile among those (their soft-max exchange costs two barriers a round)
        const int kqmax = pmc_freg_kqmax(dim);
        int best = -1;
        for (int qs = 1; qs <= PMC_F_WAVES; qs *= 2) {
            const int kq = (K + qs - 1) / qs;
            if (kq > kqmax) continue;
            if (best < 0 || qs * kq - K < best) {
                best = qs * kq - K;
                g.qs = qs;
                g.kq = kq;
            }
        }
        g.cw = 1;
        wgs_per_cu = 2;
    } else {
        g.qs = K <= PMC_F_KQMAX ? 1 : (K <= 2 * PMC_F_KQMAX ? 2 : 4);
        g.kq = (K + g.qs - 1) / g.qs;
        g.cw = PMC_F_WAVES;
        if (g.qs == 1) {
            g.cw = 1;
            while (g.cw < K) g.cw *= 2;
        }
    }
    g.tpr = PMC_F_WAVES / g.qs;
    g.ntiles = ceil_div(N, PMC_TILE);
    g.nrounds = ceil_div(g.ntiles, g.tpr);
    long long wgs = 256LL * wgs_per_cu;                   // resident workgroups: each is persistent over its rounds
    if (wgs > g.nrounds) wgs = g.nrounds;
    if (wgs < 1) wgs = 1;
    g.rounds_per_wg = (int)ceil_div(g.nrounds > 0 ? g.nrounds : 1, wgs);
    g.grid = (unsigned)ceil_div(g.nrounds > 0 ? g.nrounds : 1, g.rounds_per_wg);
    g.nchunks = g.reg ? (long long)g.grid * g.tpr : (long long)g.grid * (PMC_F_WAVES / g.cw);
    return g;
}

// Run-time-dimension unit: maha_nk of `pack` for all samples into `mtile` (tile-major, pmc_maha_tiles_size doubles)
hipError_t big_maha(const double *d_x, long long N, int D, const double *d_pack, int K, double *mtile, hipStream_t st)
{
    PmcArgsM m;
    std::memset(&m, 0, sizeof(m));
    m.x = d_x; m.N = N; m.D = D; m.pack = d_pack; m.K = K; m.stride = pmc_pack_stride_c(D); m.mtile = mtile;
    return pmc_launch_big_maha(m, st);
}
// stream-ordered scratch of the library's own (the Mahalanobis forms of the run-time-dimension unit when the caller
// does not keep them): allocated and freed on the caller's stream
struct StreamScratch {
    void *p = nullptr;
    hipStream_t st;
    explicit StreamScratch(hipStream_t s) : st(s) {}
    hipError_t get(size_t bytes) { return hipMallocAsync(&p, bytes, st); }
    ~StreamScratch() { if (p) (void)hipFreeAsync(p, st); }
};

// Run-time-dimension unit (D > PMC_MAX_DIM): the per-sample kernel reads the Mahalanobis forms k_big_maha made.  Forms
// the caller does not keep live in a stream-ordered scratch of BOUNDED size (advice r2: it was 8 N (K + K_target)
// bytes, 10 GB at N = 1e7, K = 128, outside pmc_workspace_bytes and outside the caller's allocator): the samples go in
// chunks of a multiple of 256 (one workgroup of the per-sample kernels, so block indices and tile buffers line up),
// each chunk = k_big_maha [x 2] + k_logpdf<0> on pointers moved to the chunk.
#define g_big_scratch_bytes (tun().big_scratch_bytes)     // pmc_configure("big_dim_scratch_bytes")
hipError_t big_logpdf(const PmcKernelSet *ks, int kind, int kind2, const PmcArgsA &full, int D, hipStream_t st)
{
    const bool keep = full.atile != nullptr;
    const int K = full.K, K2 = full.pack2 ? full.K2 : 0;
    const size_t per_sample = sizeof(double) * ((keep ? 0 : K) + K2);
    long long chunk = full.N;
    if (per_sample > 0) {
        chunk = (long long)(g_big_scratch_bytes / per_sample) / 256 * 256;
        if (chunk < 256) chunk = 256;
    }
    StreamScratch scratch(st);
    if (per_sample > 0) {
        const long long cn = chunk < full.N ? chunk : full.N;
        hipError_t em = scratch.get(sizeof(double) * ((keep ? 0 : (size_t)pmc_maha_tiles_size(cn, K)) +
                                                      (K2 ? (size_t)pmc_maha_tiles_size(cn, K2) : 0)));
        if (em != hipSuccess) return em;
    }
    for (long long n0 = 0; n0 < full.N; n0 += chunk) {
        const long long n = (full.N - n0 < chunk) ? full.N - n0 : chunk;
        PmcArgsA a = full;
        a.x = full.x + n0 * D;
        a.N = n;
        if (full.out) a.out = full.out + n0;
        if (full.log_target_out) a.log_target_out = full.log_target_out + n0;
        if (full.individual) a.individual = full.individual + n0 * full.ld;
        if (full.log_target) a.log_target = full.log_target + n0;
        if (full.weights) a.weights = full.weights + n0;
        if (full.sample_w) a.sample_w = full.sample_w + n0;
        if (full.partials) a.partials = full.partials + (n0 / 256) * PMC_NSCALARS;
        double *mt2 = (double *)scratch.p;
        double *mt = keep ? full.atile + (n0 / 64) * K * 64 : mt2 + (K2 ? (size_t)pmc_maha_tiles_size(n, K2) : 0);
        if (keep) a.atile = mt;
        hipError_t em = big_maha(a.x, n, D, full.pack, K, mt, st);
        if (em == hipSuccess && K2) em = big_maha(a.x, n, D, full.pack2, K2, mt2, st);
        if (em != hipSuccess) return em;
        a.mtile = mt;
        a.mtile2 = K2 ? mt2 : nullptr;
        em = ks->logpdf(kind, kind2, a, (unsigned)ceil_div(ceil_div(n, PMC_TILE), PMC_A_WAVES), st);
        if (em != hipSuccess) return em;
    }
    return hipSuccess;
}

size_t scalar_partials_bytes(long long N)
{
    const long long blocks = ceil_div(ceil_div(N, PMC_TILE), PMC_A_WAVES);
    return (size_t)(blocks > 0 ? blocks : 1) * PMC_NSCALARS * sizeof(double);
}
// k_resp_groups writes its factors while its scalar partials (front of the workspace, growing with N) are still live:
// the factors start behind whichever is longer, the statistics' regions or those partials
size_t gscale_offset(long long N, int K, const PmcKernelSet *ks)
{
    const size_t a = stats_region_bytes(N, K, ks) + stats_tail_bytes(ks);
    const size_t b = (scalar_partials_bytes(N > 0 ? N : 1) + 255) & ~(size_t)255;
    return a > b ? a : b;
}

// ---------------------------------------------------------------------------------------------
// the Mahalanobis forms as one matrix product (pmc_mgemm.hip): selection, workspace region, launches
// ---------------------------------------------------------------------------------------------
// Tolerance of the guard in units of a_nk (0 switches the form off): a sample whose priced rounding error
// eps_g (Theta_1 |d|^2 + Theta_2 |d| + Theta_3) exceeds it sends its workgroup to the exact kernel (eps_g: mgemm_eps below,
// ~1e-15); the tolerance leaves a factor 2 to the contract's 1e-10 on responsibilities, i.e. on differences of a_nk.
#define g_mgemm_tol (tun().mgemm_tol)                     // default 5e-11
// The guard's error constant, per compiled dimension (advice r4): the expanded form is a dot product of n = (D + 1)(D + 2) / 2
// monomial terms accumulated in fp64 -- a hard bound would be n u sum|theta z| (u = 1.1e-16: 9.5e-14 at D = 40), which refuses
// healthy data; what is used is the PROBABILISTIC sqrt(n) u growth of a sum of rounding errors with a safety factor:
// eps_g = 0.32 sqrt(n) u = 3.5e-17 sqrt(n): 8.3e-16 at D = 32, 1.03e-15 at D = 40 (the constant of round 4), 1.22e-15 at
// D = 48.  The largest (difference to the exact kernel) / (Theta-sum) seen over all tests and the fuzz is 5.1e-16 at D = 40
// (scripts/mgemm_check.py); tests/test_gpu_mgemm.py holds every case -- ill-conditioned covariances and dimensions below the
// padded one included -- to 0.75 eps_g.  It is not a proof: a caller who needs the exact kernels' bits sets
// "maha_gemm_tolerance" to 0.
inline double mgemm_eps(int Dc) { return 3.5e-17 * std::sqrt(0.5 * (Dc + 1.0) * (Dc + 2.0)); }
// Samples from which the form is tried at all: 256, one workgroup (32768 until round 5).  Small batches are latency-bound in both
// forms -- one wavefront walks ALL components, 2.7 us each in the exact engine at D = 40 -- and the matrix kernel's walk is the
// shorter one at every size, k_theta_build included: D = 40, K = 128: 347 -> 248 us for 256 ... 65536 samples, D = 64, K = 128:
// 1100 -> 563 us (profiles/r05_mgemm_crossover.txt).
#define g_mgemm_min_n (tun().mgemm_min_n)
// component tiles per pass (0: the exact kernels).  K is padded to a multiple of 16 NCT and a padded component costs
// what a real one does; two tiles per pass cost 4.5 % more per pair than four (profiles/r03_maha_gemm_prototype.txt),
// and the form as a whole is ~20 % ahead of the exact kernels, so more padding than that is not worth it.
// Compiled dimensions 20 and 24 (round 5): 61 / 85 steps per pass instead of 216 at D = 40, so a pass's prologue and epilogue
// weigh three times as much and the vector kernels are closer -- the form pays for the NON-emitting passes (log-pdf,
// importance weights) of mixtures with four full tiles per pass only: D = 24: -10 % at K = 64, -16 % at K = 128; D = 20: -2 % at
// K = 64 (not taken), -11 % at K = 128; with two tiles per pass it loses (D = 20, K = 32: +14 %), and its emitting epilogue
// loses against k_resp_groups at every K (profiles/r05_mgemm_small.txt).
int mgemm_pick(const PmcKernelSet *ks, long long N, int K, bool emit)
{
    if (!ks->mgemm || ks->mg_nstepp <= 0 || !(g_mgemm_tol > 0.0) || N < g_mgemm_min_n || K < 24) return 0;
    const bool small_dim = ks->dim < 32;
    if (small_dim && (emit || K < (ks->dim <= 20 ? 96 : 48))) return 0;
    int best = 0;
    double bestc = 1.2 * K;
    for (int nct = ks->mg_nct_max; nct >= (small_dim ? 4 : 2); nct /= 2) {
        const double c = (double)(ceil_div(K, 16 * nct) * 16 * nct) * (nct >= 4 ? 1.0 : 1.045);
        if (c <= bestc) { bestc = c; best = nct; }
    }
    return best;
}
struct MgRegion {
    size_t head, center, ctab, img, flags, lt, lse, bytes;  // byte offsets from the region's start
};
MgRegion mgemm_region(long long N, int K, const PmcKernelSet *ks)
{
    MgRegion r;
    const size_t kpad = (size_t)ceil_div(K, 64) * 64;       // (whatever NCT is picked)
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    r.head = 0;                                             // 3 guard norms (as doubles' bit patterns) | redo flag
    r.center = 256;
    r.ctab = r.center + up(sizeof(double) * 64);
    r.img = r.ctab + up(sizeof(double) * kpad * 4);
    r.flags = r.img + up(sizeof(double) * (kpad / 16) * (size_t)ks->mg_nstepp * 64);
    const size_t nblocks = (size_t)ceil_div(ceil_div(N > 0 ? N : 1, PMC_TILE), PMC_A_WAVES);
    r.lt = r.flags + up(sizeof(int) * nblocks);
    r.lse = r.lt + up(sizeof(double) * (size_t)(N > 0 ? N : 1));     // log q of a Student-t emitting pass (k_dof_sums reads it)
    r.bytes = r.lse + up(sizeof(double) * (size_t)(N > 0 ? N : 1));
    return r;
}
size_t mgemm_bytes(long long N, int K, const PmcKernelSet *ks)
{
    return (ks->mgemm && ks->mg_nstepp > 0) ? mgemm_region(N, K, ks).bytes : 0;
}
size_t mgemm_offset(long long N, int K, const PmcKernelSet *ks)
{
    return (gscale_offset(N, K, ks) + gscale_bytes(N, K, ks) + 255) & ~(size_t)255;
}
// k_theta_build + k_mgemm on `a` (a.blockflag / a.redo are set here for the exact kernel the caller launches behind)
// allow_dead: components with weight 0 are handled by the matrix kernel (the passes that emit no u); otherwise such a
// mixture is refused as a whole and the exact kernel behind does it
hipError_t mgemm_run(const PmcKernelSet *ks, int nct, int kind, const PmcArgsA &a, PmcArgsA &fallback, void *d_workspace,
                     hipStream_t st, bool allow_dead, bool emitting = false)
{
    const MgRegion r = mgemm_region(a.N, a.K, ks);
    char *base = (char *)d_workspace + mgemm_offset(a.N, a.K, ks);
    hipError_t e = hipMemsetAsync(base + r.head, 0, 64, st);
    if (e != hipSuccess) return e;
    const int kpad = (int)(ceil_div(a.K, 16 * nct) * 16 * nct);
    PmcArgsQ q;
    std::memset(&q, 0, sizeof(q));
    q.kind = kind;
    q.npass = kpad / (16 * nct);
    q.img = (const double *)(base + r.img);
    q.ctab = (const double *)(base + r.ctab);
    q.center = (const double *)(base + r.center);
    q.guard = (const double *)(base + r.head);
    q.eps_tol = g_mgemm_tol / mgemm_eps(ks->dim);
    q.blockflag = (int *)(base + r.flags);
    q.redo = (int *)(base + r.head + 32);
    e = ks->theta(a.pack, a.K, kpad, kind | (allow_dead ? 0x100 : 0) | (emitting ? 0x200 : 0), (double *)(base + r.img), (double *)(base + r.ctab), (double *)(base + r.center),
                  (unsigned long long *)(base + r.head), st);
    if (e != hipSuccess) return e;
    q.a = a;
    e = ks->mgemm(nct, q, (unsigned)ceil_div(ceil_div(a.N, PMC_TILE), PMC_A_WAVES), st);
    fallback.blockflag = q.blockflag;
    fallback.redo = q.redo;
    return e;
}
// diagnostics of the last such call in this workspace (synchronises the stream): the guard's three norms and the number
// of workgroups it refused
int mgemm_report(const PmcKernelSet *ks, const void *d_workspace, long long N, int K, hipStream_t st, double *norms,
                 long long *refused, long long *nblocks_out)
{
    const MgRegion r = mgemm_region(N, K, ks);
    const char *base = (const char *)d_workspace + mgemm_offset(N, K, ks);
    const long long nblocks = ceil_div(ceil_div(N, PMC_TILE), PMC_A_WAVES);
    std::vector<int> flags((size_t)nblocks);
    hipError_t e = hipMemcpyAsync(norms, base + r.head, 3 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(flags.data(), base + r.flags, sizeof(int) * (size_t)nblocks, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hipfail(e, "pmc_maha_gemm_report");
    long long c = 0;
    for (int f : flags) c += f != 0;
    *refused = c;
    *nblocks_out = nblocks;
    return PMC_OK;
}

// ---------------------------------------------------------------------------------------------
// components of a sample block split over workgroups (k_logpdf_split / k_resp_groups_split, pmc_persample.hip): the plan
// ---------------------------------------------------------------------------------------------
// A workgroup of the per-sample kernels walks all components of its 256 samples: ~1 us per component at D = 20, 2.7 us at
// D = 40.  A launch of fewer workgroups than the chip has slots therefore costs K such steps whatever N is (the
// reference's own batches -- examples/pmc.py:61-65 draws 1e3 samples per step -- live there), and a launch of a few
// rounds ends with up to one such walk during which the chip runs empty (the 8-way shards of BASELINE's configurations:
// 5-10 % of their kernels' time).  The plan: the LAST blocks of the launch are walked in pieces (pmc_internal.h,
// PmcArgsA::split_*) -- all of them when the launch does not fill the chip.
#define g_split (tun().split)
#define g_split_min_comps (tun().split_min_comps)
#define g_split_tail_pieces (tun().split_tail_pieces)
#define g_split_max_rounds (tun().split_max_rounds)
#define g_split_fill (tun().split_fill)
#define g_split_tail_rounds (tun().split_tail_rounds)
#define g_split_tail_min_comps (tun().split_tail_min_comps)
int device_cus()
{
    static std::mutex m;
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(m);
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount
                                                                                                        : 256;
    }
    return cus[dev];
}
// workgroups of the per-sample kernels (4 wavefronts) a compute unit holds at a time: the registers of the compiled dimension
// decide (pmc_min_waves, pmc_persample.hip; measured occupancies: DESIGN section 6)
int split_slots_per_cu(int dim) { return dim <= 16 ? 5 : (dim <= 20 ? 4 : (dim <= 32 ? 3 : 2)); }
// smallest piece in components: a piece pays the load of its samples, its launch slot and the merge once
int split_min_units(const PmcKernelSet *ks)
{
    if (g_split_min_comps > 0) return g_split_min_comps;
    return 4;
}
// ... and in a launch that fills the chip, where the pieces only have to shorten the last round
int split_tail_min_units(const PmcKernelSet *ks)
{
    if (g_split_tail_min_comps > 0) return g_split_tail_min_comps;
    return ks->dim >= PMC_MFMA_FROM && ks->dim % 4 == 0 ? 4 : 8;
}
struct SplitPlan {
    bool on = false;
    int b1 = 0, s1 = 0, c1 = 0, s2 = 0, c2 = 0;
    long long grid = 0;
};
// units1 / units2: what the pieces divide -- components of the two mixtures (log-pdf; units2 = 0 without a second one) or
// groups of 16 components (responsibilities); min_units: smallest piece
SplitPlan split_plan(const PmcKernelSet *ks, long long nblocks, int units1, int units2, int min_units, int tail_min_units = 0)
{
    SplitPlan sp;
    if (!g_split || nblocks < 1 || min_units < 1) return sp;
    const long long slots = (long long)device_cus() * split_slots_per_cu(ks->dim);
    if ((double)nblocks > g_split_max_rounds * (double)slots) return sp;
    long long bt;
    int want;
    if (nblocks <= slots) {                                 // the launch does not fill the chip: every block in pieces
        bt = nblocks;
        want = (int)ceil_div((long long)std::ceil(g_split_fill * (double)slots), nblocks);
        if (want > tun().split_max_pieces) want = tun().split_max_pieces;
    } else {                                                // the last round(s) in pieces
        bt = nblocks % slots + (long long)(g_split_tail_rounds * (double)slots);
        if (bt > nblocks) bt = nblocks;
        if (bt < 1) return sp;
        want = g_split_tail_pieces;
        if (tail_min_units > min_units) min_units = tail_min_units;
    }
    int s1 = units1 / min_units;
    if (s1 > want) s1 = want;
    if (s1 < 1) s1 = 1;
    const int c1 = (int)ceil_div(units1, s1);
    s1 = (int)ceil_div(units1, c1);
    // the second mixture by the same rule on its own: its pieces are then the ones a call on that mixture alone would
    // make, and pmc_importance_weights' log P stays bitwise pmc_mixture_logpdf(target)'s (pmc_hip.h)
    int s2 = 0, c2 = 0;
    if (units2 > 0) {
        s2 = units2 / min_units;
        if (s2 > want) s2 = want;
        if (s2 < 1) s2 = 1;
        c2 = (int)ceil_div(units2, s2);
        s2 = (int)ceil_div(units2, c2);
    }
    if (s1 + s2 < 2) return sp;
    // A launch that fills the chip: only where the FIRST mixture is cut (a piece for the target mixture alone costs a second
    // load of the block's samples for a handful of components: K = 8 + 4: +3 ... +9 %) and not at compiled dimension 64 (the
    // split kernel spills a little more than k_logpdf<64>, which its whole blocks pay: K = 32: +2.6 %) --
    // scripts/split_regression_check.py, profiles/r06_split_regression_check.txt
    if (nblocks > slots && (s1 < 2 || ks->dim > 48)) return sp;
    const int per_block = (s1 + s2 > units1) ? s1 + s2 : units1;      // (the responsibilities keep three numbers per group)
    if (bt > PMC_SPLIT_MAX_BLOCKS) bt = PMC_SPLIT_MAX_BLOCKS;
    if (bt * per_block > PMC_SPLIT_MAX_PIECES) bt = PMC_SPLIT_MAX_PIECES / per_block;
    if (bt < 1) return sp;
    sp.on = true;
    sp.b1 = (int)(nblocks - bt);
    sp.s1 = s1; sp.c1 = c1; sp.s2 = s2; sp.c2 = c2;
    sp.grid = (nblocks - bt) + bt * (s1 + s2);
    return sp;
}
// the pieces' region of the workspace: behind everything else (offset: split_offset below)
size_t split_bytes(long long N, int K)
{
    long long blocks = ceil_div(ceil_div(N > 0 ? N : 1, PMC_TILE), PMC_A_WAVES);
    if (blocks > PMC_SPLIT_MAX_BLOCKS) blocks = PMC_SPLIT_MAX_BLOCKS;
    const size_t a = (size_t)blocks * 2 * (size_t)K * (2 * PMC_A_WAVES * 64 * sizeof(double));
    const size_t b = (size_t)PMC_SPLIT_MAX_PIECES * (3 * PMC_A_WAVES * 64 * sizeof(double));
    return a < b ? a : b;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int pmc_abi_version(void) { return PMC_ABI_VERSION; }
const char *pmc_last_error(void) { return g_err; }
// (not in the header: lets the handle layer, pmc_ctx.hip, report through the same thread-local message)
int pmc_internal_fail(int code, const char *msg) { return fail(code, "%s", msg ? msg : ""); }

int pmc_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(PMC_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int pmc_device_arch(int device, char *buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(PMC_EINVAL, "pmc_device_arch: NULL buffer");
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hipfail(e, "hipGetDeviceProperties");
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return PMC_OK;
}

int pmc_max_dim(void) { return PMC_BIG_MAX_DIM; }
int pmc_max_compiled_dim(void) { return PMC_MAX_DIM; }

int pmc_padded_dim(int D)
{
    if (D < 1) return fail(PMC_EINVAL, "dimension must be >= 1 (got %d)", D);
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    return ks->dim;
}

int64_t pmc_pack_stride(int D)
{
    const int Dp = pmc_padded_dim(D);
    if (Dp < 0) return Dp;
    return pmc_pack_stride_c(Dp);
}

int pmc_tile(void) { return PMC_TILE; }

int64_t pmc_tile_buffer_len(int64_t N, int K)
{
    if (N < 0 || K < 1) return fail(PMC_EINVAL, "pmc_tile_buffer_len: bad N/K");
    return ceil_div(N, PMC_TILE) * (int64_t)K * PMC_TILE;
}

int64_t pmc_stats_stride(int D)
{
    if (D < 1) return fail(PMC_EINVAL, "dimension must be >= 1 (got %d)", D);
    return pmc_stats_stride_c(D);
}

// bytes one call with exactly K components lays out (the regions' offsets are functions of (N, K): stats_geom / gemm_geom
// halve their chunk counts where ceil(K / 32) steps up, so this is NOT monotone in K)
static size_t workspace_bytes_nosplit(long long N, int K, const PmcKernelSet *ks);
// the pieces' region (split_plan) starts behind everything else a call with these N, K places in the workspace
static size_t split_offset(long long N, int K, const PmcKernelSet *ks)
{
    return (workspace_bytes_nosplit(N, K, ks) + 255) & ~(size_t)255;
}
static size_t workspace_bytes_exact(long long N, int K, const PmcKernelSet *ks)
{
    return split_offset(N, K, ks) + (ks->logpdf_split ? split_bytes(N, K) : 0);
}
static size_t workspace_bytes_nosplit(long long N, int K, const PmcKernelSet *ks)
{
    const size_t stats = mgemm_offset(N, K, ks) + mgemm_bytes(N, K, ks);
    const size_t scal = scalar_partials_bytes(N) +
                        (size_t)ceil_div(N > 0 ? N : 1, PMC_TILE) * K * 2 * sizeof(double);
    size_t total = stats > scal ? stats : scal;
    if (ks->dim <= PMC_FUSED_MAX_DIM && (K <= PMC_FUSED_MAX_K || fused_reg(ks->dim, K))) {
        const FusedGeom f = fused_geom(N > 0 ? N : 1, K, ks->dim);
        const size_t fused = ((size_t)f.nchunks * K * pmc_stats_stride_c(ks->dim) + (size_t)f.grid * PMC_NSCALARS +
                              (size_t)f.grid * f.tpr * K * 2) * sizeof(double);        // + Student-t dof sums
        if (fused > total) total = fused;
    }
    return total;
}

// The documented contract sizes ONE workspace for the largest component count of a call ("workspace for max(K,
// K_target)", pmc_importance_weights) while every launch places its regions by its own K: the size is therefore the
// running maximum over all K' <= K (advice r4: D = 40, N = 2e5 laid out 62.1 MB for K = 32 and 36.3 MB for K = 33; a
// proposal of 32 components weighted against a target of 33 wrote 25 MB past a workspace sized by the contract).
int64_t pmc_workspace_bytes(int64_t N, int K, int D)
{
    if (N < 0 || K < 1) return fail(PMC_EINVAL, "pmc_workspace_bytes: bad N/K");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    // (front-ends ask once per call: the last answer of this thread is kept; the options a geometry depends on are
    //  compile-time tables of the kernel set, not pmc_configure's)
    static thread_local struct { int64_t N; int K, D; int64_t bytes; } last = {-1, 0, 0, 0};
    if (last.N == N && last.K == K && last.D == D) return last.bytes;
    size_t total = 0;
    for (int k = 1; k <= K; ++k) {
        const size_t b = workspace_bytes_exact(N, k, ks);
        if (b > total) total = b;
    }
    last = {N, K, D, (int64_t)((total + 255) & ~(size_t)255)};
    return last.bytes;
}

int pmc_pack_components(int K, int D, const double *mu, const double *prec, const double *c0,
                        const double *c1, const double *c2, const double *c3, const double *weight,
                        const int32_t *column, double *pack)
{
    if (K < 1 || !mu || !prec || !pack) return fail(PMC_EINVAL, "pmc_pack_components: bad argument");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    const int Dp = ks->dim, Tp = pmc_tri(Dp), stride = pmc_pack_stride_c(Dp);
    // The K factorisations are independent: from ~1e6 multiply-adds on they are spread over a few host threads (K = 128,
    // D = 40: 0.45 -> 0.1 ms; part of every iteration's host share at an 8-GPU shard size).  A component that does not
    // factorise is reported as before: the lowest such index, its pivot and its value.
    struct Bad { int k, i; double s; };
    std::vector<Bad> bad_of;                                   // one slot per worker
    auto range = [&](int k0, int k1, Bad *bad) {
    std::vector<double> R((size_t)D * D);
    for (int k = k0; k < k1; ++k) {
        double *pk = pack + (size_t)k * stride;
        std::memset(pk, 0, sizeof(double) * stride);
        for (int j = 0; j < D; ++j) pk[j] = mu[(size_t)k * D + j];
        // upper Cholesky factor: precision = R^T R  (row by row, reading the upper triangle)
        const double *A = prec + (size_t)k * D * D;
        for (int i = 0; i < D; ++i) {
            for (int j = i; j < D; ++j) {
                double s = A[(size_t)i * D + j];
                for (int l = 0; l < i; ++l) s -= R[(size_t)l * D + i] * R[(size_t)l * D + j];
                if (j == i) {
                    if (!(s > 0.0) || !std::isfinite(s)) {
                        *bad = Bad{k, i, s};
                        return;
                    }
                    R[(size_t)i * D + i] = std::sqrt(s);
                } else {
                    R[(size_t)i * D + j] = s / R[(size_t)i * D + i];
                }
            }
        }
        int idx = Dp;
        if (pmc_engine(Dp) == PMC_ENG_DPP) {
            // unit-diagonal form U_ij = R_ij / R_ii (j > i), s_i = R_ii^2, rows in pairs in the order the
            // DPP engine consumes them:  U_i,i+1 | U_i+1,j U_i,j (j = i+2 ..) | s_i s_i+1 ; padding is zero
            auto U = [&](int i, int j) { return (i < D && j < D) ? R[(size_t)i * D + j] / R[(size_t)i * D + i] : 0.0; };
            auto S = [&](int i) { return i < D ? R[(size_t)i * D + i] * R[(size_t)i * D + i] : 0.0; };
            for (int i = 0; i < Dp; i += 2) {
                if (i + 1 < Dp) {
                    pk[idx++] = U(i, i + 1);
                    for (int j = i + 2; j < Dp; ++j) {
                        pk[idx++] = U(i + 1, j);
                        pk[idx++] = U(i, j);
                    }
                    pk[idx++] = S(i);
                    pk[idx++] = S(i + 1);
                } else {
                    pk[idx++] = S(i);
                }
            }
        } else {
            // packed row-major (i, j>=i) for the compiled dimension; padding rows/cols are zero
            for (int i = 0; i < Dp; ++i)
                for (int j = i; j < Dp; ++j, ++idx)
                    pk[idx] = (i < D && j < D) ? R[(size_t)i * D + j] : 0.0;
        }
        double *c = pk + Dp + Tp;
        c[0] = c0 ? c0[k] : 0.0;
        c[1] = c1 ? c1[k] : 0.0;
        c[2] = c2 ? c2[k] : 0.0;
        c[3] = c3 ? c3[k] : 0.0;
        c[4] = weight ? weight[k] : 1.0;
        const long long col = column ? (long long)column[k] : (long long)k;
        std::memcpy(&c[5], &col, sizeof(col));
    }
    };
    int nt = 1;
    if ((double)K * D * D * D / 6.0 >= 1e6) {
        const unsigned hw = std::thread::hardware_concurrency();
        nt = (int)(hw ? hw : 1);
        if (nt > 8) nt = 8;
        if (nt > K / 4) nt = K / 4;
        if (nt < 1) nt = 1;
    }
    bad_of.assign((size_t)nt, Bad{-1, 0, 0.0});
    if (nt == 1) {
        range(0, K, &bad_of[0]);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back(range, (int)((long long)K * t / nt), (int)((long long)K * (t + 1) / nt), &bad_of[(size_t)t]);
        for (std::thread &t : th) t.join();
    }
    for (const Bad &b : bad_of)                                 // (workers hold ascending ranges: the first hit is the lowest index)
        if (b.k >= 0)
            return fail(PMC_ENOTPOSDEF, "precision matrix of component %d is not positive definite (pivot %d = %g)", b.k, b.i, b.s);
    return PMC_OK;
}

// ---- the K-sized host steps on the device (include/pmc_hip.h: "device-side packs and conversion") ---------------------
int pmc_pack_components_device(int K, int D, const double *d_mu, const double *d_prec, const double *d_c0, const double *d_c1,
                               const double *d_c2, const double *d_c3, const double *d_weight, const int32_t *d_column,
                               double *d_pack, double *d_status, const double *d_shift, double *d_shift_pack, void *stream)
{
    if (K < 1 || !d_mu || !d_prec || !d_pack || !d_status || ((d_shift != nullptr) != (d_shift_pack != nullptr)))
        return fail(PMC_EINVAL, "pmc_pack_components_device: bad argument");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    if (D > PMC_MAX_DIM || pmc_engine(ks->dim) == PMC_ENG_DPP)
        return fail(PMC_EINVAL, "pmc_pack_components_device: compiled dimensions up to %d in the row-major layout only (D = %d): "
                    "build the pack on the host (pmc_pack_components)", PMC_MAX_DIM, D);
    hipLaunchKernelGGL(k_pack_build, dim3((unsigned)(d_shift ? 2 * K : K)), dim3(64), sizeof(double) * 2 * (size_t)D * D,
                       (hipStream_t)stream, d_mu, d_prec, d_c0, d_c1, d_c2, d_c3, d_weight, (const int *)d_column, K, D, ks->dim,
                       pmc_pack_stride_c(ks->dim), d_pack, d_status, d_shift, d_shift_pack, PmcVbExpect{});
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_pack_build launch");
    return PMC_OK;
}

// pmc_pack_components_device for the posterior of a VB state, the expectations of the E-step formed in the same launch
int pmc_vb_pack_device(int K, int D, const pmc_vb_fields *f, const double *d_psi_parts, double *d_pack, double *d_status,
                       const double *d_shift, double *d_shift_pack, void *stream)
{
    if (K < 1 || !f || !f->m || !f->W || !f->beta || !f->nu || !f->log_det_W || !f->ln_lambda || !f->ln_pi || !d_psi_parts || !d_pack ||
        !d_status || ((d_shift != nullptr) != (d_shift_pack != nullptr)))
        return fail(PMC_EINVAL, "pmc_vb_pack_device: bad argument");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    if (D > PMC_MAX_DIM || pmc_engine(ks->dim) == PMC_ENG_DPP)
        return fail(PMC_EINVAL, "pmc_vb_pack_device: compiled dimensions up to %d in the row-major layout only (D = %d)", PMC_MAX_DIM, D);
    PmcVbExpect vbx;
    vbx.on = 1;
    vbx.beta = f->beta; vbx.nu = f->nu; vbx.log_det_W = f->log_det_W; vbx.parts = d_psi_parts;
    vbx.ln_lambda = f->ln_lambda; vbx.ln_pi = f->ln_pi;
    // (D ln 2 pi with the host's log(2 pi): the logarithm of the DOUBLE 2 pi, as pmc_vb_estep and variational.py compute it)
    vbx.d_ln_2pi = D * 0x1.d67f1c864beb4p+0;
    hipLaunchKernelGGL(k_pack_build, dim3((unsigned)(d_shift ? 2 * K : K)), dim3(64), sizeof(double) * 2 * (size_t)D * D,
                       (hipStream_t)stream, (const double *)f->m, (const double *)f->W, (const double *)nullptr, (const double *)nullptr,
                       (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, (const int *)nullptr, K, D, ks->dim,
                       pmc_pack_stride_c(ks->dim), d_pack, d_status, d_shift, d_shift_pack, vbx);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_pack_build (VB) launch");
    return PMC_OK;
}

int pmc_pack_status(int K, const double *h_status)
{
    if (K < 1 || !h_status) return fail(PMC_EINVAL, "pmc_pack_status: bad argument");
    for (int k = 0; k < K; ++k)
        if (h_status[k] != 0.0)
            return fail(PMC_ENOTPOSDEF, "precision matrix of component %d is not positive definite (pivot %d = %g)", k,
                        (int)h_status[k] - 1, h_status[K + k]);
    return PMC_OK;
}

int pmc_pack_means_device(int K, int D, const double *d_mu, double *d_pack, void *stream)
{
    if (K < 1 || !d_mu || !d_pack) return fail(PMC_EINVAL, "pmc_pack_means_device: bad argument");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks || D > PMC_MAX_DIM) return fail(PMC_EINVAL, "pmc_pack_means_device: compiled dimensions only (D = %d)", D);
    // (the means half of k_pack_build alone: its first K blocks are skipped by an offset of -K in the block index)
    hipLaunchKernelGGL(k_pack_build, dim3((unsigned)K), dim3(64), 0, (hipStream_t)stream, (const double *)nullptr,
                       (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr,
                       (const double *)nullptr, (const double *)nullptr, (const int *)nullptr, 0, D, ks->dim,
                       pmc_pack_stride_c(ks->dim), (double *)nullptr, (double *)nullptr, d_mu, d_pack, PmcVbExpect{});
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_pack_build (means) launch");
    return PMC_OK;
}

int64_t pmc_convert_stats_len(int K, int D)
{
    if (K < 1 || D < 1) return fail(PMC_EINVAL, "pmc_convert_stats_len: bad K / D");
    return (int64_t)K * (2 + 2 * (int64_t)D + (int64_t)D * D) + PMC_NSCALARS;
}

int pmc_convert_stats_device(int K, int D, const double *d_stats, const double *d_shift, const double *d_n_cov,
                             const double *d_scalars, double *d_out, void *stream)
{
    if (K < 1 || D < 1 || !d_stats || !d_shift || !d_out) return fail(PMC_EINVAL, "pmc_convert_stats_device: bad argument");
    hipLaunchKernelGGL(k_convert_stats, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, d_stats, d_shift, d_n_cov, K, D,
                       d_scalars, d_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_convert_stats launch");
    return PMC_OK;
}

int pmc_pack_means(int K, int D, const double *mu, double *pack)
{
    if (K < 1 || !mu || !pack) return fail(PMC_EINVAL, "pmc_pack_means: bad argument");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    const int Dp = ks->dim, stride = pmc_pack_stride_c(Dp);
    std::memset(pack, 0, sizeof(double) * (size_t)K * stride);
    for (int k = 0; k < K; ++k) {
        double *pk = pack + (size_t)k * stride;
        for (int j = 0; j < D; ++j) pk[j] = mu[(size_t)k * D + j];
        double *c = pk + Dp + pmc_tri(Dp);
        c[4] = 1.0;
        const long long col = k;
        std::memcpy(&c[5], &col, sizeof(col));
    }
    return PMC_OK;
}

// scratch of the scalar finishing kernel -- slices + ticket counter, zeroed once (the counter wraps) -- one per
// (device, stream handle): launches on one stream are ordered, different streams never share it.  A handle that
// stands for several real streams (hipStreamPerThread, the null stream under per-thread default streams) is NOT
// supported for concurrent calls from several threads: give each thread a stream of its own.
struct FinScratch {
    int device;
    hipStream_t stream;
    double *slices;
    unsigned *counter;
    unsigned *tickets;      // PMC_SPLIT_MAX_BLOCKS ticket counters of the split kernels (zero between launches)
};
std::mutex g_fin_mutex;
std::vector<FinScratch> g_fin_all;
const hipStream_t FIN_FREE = (hipStream_t)(intptr_t)-1;         // a released slot (pmc_stream_release), memory kept

FinScratch *fin_scratch(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_fin_mutex);
    std::vector<FinScratch> &all = g_fin_all;
    for (FinScratch &f : all)
        if (f.device == dev && f.stream == st) return &f;
    const size_t bytes = sizeof(double) * FIN_GROUPS * PMC_NSCALARS + 256 + sizeof(unsigned) * PMC_SPLIT_MAX_BLOCKS;
    // hipMemset of device memory runs on the NULL stream and may return before it has run: a launch that follows on a
    // NON-BLOCKING stream is not ordered behind it and could take its tickets from a counter that is zeroed under it
    // (found with the handle layer's own stream: the last-ticket block never came and d_scalars was not written).
    // Wait for it here, once per (device, stream).
    for (FinScratch &f : all)
        if (f.device == dev && f.stream == FIN_FREE) {          // a slot some released stream left behind
            if (hipMemset(f.slices, 0, bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return nullptr;
            f.stream = st;
            return &f;
        }
    all.reserve(256);                                       // (pointers handed out stay valid)
    if (all.size() >= 256) return nullptr;                  // before anything is allocated
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipFree(p);
        return nullptr;
    }
    all.push_back(FinScratch{dev, st, (double *)p, (unsigned *)((char *)p + sizeof(double) * FIN_GROUPS * PMC_NSCALARS),
                             (unsigned *)((char *)p + sizeof(double) * FIN_GROUPS * PMC_NSCALARS + 256)});
    return &all.back();
}

static int finish_scalars(const double *partials, long long nblocks, double *d_scalars,
                          hipStream_t st)
{
    FinScratch *f = fin_scratch(st);
    if (!f) return fail(PMC_EHIP, "finishing scratch allocation failed");
    Timed t(T_FINISH, st, 0.0, 8.0 * PMC_NSCALARS * (double)nblocks);
    hipLaunchKernelGGL(k_finish_scalars, dim3(FIN_GROUPS), dim3(256), 0, st, partials, nblocks, f->slices, f->counter,
                       d_scalars);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_finish_scalars launch");
    return PMC_OK;
}

static int sufficient_stats_impl(const double *d_x, int64_t N, int D, const double *d_pack, int K,
                                 const double *d_u, double *d_stats, void *d_workspace, void *stream, int kind,
                                 const double *d_gscale = nullptr, const double *d_rpack = nullptr);

// fill the split_* fields of a launch from its plan (the pieces' region of the caller's workspace -- sized for Kws, the
// larger component count of the call -- and the stream's ticket counters)
static int split_apply(PmcArgsA &a, const SplitPlan &sp, void *d_workspace, int Kws, const PmcKernelSet *ks, hipStream_t st)
{
    FinScratch *f = fin_scratch(st);
    if (!f) return fail(PMC_EHIP, "finishing scratch allocation failed");
    a.split_b1 = sp.b1; a.split_s1 = sp.s1; a.split_c1 = sp.c1; a.split_s2 = sp.s2; a.split_c2 = sp.c2;
    a.split_part = (double *)((char *)d_workspace + split_offset(a.N, Kws, ks));
    a.split_ticket = f->tickets;
    return PMC_OK;
}

int pmc_mixture_logpdf_keep(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                            int max_init_zero, double *d_out, double *d_individual, int64_t ld,
                            const double *d_log_target, double *d_weights, const double *d_sample_w,
                            double *d_scalars, void *d_workspace, double *d_maha_tiles, void *stream)
{
    TuneScope options;                                     // one snapshot of the options for the whole call
    if (N < 0 || K < 1 || !d_pack) return fail(PMC_EINVAL, "pmc_mixture_logpdf: bad N/K/pack");
    if (kind != PMC_KIND_GAUSS && kind != PMC_KIND_STUDENT_T)
        return fail(PMC_EINVAL, "pmc_mixture_logpdf: kind must be GAUSS or STUDENT_T (got %d)", kind);
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    if (d_individual && ld < K) return fail(PMC_EINVAL, "pmc_mixture_logpdf: ld (%lld) < K (%d)", (long long)ld, K);
    if (d_log_target && !d_weights) return fail(PMC_EINVAL, "pmc_mixture_logpdf: d_log_target needs d_weights");
    if (d_scalars && !d_workspace) return fail(PMC_EINVAL, "pmc_mixture_logpdf: d_scalars needs d_workspace");
    if (N > 0 && !d_x) return fail(PMC_EINVAL, "pmc_mixture_logpdf: d_x is NULL");
    hipStream_t st = (hipStream_t)stream;
    const long long nblocks = ceil_div(ceil_div(N, PMC_TILE), PMC_A_WAVES);
    if (nblocks > 0) {
        PmcArgsA a;
        std::memset(&a, 0, sizeof(a));
        a.x = d_x; a.N = N; a.dreal = D; a.pack = d_pack; a.K = K; a.max_init_zero = max_init_zero;
        a.ld = ld; a.out = d_out; a.individual = d_individual; a.log_target = d_log_target;
        a.weights = d_weights; a.sample_w = d_sample_w; a.atile = d_maha_tiles;
        a.partials = d_scalars ? (double *)d_workspace : nullptr;
        Timed t(T_LOGPDF, st, flops_pairs((double)N, K, D),
                8.0 * N * (D + 1 + (d_individual ? K : 0) + (d_maha_tiles ? K : 0)));
        // D >= 32: the forms of all components as one matrix product where its guard allows (pmc_mgemm.hip), the exact
        // kernel behind it for the workgroups it refused
        // (with `individual` too since round 5: the N x K matrix is written from the accumulator layout)
        // (kept forms too since round 5: written from the accumulator layout, recovered from the value)
        const int nct = (d_workspace && !max_init_zero && ks->padded != 2 && (d_out || !d_individual))
                            ? mgemm_pick(ks, N, K, false) : 0;
        hipError_t e = hipSuccess;
        if (nct) e = mgemm_run(ks, nct, kind, a, a, d_workspace, st, true);
        if (e != hipSuccess) return hipfail(e, "k_mgemm launch");
        // small batches, and the last round of larger ones: the components of a block in pieces (split_plan)
        const SplitPlan sp = (!nct && d_workspace && ks->logpdf_split) ? split_plan(ks, nblocks, K, 0, split_min_units(ks), split_tail_min_units(ks))
                                                                       : SplitPlan();
        if (sp.on) {
            const int rc = split_apply(a, sp, d_workspace, K, ks, st);
            if (rc != PMC_OK) return rc;
            e = ks->logpdf_split(kind, kind, a, (unsigned)sp.grid, st);
        } else {
            e = ks->padded == 2 ? big_logpdf(ks, kind, kind, a, D, st) : ks->logpdf(kind, kind, a, (unsigned)nblocks, st);
        }
        if (e != hipSuccess) return hipfail(e, "k_logpdf launch");
    }
    if (d_scalars) return finish_scalars((const double *)d_workspace, nblocks, d_scalars, st);
    return PMC_OK;
}

int pmc_mixture_logpdf(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                       int max_init_zero, double *d_out, double *d_individual, int64_t ld,
                       const double *d_log_target, double *d_weights, const double *d_sample_w,
                       double *d_scalars, void *d_workspace, void *stream)
{
    return pmc_mixture_logpdf_keep(d_x, N, D, d_pack, K, kind, max_init_zero, d_out, d_individual, ld, d_log_target,
                                   d_weights, d_sample_w, d_scalars, d_workspace, nullptr, stream);
}

static int importance_weights_impl(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                                   const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                   double *d_log_target_out, double *d_weights, const double *d_sample_w,
                                   double *d_scalars, void *d_workspace, double *d_maha_tiles, double *d_u,
                                   double *d_vsums, void *stream, double *d_gscale = nullptr, int K_live = 0)
{
    TuneScope options;                                     // one snapshot of the options for the whole call
    if (N < 0 || K < 1 || K_target < 1 || !d_pack || !d_target_pack)
        return fail(PMC_EINVAL, "pmc_importance_weights: bad N/K/pack");
    if ((kind != PMC_KIND_GAUSS && kind != PMC_KIND_STUDENT_T) ||
        (target_kind != PMC_KIND_GAUSS && target_kind != PMC_KIND_STUDENT_T))
        return fail(PMC_EINVAL, "pmc_importance_weights: kinds must be GAUSS or STUDENT_T");
    if (N > 0 && (!d_x || !d_weights)) return fail(PMC_EINVAL, "pmc_importance_weights: d_x / d_weights is NULL");
    if (d_scalars && !d_workspace) return fail(PMC_EINVAL, "pmc_importance_weights: d_scalars needs d_workspace");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    hipStream_t st = (hipStream_t)stream;
    const long long nblocks = ceil_div(ceil_div(N, PMC_TILE), PMC_A_WAVES);
    long long dof_sum_count = 0;                           // partial sums of k_dof_sums (0: the exact kernel's, one per tile)
    if (nblocks > 0) {
        PmcArgsA a;
        std::memset(&a, 0, sizeof(a));
        a.x = d_x; a.N = N; a.dreal = D; a.pack = d_pack; a.K = K; a.ld = K;
        a.pack2 = d_target_pack; a.K2 = K_target; a.log_target_out = d_log_target_out;
        a.out = d_out; a.weights = d_weights; a.sample_w = d_sample_w; a.atile = d_maha_tiles; a.u = d_u;
        a.partials = d_scalars ? (double *)d_workspace : nullptr;
        if (d_u && kind == PMC_KIND_STUDENT_T) a.vpartials = (double *)((char *)d_workspace + scalar_partials_bytes(N));
        a.gscale = d_u ? d_gscale : nullptr;               // (the exact kernel's u is complete: it writes ones there)
        a.ku = (d_u && K_live > 0 && K_live < K) ? K_live : 0;   // pruned components at the end of the pack: no columns of u
        Timed t(T_LOGPDF, st, flops_pairs((double)N, K + K_target, D),
                8.0 * N * (D + 1 + (d_maha_tiles ? K : 0) + (d_u ? K : 0)));
        // D >= 32: the proposal's forms as one matrix product (pmc_mgemm.hip).  The target mixture -- a handful of
        // components, which would pad a pass of 32 / 64 -- goes first, through the exact kernel, into log P; the matrix
        // kernel reads it as given target values; the two-mixture exact kernel behind does the workgroups the guard refused.
        // (an emitting Student-t pass too since round 6: u' = rho' gamma from the epilogue, the sums of the degree-of-freedom
        //  condition from k_dof_sums behind it -- they need the row's factor, which a pass does not have yet)
        int nct = (d_workspace && ks->padded != 2 && (!d_u || d_gscale)) ? mgemm_pick(ks, N, K, d_u != nullptr) : 0;
        const bool dof_kernel = nct && d_u && kind == PMC_KIND_STUDENT_T;
        const int Ku = a.ku > 0 ? a.ku : K;
        long long dof_chunks = 0, dof_tpc = 0;
        if (dof_kernel) {
            // its partial sums stand where the exact kernel's per-tile sums would: behind the scalar partials, in front of
            // everything the matrix kernel keeps in the workspace
            const long long ntl = ceil_div(N, PMC_TILE), G = ceil_div(Ku, PMC_RESP_GROUP);
            const size_t spb = scalar_partials_bytes(N), lim = gscale_offset(N, K, ks);
            const long long room = lim > spb ? (long long)((lim - spb) / (sizeof(double) * 2 * (size_t)Ku)) : 0;
            dof_chunks = ceil_div(2048, G);
            if (dof_chunks > ntl) dof_chunks = ntl;
            if (dof_chunks > room) dof_chunks = room;
            if (dof_chunks < 1) nct = 0;                   // (no room: the exact kernel)
            else {
                dof_tpc = ceil_div(ntl, dof_chunks);
                dof_chunks = ceil_div(ntl, dof_tpc);
            }
        }
        hipError_t e = hipSuccess;
        if (nct) {
            double *lt = d_log_target_out ? d_log_target_out
                                          : (double *)((char *)d_workspace + mgemm_offset(N, K, ks) + mgemm_region(N, K, ks).lt);
            PmcArgsA tg;
            std::memset(&tg, 0, sizeof(tg));
            tg.x = d_x; tg.N = N; tg.dreal = D; tg.pack = d_target_pack; tg.K = K_target; tg.ld = K_target; tg.out = lt;
            e = ks->logpdf(target_kind, target_kind, tg, (unsigned)nblocks, st);
            if (e != hipSuccess) return hipfail(e, "k_logpdf (target) launch");
            if (dof_kernel) {
                // log q of every sample is needed behind the pass: into the caller's buffer or the region's own; the exact
                // kernel behind (the workgroups the guard refused) leaves its sums to k_dof_sums too
                if (!a.out) a.out = (double *)((char *)d_workspace + mgemm_offset(N, K, ks) + mgemm_region(N, K, ks).lse);
                a.vpartials = nullptr;
            }
            PmcArgsA ga = a;
            ga.pack2 = nullptr; ga.K2 = 0; ga.log_target_out = nullptr; ga.log_target = lt; ga.vpartials = nullptr;
            e = mgemm_run(ks, nct, kind, ga, a, d_workspace, st, d_u == nullptr || a.ku > 0, d_u != nullptr);
            if (e != hipSuccess) return hipfail(e, "k_mgemm launch");
        }
        const SplitPlan sp = (!nct && !d_u && d_workspace && ks->logpdf_split)
                                 ? split_plan(ks, nblocks, K, K_target, split_min_units(ks), split_tail_min_units(ks)) : SplitPlan();
        // (A/B build -DPMC_TWO_PER_LANE with PMC_AB_TWO_PER_LANE=1 in the environment: two samples per lane, round 6)
        static const bool two_per_lane = std::getenv("PMC_AB_TWO_PER_LANE") != nullptr;
        if (two_per_lane && !nct && !d_u && !d_maha_tiles && !d_sample_w && !d_log_target_out && ks->logpdf2 && kind == PMC_KIND_GAUSS &&
            target_kind == PMC_KIND_GAUSS &&
            ks->logpdf2(a, (unsigned)ceil_div(nblocks, 2), st) == hipSuccess) {
            e = hipSuccess;
        } else if (sp.on) {
            const int rc = split_apply(a, sp, d_workspace, K > K_target ? K : K_target, ks, st);
            if (rc != PMC_OK) return rc;
            e = ks->logpdf_split(kind, target_kind, a, (unsigned)sp.grid, st);
        } else {
            e = ks->padded == 2 ? big_logpdf(ks, kind, target_kind, a, D, st)
                                : ks->logpdf(kind, target_kind, a, (unsigned)nblocks, st);
        }
        if (e != hipSuccess) return hipfail(e, "k_logpdf launch");
        if (nct && dof_kernel) {
            PmcArgsV v;
            std::memset(&v, 0, sizeof(v));
            v.u = d_u; v.gscale = d_gscale; v.weights = d_weights; v.lse = a.out; v.N = N; v.K = Ku; v.dreal = D;
            v.pack = d_pack; v.stride = pmc_pack_stride_c(ks->dim); v.coff = ks->dim + pmc_tri(ks->dim);
            v.ntiles = ceil_div(N, PMC_TILE); v.tiles_per_chunk = (int)dof_tpc;
            v.vpartials = (double *)((char *)d_workspace + scalar_partials_bytes(N));
            e = pmc_launch_dof_sums(v, (unsigned)ceil_div(Ku, PMC_RESP_GROUP), (unsigned)dof_chunks, st);
            if (e != hipSuccess) return hipfail(e, "k_dof_sums launch");
            dof_sum_count = dof_chunks;
        }
    }
    if (d_u && kind == PMC_KIND_STUDENT_T) {
        const int Ku = (K_live > 0 && K_live < K) ? K_live : K;          // columns of u: the components with a weight
        if (N == 0) {
            hipError_t e0 = hipMemsetAsync(d_vsums, 0, sizeof(double) * 2 * (size_t)Ku, st);
            if (e0 != hipSuccess) return hipfail(e0, "hipMemsetAsync");
        } else {
            hipLaunchKernelGGL(k_finish_vsums, dim3((unsigned)(2 * Ku)), dim3(256), 0, st,
                               (const double *)((char *)d_workspace + scalar_partials_bytes(N)),
                               dof_sum_count ? dof_sum_count : ceil_div(N, PMC_TILE), Ku, d_vsums);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hipfail(e, "k_finish_vsums launch");
        }
    }
    if (d_scalars) return finish_scalars((const double *)d_workspace, nblocks, d_scalars, st);
    return PMC_OK;
}

int pmc_importance_weights_keep(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                                const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                double *d_log_target_out, double *d_weights, const double *d_sample_w,
                                double *d_scalars, void *d_workspace, double *d_maha_tiles, void *stream)
{
    return importance_weights_impl(d_x, N, D, d_pack, K, kind, d_target_pack, K_target, target_kind, d_out,
                                   d_log_target_out, d_weights, d_sample_w, d_scalars, d_workspace, d_maha_tiles, nullptr,
                                   nullptr, stream);
}

int pmc_importance_weights(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                           const double *d_target_pack, int K_target, int target_kind, double *d_out,
                           double *d_log_target_out, double *d_weights, const double *d_sample_w,
                           double *d_scalars, void *d_workspace, void *stream)
{
    return importance_weights_impl(d_x, N, D, d_pack, K, kind, d_target_pack, K_target, target_kind, d_out,
                                   d_log_target_out, d_weights, d_sample_w, d_scalars, d_workspace, nullptr, nullptr, nullptr,
                                   stream);
}

int pmc_importance_weights_emit(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                                const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                double *d_log_target_out, double *d_weights, double *d_scalars, void *d_workspace,
                                double *d_u, double *d_vsums, void *stream)
{
    if (!d_u) return fail(PMC_EINVAL, "pmc_importance_weights_emit: d_u is NULL");
    if (kind == PMC_KIND_STUDENT_T && (!d_vsums || !d_workspace))
        return fail(PMC_EINVAL, "pmc_importance_weights_emit: Student-t needs d_vsums and d_workspace");
    if (D > PMC_MAX_DIM)
        return fail(PMC_EINVAL, "pmc_importance_weights_emit: compiled dimensions only (D <= %d); keep the Mahalanobis "
                                "forms (pmc_importance_weights_keep) and use pmc_estep_from_tiles", PMC_MAX_DIM);
    return importance_weights_impl(d_x, N, D, d_pack, K, kind, d_target_pack, K_target, target_kind, d_out,
                                   d_log_target_out, d_weights, nullptr, d_scalars, d_workspace, nullptr, d_u, d_vsums, stream);
}

int pmc_estep_from_u(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind, const double *d_u,
                     double *d_stats, void *d_workspace, void *stream)
{
    TuneScope options;                                     // one snapshot of the options for the whole call
    if (kind != PMC_KIND_GAUSS && kind != PMC_KIND_STUDENT_T && kind != PMC_KIND_VB)
        return fail(PMC_EINVAL, "pmc_estep_from_u: unknown kind %d", kind);
    return sufficient_stats_impl(d_x, N, D, d_pack, K, d_u, d_stats, d_workspace, stream, kind);
}

int pmc_maha_gemm_tiles(int64_t N, int K, int D)
{
    if (N < 0 || K < 1) return fail(PMC_EINVAL, "pmc_maha_gemm_tiles: bad N/K");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    return ks->padded == 2 ? 0 : mgemm_pick(ks, N, K, false);
}

int pmc_maha_gemm_report(const void *d_workspace, int64_t N, int K, int D, void *stream, double *h_norms,
                         int64_t *h_refused, int64_t *h_workgroups)
{
    if (!d_workspace || !h_norms || !h_refused || !h_workgroups || N < 1 || K < 1)
        return fail(PMC_EINVAL, "pmc_maha_gemm_report: bad argument");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks || ks->padded == 2 || !mgemm_pick(ks, N, K, false))
        return fail(PMC_EINVAL, "pmc_maha_gemm_report: the matrix-product form is not taken for this shape");
    long long refused = 0, nb = 0;
    const int rc = mgemm_report(ks, d_workspace, N, K, (hipStream_t)stream, h_norms, &refused, &nb);
    *h_refused = refused;
    *h_workgroups = nb;
    return rc;
}

int64_t pmc_gscale_len(int64_t N, int K)
{
    if (N < 0 || K < 1) return fail(PMC_EINVAL, "pmc_gscale_len: bad N/K");
    return ceil_div(N, PMC_TILE) * (int64_t)ceil_div(K, PMC_RESP_GROUP) * PMC_TILE;
}

int pmc_importance_weights_emit_grouped(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                                        const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                        double *d_log_target_out, double *d_weights, double *d_scalars, void *d_workspace,
                                        double *d_u, double *d_gscale, double *d_vsums, void *stream)
{
    if (!d_u || !d_gscale) return fail(PMC_EINVAL, "pmc_importance_weights_emit_grouped: d_u / d_gscale is NULL");
    if (kind == PMC_KIND_STUDENT_T && (!d_vsums || !d_workspace))
        return fail(PMC_EINVAL, "pmc_importance_weights_emit_grouped: Student-t needs d_vsums and d_workspace");
    if (D > PMC_MAX_DIM)
        return fail(PMC_EINVAL, "pmc_importance_weights_emit_grouped: compiled dimensions only (D <= %d)", PMC_MAX_DIM);
    return importance_weights_impl(d_x, N, D, d_pack, K, kind, d_target_pack, K_target, target_kind, d_out,
                                   d_log_target_out, d_weights, nullptr, d_scalars, d_workspace, nullptr, d_u, d_vsums, stream,
                                   d_gscale);
}

int pmc_importance_weights_emit_live(const double *d_x, int64_t N, int D, const double *d_pack, int K, int K_live, int kind,
                                     const double *d_target_pack, int K_target, int target_kind, double *d_out,
                                     double *d_log_target_out, double *d_weights, double *d_scalars, void *d_workspace,
                                     double *d_u, double *d_gscale, double *d_vsums, void *stream)
{
    if (!d_u || !d_gscale) return fail(PMC_EINVAL, "pmc_importance_weights_emit_live: d_u / d_gscale is NULL");
    if (K_live < 1 || K_live > K) return fail(PMC_EINVAL, "pmc_importance_weights_emit_live: K_live (%d) not in 1 ... K (%d)", K_live, K);
    if (kind == PMC_KIND_STUDENT_T && (!d_vsums || !d_workspace))
        return fail(PMC_EINVAL, "pmc_importance_weights_emit_live: Student-t needs d_vsums and d_workspace");
    if (D > PMC_MAX_DIM)
        return fail(PMC_EINVAL, "pmc_importance_weights_emit_live: compiled dimensions only (D <= %d)", PMC_MAX_DIM);
    return importance_weights_impl(d_x, N, D, d_pack, K, kind, d_target_pack, K_target, target_kind, d_out,
                                   d_log_target_out, d_weights, nullptr, d_scalars, d_workspace, nullptr, d_u, d_vsums, stream,
                                   d_gscale, K_live);
}

int pmc_estep_from_u_grouped(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind, double *d_u,
                             double *d_gscale, double *d_stats, void *d_workspace, void *stream)
{
    TuneScope options;                                     // one snapshot of the options for the whole call
    if (kind != PMC_KIND_GAUSS && kind != PMC_KIND_STUDENT_T && kind != PMC_KIND_VB)
        return fail(PMC_EINVAL, "pmc_estep_from_u_grouped: unknown kind %d", kind);
    if (!d_gscale) return fail(PMC_EINVAL, "pmc_estep_from_u_grouped: d_gscale is NULL");
    return sufficient_stats_impl(d_x, N, D, d_pack, K, d_u, d_stats, d_workspace, stream, kind, d_gscale);
}

int64_t pmc_maha_tiles_size(int64_t N, int K)
{
    if (N < 0 || K < 1) return 0;
    return (int64_t)ceil_div(N, PMC_TILE) * K * PMC_TILE;
}

int pmc_weight_sums(const double *d_w, int64_t N, double *d_scalars, void *d_workspace, void *stream)
{
    if (N < 0 || !d_scalars || !d_workspace || (N > 0 && !d_w))
        return fail(PMC_EINVAL, "pmc_weight_sums: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long long nblocks = ceil_div(N, 256);
    if (nblocks > 0) {
        hipLaunchKernelGGL(k_weight_sums, dim3((unsigned)nblocks), dim3(256), 0, st, d_w,
                           (long long)N, (double *)d_workspace);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hipfail(e, "k_weight_sums launch");
    }
    return finish_scalars((const double *)d_workspace, nblocks, d_scalars, st);
}

int pmc_propose(const double *d_mu, const double *d_chol, const double *d_dof, const int64_t *d_offsets,
                int K, int D, int64_t N, int64_t first_sample, uint64_t seed, double *d_x,
                int64_t *d_origin, void *stream)
{
    if (N < 0 || K < 1 || !d_mu || !d_chol || !d_offsets || (N > 0 && !d_x))
        return fail(PMC_EINVAL, "pmc_propose: bad argument");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    if (N == 0) return PMC_OK;
    PmcArgsP a;
    std::memset(&a, 0, sizeof(a));
    a.mu = d_mu; a.chol = d_chol; a.dof = d_dof; a.offsets = (const long long *)d_offsets;
    a.K = K; a.dreal = D; a.N = N; a.first_sample = first_sample; a.seed = seed;
    a.x = d_x; a.origin = (long long *)d_origin;
    Timed t(T_PROPOSE, (hipStream_t)stream, (double)N * D * (D + 1), 8.0 * N * (D + 1));
    hipError_t e = ks->propose(a, (unsigned)ceil_div(N, 256), (hipStream_t)stream);
    if (e != hipSuccess) return hipfail(e, "k_propose launch");
    return PMC_OK;
}

int pmc_logsumexp2d(const double *d_a, const double *d_w, int64_t N, int K, double *d_out, void *stream)
{
    if (N < 0 || K < 1 || !d_w || (N > 0 && (!d_a || !d_out)))
        return fail(PMC_EINVAL, "pmc_logsumexp2d: bad argument");
    if (N == 0) return PMC_OK;
    hipLaunchKernelGGL(k_logsumexp2d, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_a, d_w, (long long)N, K, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_logsumexp2d launch");
    return PMC_OK;
}

int pmc_combine_weights(const double *d_q, int64_t N, int T, const double *d_counts, int t,
                        const double *d_omega, double n_total, int log_scale, double *d_out,
                        double *d_flag, void *stream)
{
    if (N < 0 || T < 1 || t < 0 || t >= T || !d_counts || !(n_total > 0) || (N > 0 && (!d_q || !d_omega || !d_out)))
        return fail(PMC_EINVAL, "pmc_combine_weights: bad argument");
    if (N == 0) return PMC_OK;
    hipLaunchKernelGGL(k_combine_weights, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_q, (long long)N, T, d_counts, t, d_omega, n_total, log_scale, d_out, d_flag);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_combine_weights launch");
    return PMC_OK;
}

int pmc_responsibilities(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                         int mode, int max_init_zero, const double *d_sample_w,
                         const int64_t *d_latent, double *d_u, double *d_scratch, double *d_vsums,
                         double *d_r, double *d_log_rho, double *d_exponent, int64_t ld,
                         double *d_scalars, void *d_workspace, void *stream)
{
    if (N < 0 || K < 1 || !d_pack || !d_u) return fail(PMC_EINVAL, "pmc_responsibilities: bad N/K/pack/u");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    if (mode == PMC_RESP_VB) {
        if (kind != PMC_KIND_VB) return fail(PMC_EINVAL, "pmc_responsibilities: mode VB needs kind VB");
    } else if (mode == PMC_RESP_PMC_RB || mode == PMC_RESP_PMC_LATENT) {
        if (kind != PMC_KIND_GAUSS && kind != PMC_KIND_STUDENT_T)
            return fail(PMC_EINVAL, "pmc_responsibilities: PMC modes need kind GAUSS or STUDENT_T");
        if (mode == PMC_RESP_PMC_LATENT && !d_latent)
            return fail(PMC_EINVAL, "pmc_responsibilities: mode LATENT needs d_latent");
        if (kind == PMC_KIND_STUDENT_T && (!d_scratch || !d_vsums || !d_workspace))
            return fail(PMC_EINVAL, "pmc_responsibilities: Student-t needs d_scratch, d_vsums and d_workspace");
    } else {
        return fail(PMC_EINVAL, "pmc_responsibilities: unknown mode %d", mode);
    }
    if ((d_r || d_log_rho || d_exponent) && ld < K)
        return fail(PMC_EINVAL, "pmc_responsibilities: ld (%lld) < K (%d)", (long long)ld, K);
    if (d_scalars && !d_workspace) return fail(PMC_EINVAL, "pmc_responsibilities: d_scalars needs d_workspace");
    if (N > 0 && !d_x) return fail(PMC_EINVAL, "pmc_responsibilities: d_x is NULL");
    hipStream_t st = (hipStream_t)stream;
    const long long ntiles = ceil_div(N, PMC_TILE);
    const long long nblocks = ceil_div(ntiles, PMC_A_WAVES);
    double *vpartials = d_workspace ? (double *)((char *)d_workspace + scalar_partials_bytes(N)) : nullptr;
    if (nblocks > 0) {
        PmcArgsA a;
        std::memset(&a, 0, sizeof(a));
        a.x = d_x; a.N = N; a.dreal = D; a.pack = d_pack; a.K = K; a.max_init_zero = max_init_zero;
        a.mode = mode; a.ld = ld; a.sample_w = d_sample_w; a.latent = (const long long *)d_latent;
        a.u = d_u; a.scratch = d_scratch; a.vpartials = vpartials;
        a.klds = K < pmc_resp_klds(ks->dim) ? K : pmc_resp_klds(ks->dim);
        a.r = d_r; a.log_rho = d_log_rho; a.exponent = d_exponent;
        a.partials = d_scalars ? (double *)d_workspace : nullptr;
        Timed t(T_RESP, st, flops_pairs((double)N, K, D), 8.0 * N * (D + K));
        if (ks->padded == 2) {
            // the Mahalanobis forms go into the responsibility buffer itself: pass 1 of k_resp reads a pair's
            // form and parks its value in the same place
            hipError_t em = big_maha(d_x, N, D, d_pack, K, d_u, st);
            if (em != hipSuccess) return hipfail(em, "k_big_maha launch");
            a.mtile = d_u;
        }
        hipError_t e = ks->resp(kind, a, (unsigned)nblocks, st);
        if (e != hipSuccess) return hipfail(e, "k_resp launch");
    }
    if (kind == PMC_KIND_STUDENT_T) {
        hipLaunchKernelGGL(k_finish_vsums, dim3((unsigned)(2 * K)), dim3(256), 0, st,
                           (const double *)vpartials, ntiles, K, d_vsums);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hipfail(e, "k_finish_vsums launch");
    }
    if (d_scalars) return finish_scalars((const double *)d_workspace, nblocks, d_scalars, st);
    return PMC_OK;
}

// kind: what d_pack describes (pmc_kind) when the caller is an E-step and the fast common-shift form may be tried;
// -1: the per-component-shift kernel, always (the public pmc_sufficient_stats)
// d_gscale: factors k_resp_groups left to be applied to d_u (NULL: d_u is complete); only with the common-shift form
// d_rpack: the pack that holds the triangular factors (for the a-priori test) when d_pack only carries shifts (NULL: d_pack)
static int sufficient_stats_impl(const double *d_x, int64_t N, int D, const double *d_pack, int K,
                                 const double *d_u, double *d_stats, void *d_workspace, void *stream, int kind,
                                 const double *d_gscale, const double *d_rpack)
{
    if (N < 0 || K < 1 || !d_pack || !d_u || !d_stats || !d_workspace)
        return fail(PMC_EINVAL, "pmc_sufficient_stats: bad argument");
    if (N > 0 && !d_x) return fail(PMC_EINVAL, "pmc_sufficient_stats: d_x is NULL");
    if (((uintptr_t)d_x & 7u) != 0) return fail(PMC_EINVAL, "pmc_sufficient_stats: d_x must be 8-byte aligned");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    hipStream_t st = (hipStream_t)stream;
    const int PS = pmc_stats_stride_c(D);
    if (N == 0) {
        hipError_t e = hipMemsetAsync(d_stats, 0, sizeof(double) * (size_t)K * PS, st);
        if (e != hipSuccess) return hipfail(e, "hipMemsetAsync");
        return PMC_OK;
    }
    if (N * (int64_t)D == 1) {
        // a single scalar sample: the tile kernel moves 16-byte pieces, which do not fit into 8 bytes
        hipLaunchKernelGGL(k_stats_single, dim3((unsigned)ceil_div(K, 256)), dim3(256), 0, st, d_x, d_pack,
                           pmc_pack_stride_c(ks->dim), K, d_u, d_stats);
        hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return hipfail(e1, "k_stats_single launch");
        return PMC_OK;
    }
    const StatsGeom g = stats_geom(N, K, ks);
    hipError_t e;
    int *ctl = nullptr;
    int counted = 1;
    const bool use_gemm = gemm_selected(ks, N, K, kind);  // (evaluated once: pmc_configure may run concurrently)
    if (d_gscale && !use_gemm) {
        // factors, but the statistics kernel that applies them is not the one this shape gets: complete u in place, the
        // factors become ones (only a caller-owned pair gets here: pmc_estep decides both halves together)
        const long long ntl = ceil_div(N, PMC_TILE), total_u = ntl * (long long)K * 64;
        const long long glen = ntl * ceil_div(K, PMC_RESP_GROUP) * 64;
        const long long blocks = ceil_div(total_u, 256), gblocks = ceil_div(glen, 256);
        hipLaunchKernelGGL(k_apply_scale, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, (double *)d_u,
                           d_gscale, ntl, K, (const int *)nullptr);
        hipLaunchKernelGGL(k_reset_scale, dim3((unsigned)(gblocks < 2048 ? gblocks : 2048)), dim3(256), 0, st,
                           (double *)d_gscale, glen, (const int *)nullptr);
        e = hipGetLastError();
        if (e != hipSuccess) return hipfail(e, "k_apply_scale launch");
        d_gscale = nullptr;
    }
    if (use_gemm) {
        // The common-shift form first (k_stats_gemm, pmc_stats.hip); the per-component-shift kernel below then
        // returns at once unless the a-posteriori test of k_gemm_convert asks for it.
        if (((uintptr_t)d_u & 15u) != 0) return fail(PMC_EINVAL, "pmc_estep: d_u must be 16-byte aligned");
        const GemmGeom gg = gemm_geom(N, K, ks);
        char *tail = (char *)d_workspace + stats_region_bytes(N, K, ks);
        double *center = (double *)tail;
        ctl = (int *)(tail + (((size_t)ks->dim * sizeof(double) + 255) & ~(size_t)255));
        double *gpart = (double *)d_workspace;
        double *totals = gpart + (size_t)gg.nce * K * ks->gemm_msp;
        const int stride = pmc_pack_stride_c(ks->dim);
        {
            Timed t(T_STATS, st, flops_stats((double)N, K, D), 8.0 * N * (D + K));
            PmcArgsG a;
            std::memset(&a, 0, sizeof(a));
            a.x = d_x; a.N = N; a.dreal = D; a.pack = d_rpack ? d_rpack : d_pack; a.spack = d_pack; a.kind = kind;
            a.limit_prior = 4.0 * g_gemm_limit;
            a.center = center; a.K = K; a.u = d_u; a.gscale = d_gscale; a.partials = gpart;
            a.ntiles = gg.ntiles; a.nchunks = gg.nchunks; a.tiles_per_chunk = gg.tiles_per_chunk;
            a.ngroups = gg.ngroups; a.ncs = gg.ncs; a.ctl = ctl;
            e = ks->stats_gemm(a, gg.grid, st);
            if (e != hipSuccess) return hipfail(e, "k_stats_gemm launch");
        }
        {
            Timed tf(T_FINISH, st, 0.0, 8.0 * gg.nce * K * ks->gemm_msp);
            hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)ceil_div(ks->gemm_msp, 64), (unsigned)K), dim3(256), 0, st,
                               (const double *)gpart, gg.nce, K, ks->gemm_msp, totals, (const int *)ctl);
            e = hipGetLastError();
            if (e != hipSuccess) return hipfail(e, "k_gemm_reduce launch");
            hipLaunchKernelGGL(k_gemm_convert, dim3((unsigned)K), dim3(256), 0, st, (const double *)totals, K, D,
                               ks->gemm_msp, d_pack, stride, (const double *)center, g_gemm_limit, d_stats, ctl);
            e = hipGetLastError();
            if (e != hipSuccess) return hipfail(e, "k_gemm_convert launch");
        }
        counted = 0;                                       // the launches below continue this call's record
    }
    PmcArgsB b;
    std::memset(&b, 0, sizeof(b));
    b.x = d_x; b.N = N; b.dreal = D; b.pack = d_pack; b.K = K; b.u = d_u;
    b.partials = (double *)d_workspace; b.ntiles = g.ntiles; b.nchunks = g.nchunks;
    b.tiles_per_chunk = g.tiles_per_chunk; b.ngroups = g.ngroups; b.ctl = ctl;
    const long long total = (long long)K * PS;
    static const bool ab_skip_fallback = std::getenv("PMC_AB_SKIP_FALLBACK") != nullptr;   // TIMING ONLY (profiles/r06_fallback_launches_ab.txt)
    if (!counted && ab_skip_fallback) return PMC_OK;
    if (!counted) {
        // behind the common-shift form: its fall-back -- the factors k_resp_groups left to the statistics kernel applied to
        // u, the per-component-shift kernel, its finishing reduction -- three launches that return at once unless the
        // control block asks for them, under ONE timing bracket (continuing the statistics' record): launches inside a
        // bracket follow each other without a gap, every bracket of its own put ~10 us in front of its launch
        // (profiles/r03_timeline_gaps_share.txt)
        Timed t(T_STATS, st, 0.0, 0.0, 0);
        if (d_gscale) {
            const GemmGeom gg = gemm_geom(N, K, ks);
            const long long total_u = gg.ntiles * (long long)K * 64;
            const long long blocks = ceil_div(total_u, 256);
            hipLaunchKernelGGL(k_apply_scale, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, (double *)d_u,
                               d_gscale, gg.ntiles, K, (const int *)ctl);
            const long long glen = gg.ntiles * ceil_div(K, PMC_RESP_GROUP) * 64;
            const long long gblocks = ceil_div(glen, 256);
            hipLaunchKernelGGL(k_reset_scale, dim3((unsigned)(gblocks < 2048 ? gblocks : 2048)), dim3(256), 0, st,
                               (double *)d_gscale, glen, (const int *)ctl);
            e = hipGetLastError();
            if (e != hipSuccess) return hipfail(e, "k_apply_scale launch");
        }
        e = ks->stats(b, g.grid, st);
        if (e != hipSuccess) return hipfail(e, "k_stats launch");
        hipLaunchKernelGGL(k_finish_stats, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, st,
                           (const double *)d_workspace, g.nchunks, K, D, ks->dim, d_stats, (const int *)ctl);
        e = hipGetLastError();
        if (e != hipSuccess) return hipfail(e, "k_finish_stats launch");
        return PMC_OK;
    }
    {
        Timed t(T_STATS, st, flops_stats((double)N, K, D), 8.0 * N * (D + K));
        e = ks->stats(b, g.grid, st);
    }
    if (e != hipSuccess) return hipfail(e, "k_stats launch");
    Timed tf(T_FINISH, st, 0.0, 8.0 * g.nchunks * K * pmc_stats_stride_c(ks->dim));
    hipLaunchKernelGGL(k_finish_stats, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, st,
                       (const double *)d_workspace, g.nchunks, K, D, ks->dim, d_stats, (const int *)ctl);
    e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_finish_stats launch");
    return PMC_OK;
}

int pmc_sufficient_stats(const double *d_x, int64_t N, int D, const double *d_pack, int K,
                         const double *d_u, double *d_stats, void *d_workspace, void *stream)
{
    TuneScope options;                                     // one snapshot of the options for the whole call
    return sufficient_stats_impl(d_x, N, D, d_pack, K, d_u, d_stats, d_workspace, stream, -1);
}

int pmc_stream_release(void *stream)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hipfail(e, "hipGetDevice");
    std::lock_guard<std::mutex> lock(g_fin_mutex);
    for (FinScratch &f : g_fin_all)
        if (f.device == dev && f.stream == (hipStream_t)stream) f.stream = FIN_FREE;
    return PMC_OK;
}

static int configure_into(PmcTuning &t, const char *key, double value)
{
    if (!key) return fail(PMC_EINVAL, "pmc_configure: NULL key");
    if (std::strcmp(key, "stats_common_shift_min_k") == 0) {
        if (!(value >= 1.0 && value <= 1e9)) return fail(PMC_EINVAL, "pmc_configure: %s must be >= 1", key);
        t.gemm_min_k = (int)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "big_dim_scratch_bytes") == 0) {
        if (!(value >= 1.0 && value <= 1e15)) return fail(PMC_EINVAL, "pmc_configure: %s must be >= 1", key);
        t.big_scratch_bytes = (size_t)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "estep_grouped_responsibilities") == 0) {
        if (!(value == 0.0 || value == 1.0 || value == 2.0)) return fail(PMC_EINVAL, "pmc_configure: %s is 0, 1 or 2", key);
        t.resp_groups = (int)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "stats_common_shift_min_fill") == 0) {
        if (!(value >= 0.0 && value <= 1.0)) return fail(PMC_EINVAL, "pmc_configure: %s must be in [0, 1]", key);
        t.gemm_min_fill = value;
        return PMC_OK;
    }
    if (std::strcmp(key, "stats_common_shift_min_n") == 0) {
        if (!(value >= 0.0 && value <= 9e18)) return fail(PMC_EINVAL, "pmc_configure: %s must be >= 0", key);
        t.gemm_min_n = (long long)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "maha_gemm_tolerance") == 0) {
        if (!(value >= 0.0 && value <= 1.0)) return fail(PMC_EINVAL, "pmc_configure: %s must be in [0, 1]", key);
        t.mgemm_tol = value;
        return PMC_OK;
    }
    if (std::strcmp(key, "maha_gemm_min_n") == 0) {
        if (!(value >= 0.0 && value <= 9e18)) return fail(PMC_EINVAL, "pmc_configure: %s must be >= 0", key);
        t.mgemm_min_n = (long long)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "split_components") == 0) {
        if (!(value == 0.0 || value == 1.0)) return fail(PMC_EINVAL, "pmc_configure: %s is 0 or 1", key);
        t.split = (int)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "split_min_components") == 0) {
        if (!(value >= 0.0 && value <= 1e6)) return fail(PMC_EINVAL, "pmc_configure: %s must be >= 0", key);
        t.split_min_comps = (int)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "split_tail_pieces") == 0) {
        if (!(value >= 1.0 && value <= 64.0)) return fail(PMC_EINVAL, "pmc_configure: %s must be in [1, 64]", key);
        t.split_tail_pieces = (int)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "split_max_rounds") == 0) {
        if (!(value >= 0.0)) return fail(PMC_EINVAL, "pmc_configure: %s must be >= 0", key);
        t.split_max_rounds = value;
        return PMC_OK;
    }
    if (std::strcmp(key, "split_tail_rounds") == 0) {
        if (!(value >= 0.0 && value <= 64.0)) return fail(PMC_EINVAL, "pmc_configure: %s must be in [0, 64]", key);
        t.split_tail_rounds = value;
        return PMC_OK;
    }
    if (std::strcmp(key, "split_tail_min_components") == 0) {
        if (!(value >= 0.0 && value <= 1e6)) return fail(PMC_EINVAL, "pmc_configure: %s must be >= 0", key);
        t.split_tail_min_comps = (int)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "estep_small_batch_pieces") == 0) {
        if (!(value == 0.0 || value == 1.0)) return fail(PMC_EINVAL, "pmc_configure: %s is 0 or 1", key);
        t.small_grouped = (int)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "split_max_pieces") == 0) {
        if (!(value >= 1.0 && value <= 1024.0)) return fail(PMC_EINVAL, "pmc_configure: %s must be in [1, 1024]", key);
        t.split_max_pieces = (int)value;
        return PMC_OK;
    }
    if (std::strcmp(key, "split_fill") == 0) {
        if (!(value > 0.0 && value <= 64.0)) return fail(PMC_EINVAL, "pmc_configure: %s must be in (0, 64]", key);
        t.split_fill = value;
        return PMC_OK;
    }
    if (std::strcmp(key, "stats_common_shift_limit") == 0) {
        if (!(value >= 0.0)) return fail(PMC_EINVAL, "pmc_configure: %s must be >= 0", key);
        t.gemm_limit = value;
        return PMC_OK;
    }
    return fail(PMC_EINVAL, "pmc_configure: unknown key '%s'", key);
}

// the value of an option in `t` (pmc_option_default: in a default-constructed set)
static int option_get(const PmcTuning &t, const char *key, double *value)
{
    if (!key || !value) return fail(PMC_EINVAL, "pmc_option: NULL argument");
    const struct { const char *name; double v; } all[] = {
        {"stats_common_shift_min_k", (double)t.gemm_min_k}, {"stats_common_shift_limit", t.gemm_limit},
        {"stats_common_shift_min_n", (double)t.gemm_min_n}, {"stats_common_shift_min_fill", t.gemm_min_fill},
        {"estep_grouped_responsibilities", (double)t.resp_groups}, {"big_dim_scratch_bytes", (double)t.big_scratch_bytes},
        {"maha_gemm_tolerance", t.mgemm_tol}, {"maha_gemm_min_n", (double)t.mgemm_min_n},
        {"split_components", (double)t.split}, {"split_min_components", (double)t.split_min_comps},
        {"split_tail_pieces", (double)t.split_tail_pieces}, {"split_max_rounds", t.split_max_rounds},
        {"split_fill", t.split_fill}, {"split_max_pieces", (double)t.split_max_pieces}, {"split_tail_rounds", t.split_tail_rounds},
        {"split_tail_min_components", (double)t.split_tail_min_comps}, {"estep_small_batch_pieces", (double)t.small_grouped}};
    for (const auto &o : all)
        if (std::strcmp(key, o.name) == 0) {
            *value = o.v;
            return PMC_OK;
        }
    return fail(PMC_EINVAL, "pmc_option: unknown key '%s'", key);
}

int pmc_option_default(const char *key, double *value) { return option_get(PmcTuning(), key, value); }

int pmc_option_get(const char *key, double *value) { return option_get(tun(), key, value); }

int pmc_configure(const char *key, double value)
{
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    return configure_into(g_tuning, key, value);
}

// per-context options of the handle layer (pmc_ctx.hip; not in the public header): a context starts from a copy of the
// process-wide options, changes only its copy, and installs it for the calling thread around each of its calls
void *pmc_internal_tuning_new(void)
{
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    return new PmcTuning(g_tuning);
}
void pmc_internal_tuning_free(void *t) { delete (PmcTuning *)t; }
int pmc_internal_tuning_set(void *t, const char *key, double value)
{
    if (!t) return fail(PMC_EINVAL, "NULL options");
    return configure_into(*(PmcTuning *)t, key, value);
}
void pmc_internal_tuning_use(const void *t) { t_tuning = (const PmcTuning *)t; }

// ---- RCCL, opened at run time -------------------------------------------------------------------------
// (the few declarations of rccl.h the library needs; values as in NCCL's public header)
struct PmcNcclId {                                          // ncclUniqueId (passed by value)
    char internal[PMC_COMM_ID_BYTES];
};
typedef struct pmcNcclComm *pmcNcclComm_t;
namespace {
struct RcclApi {
    int (*GetUniqueId)(void *);
    int (*CommInitRank)(pmcNcclComm_t *, int, PmcNcclId, int);
    int (*AllReduce)(const void *, void *, size_t, int, int, pmcNcclComm_t, hipStream_t);
    int (*CommDestroy)(pmcNcclComm_t);
    const char *(*GetErrorString)(int);
    bool ok = false;
};
}  // namespace
struct pmc_comm {
    pmcNcclComm_t comm;
    int rank, world, device;
};
namespace {
constexpr int PMC_NCCL_DOUBLE = 8, PMC_NCCL_SUM = 0;       // ncclFloat64, ncclSum
RcclApi *rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)                          // a copy the process already holds (PyTorch-ROCm's)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        api.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (int (*)(pmcNcclComm_t *, int, PmcNcclId, int))dlsym(h, "ncclCommInitRank");
        api.AllReduce = (int (*)(const void *, void *, size_t, int, int, pmcNcclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
        api.CommDestroy = (int (*)(pmcNcclComm_t))dlsym(h, "ncclCommDestroy");
        api.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy;
    });
    return api.ok ? &api : nullptr;
}
int rcclfail(RcclApi *r, int code, const char *what)
{
    return fail(PMC_EHIP, "%s: %s", what, r->GetErrorString ? r->GetErrorString(code) : "RCCL error");
}
}  // namespace

int pmc_comm_unique_id(void *h_id)
{
    if (!h_id) return fail(PMC_EINVAL, "pmc_comm_unique_id: NULL buffer");
    RcclApi *r = rccl();
    if (!r) return fail(PMC_ENODEVICE, "librccl could not be opened (%s)", dlerror() ? dlerror() : "not found");
    const int rc = r->GetUniqueId(h_id);
    if (rc != 0) return rcclfail(r, rc, "ncclGetUniqueId");
    return PMC_OK;
}

int pmc_comm_init(int rank, int world, const void *h_id, int device, pmc_comm **out)
{
    if (!out || !h_id || world < 1 || rank < 0 || rank >= world) return fail(PMC_EINVAL, "pmc_comm_init: bad argument");
    RcclApi *r = rccl();
    if (!r) return fail(PMC_ENODEVICE, "librccl could not be opened");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hipfail(e, "hipSetDevice");
    PmcNcclId id;
    std::memcpy(id.internal, h_id, PMC_COMM_ID_BYTES);
    pmcNcclComm_t c = nullptr;
    const int rc = r->CommInitRank(&c, world, id, rank);
    if (rc != 0) return rcclfail(r, rc, "ncclCommInitRank");
    *out = new pmc_comm{c, rank, world, device};
    return PMC_OK;
}

int pmc_comm_rank(const pmc_comm *comm, int *rank, int *world)
{
    if (!comm) return fail(PMC_EINVAL, "pmc_comm_rank: NULL communicator");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return PMC_OK;
}

int pmc_comm_allreduce_sum(pmc_comm *comm, double *d_buf, int64_t n, void *stream)
{
    if (!comm || n < 0 || (n > 0 && !d_buf)) return fail(PMC_EINVAL, "pmc_comm_allreduce_sum: bad argument");
    if (n == 0) return PMC_OK;
    RcclApi *r = rccl();
    if (!r) return fail(PMC_ENODEVICE, "librccl could not be opened");
    const int rc = r->AllReduce(d_buf, d_buf, (size_t)n, PMC_NCCL_DOUBLE, PMC_NCCL_SUM, comm->comm, (hipStream_t)stream);
    if (rc != 0) return rcclfail(r, rc, "ncclAllReduce");
    return PMC_OK;
}

int pmc_comm_destroy(pmc_comm *comm)
{
    if (!comm) return PMC_OK;
    RcclApi *r = rccl();
    int rc = 0;
    if (r && comm->comm) rc = r->CommDestroy(comm->comm);
    delete comm;
    if (rc != 0) return rcclfail(r, rc, "ncclCommDestroy");
    return PMC_OK;
}

int pmc_timing_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    g_timing_on = on != 0;
    return PMC_OK;
}

// stream == FIN_FREE: the records of every stream that is not timed for a context of its own (pmc_get_timings);
// else: the records of that stream (pmc_ctx_get_timings)
static int collect_timings(hipStream_t stream, pmc_timing *h_out, int max_entries, int *n_entries)
{
    if (!n_entries || (max_entries > 0 && !h_out)) return fail(PMC_EINVAL, "pmc_get_timings: bad argument");
    std::vector<TimingRec> recs;
    {
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        std::vector<TimingRec> keep;
        for (const TimingRec &r : g_timing_recs) {
            const bool take = stream == FIN_FREE ? !timing_stream_on(r.st) : r.st == stream;
            (take ? recs : keep).push_back(r);
        }
        g_timing_recs.swap(keep);
    }
    pmc_timing acc[T_COUNT];
    std::memset(acc, 0, sizeof(acc));
    for (int i = 0; i < T_COUNT; ++i) snprintf(acc[i].name, sizeof(acc[i].name), "%s", g_timing_names[i]);
    int rc = PMC_OK;
    for (const TimingRec &r : recs) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(r.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) rc = hipfail(e, "pmc_get_timings: event");
        else {
            acc[r.id].calls += r.calls;
            acc[r.id].ms += ms;
            acc[r.id].flops += r.flops;
            acc[r.id].bytes += r.bytes;
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        for (const TimingRec &r : recs) {
            g_timing_pool[r.dev].push_back(r.a);
            g_timing_pool[r.dev].push_back(r.b);
        }
    }
    int n = 0;
    for (int i = 0; i < T_COUNT; ++i) {
        if (acc[i].calls == 0) continue;
        if (n < max_entries) h_out[n] = acc[i];
        ++n;
    }
    *n_entries = n;
    return rc;
}

int pmc_get_timings(pmc_timing *h_out, int max_entries, int *n_entries)
{
    return collect_timings(FIN_FREE, h_out, max_entries, n_entries);
}
// the handle layer's per-context timing (pmc_ctx.hip: pmc_ctx_timing_enable / pmc_ctx_get_timings)
int pmc_internal_timing_stream(void *stream, int on)
{
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    for (size_t i = 0; i < g_timing_streams.size(); ++i)
        if (g_timing_streams[i] == (hipStream_t)stream) {
            if (!on) g_timing_streams.erase(g_timing_streams.begin() + (long)i);
            return PMC_OK;
        }
    if (on) g_timing_streams.push_back((hipStream_t)stream);
    return PMC_OK;
}
int pmc_internal_get_timings(void *stream, pmc_timing *h_out, int max_entries, int *n_entries)
{
    return collect_timings((hipStream_t)stream, h_out, max_entries, n_entries);
}

// What split_plan decides for a shape, for the host-side tests (not in the public header): out[0..9] = on, b1, s1, c1, s2, c2,
// grid, bytes of the pieces' region the launch uses, bytes reserved for it, its offset in the workspace.  resp != 0: the
// responsibility kernel's plan (units = groups of 16 components).
int pmc_internal_split_plan(int64_t N, int K, int K2, int D, int resp, int64_t *out)
{
    TuneScope options;
    if (N < 0 || K < 1 || K2 < 0 || !out) return fail(PMC_EINVAL, "pmc_internal_split_plan: bad argument");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    const long long nblocks = ceil_div(ceil_div(N, PMC_TILE), PMC_A_WAVES);
    const SplitPlan sp = !ks->logpdf_split ? SplitPlan()
                         : (resp ? split_plan(ks, nblocks, (int)ceil_div(K, PMC_RESP_GROUP), 0, 1)
                                 : split_plan(ks, nblocks, K, K2, split_min_units(ks), split_tail_min_units(ks)));
    const int Kws = K > K2 ? K : K2;
    const long long bt = sp.on ? nblocks - sp.b1 : 0;
    const long long used = resp ? bt * ceil_div(K, PMC_RESP_GROUP) * 3 * PMC_A_WAVES * 64 * (long long)sizeof(double)
                                : bt * (sp.s1 + sp.s2) * 2 * PMC_A_WAVES * 64 * (long long)sizeof(double);
    out[0] = sp.on; out[1] = sp.b1; out[2] = sp.s1; out[3] = sp.c1; out[4] = sp.s2; out[5] = sp.c2; out[6] = sp.grid;
    out[7] = used; out[8] = (int64_t)split_bytes(N, Kws); out[9] = (int64_t)split_offset(N, Kws, ks);
    return PMC_OK;
}

int pmc_estep_is_fused(int K, int D, int kind, int mode)
{
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks || K < 1) return 0;
    return fused_eligible(ks, K, kind, mode) ? 1 : 0;
}

int pmc_estep(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind, int mode,
              int max_init_zero, const double *d_sample_w, const int64_t *d_latent, double *d_u,
              double *d_scratch, double *d_vsums, double *d_stats, double *d_scalars, void *d_workspace,
              void *stream)
{
    return pmc_estep_about(d_x, N, D, d_pack, K, kind, mode, max_init_zero, d_sample_w, d_latent, d_u, d_scratch, d_vsums,
                           d_stats, d_scalars, d_workspace, nullptr, stream);
}

int pmc_estep_about(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind, int mode,
                    int max_init_zero, const double *d_sample_w, const int64_t *d_latent, double *d_u,
                    double *d_scratch, double *d_vsums, double *d_stats, double *d_scalars, void *d_workspace,
                    const double *d_shift_pack, void *stream)
{
    TuneScope options;                                     // one snapshot of the options for the whole call
    const double *d_spack = d_shift_pack ? d_shift_pack : d_pack;      // whose means the moments are taken about
    if (N < 0 || K < 1 || !d_pack || !d_stats || !d_scalars || !d_workspace)
        return fail(PMC_EINVAL, "pmc_estep: bad N/K/pack/stats/scalars/workspace");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    if (N == 0) {                                          // no samples: zero statistics, zero sums
        hipStream_t st0 = (hipStream_t)stream;
        hipError_t e0 = hipMemsetAsync(d_stats, 0, sizeof(double) * (size_t)K * pmc_stats_stride_c(D), st0);
        if (e0 == hipSuccess && d_vsums) e0 = hipMemsetAsync(d_vsums, 0, sizeof(double) * 2 * (size_t)K, st0);
        if (e0 != hipSuccess) return hipfail(e0, "hipMemsetAsync");
        return finish_scalars((const double *)d_workspace, 0, d_scalars, st0);
    }
    if (!fused_eligible(ks, K, kind, mode)) {
        if (!d_u) return fail(PMC_EINVAL, "pmc_estep: d_u is required unless pmc_estep_is_fused()");
        const bool groupable = ks->resp_groups && gemm_selected(ks, N, K, kind) &&
                               ((kind == PMC_KIND_VB && mode == PMC_RESP_VB) || (kind == PMC_KIND_GAUSS && mode == PMC_RESP_PMC_RB));
        // D >= 32: the forms of all components as one matrix product with the grouped epilogue fused behind it
        // (pmc_mgemm.hip); k_resp_groups then only does the workgroups its guard refused
        const int nct = groupable ? mgemm_pick(ks, N, K, true) : 0;
        if (groupable && (nct || resp_groups_pays(ks->dim, K))) {
            // the statistics will run as the component x monomial product: responsibilities in groups of one row
            // block, written once, their per-(sample, group) factors left to that kernel (k_resp_groups)
            if (!d_x) return fail(PMC_EINVAL, "pmc_estep: d_x is NULL");
            hipStream_t st = (hipStream_t)stream;
            const long long ntiles = ceil_div(N, PMC_TILE), nblocks = ceil_div(ntiles, PMC_A_WAVES);
            double *gscale = (double *)((char *)d_workspace + gscale_offset(N, K, ks));
            PmcArgsA a;
            std::memset(&a, 0, sizeof(a));
            a.x = d_x; a.N = N; a.dreal = D; a.pack = d_pack; a.K = K; a.max_init_zero = max_init_zero; a.mode = mode;
            a.ld = K; a.sample_w = d_sample_w; a.u = d_u; a.gscale = gscale; a.klds = PMC_RESP_GROUP;
            // (the scalar partials share the front of the workspace with the statistics' partial sums: finished first)
            a.partials = (double *)d_workspace;
            {
                Timed t(T_RESP, st, flops_pairs((double)N, K, D), 8.0 * N * (D + K + ceil_div(K, PMC_RESP_GROUP)));
                hipError_t e = hipSuccess;
                if (nct) e = mgemm_run(ks, nct, kind, a, a, d_workspace, st, false);
                if (e != hipSuccess) return hipfail(e, "k_mgemm launch");
                // the last round's groups of 16 components in pieces (split_plan): k_resp_groups' bits either way
                const SplitPlan sp = (!nct && ks->resp_groups_split) ? split_plan(ks, nblocks, (int)ceil_div(K, PMC_RESP_GROUP), 0, 1)
                                                                     : SplitPlan();
                if (sp.on) {
                    const int rc = split_apply(a, sp, d_workspace, K, ks, st);
                    if (rc != PMC_OK) return rc;
                    e = ks->resp_groups_split(kind, a, (unsigned)sp.grid, st);
                } else {
                    e = ks->resp_groups(kind, a, (unsigned)nblocks, st);
                }
                if (e != hipSuccess) return hipfail(e, "k_resp_groups launch");
            }
            int rc = finish_scalars((const double *)d_workspace, nblocks, d_scalars, st);
            if (rc != PMC_OK) return rc;
            return sufficient_stats_impl(d_x, N, D, d_spack, K, d_u, d_stats, d_workspace, stream, kind, gscale, d_pack);
        }
        // Small batches (round 6): a launch that does not fill the chip walks its groups of 16 components in pieces
        // (k_resp_groups_split) -- k_resp would walk all K on a handful of compute units, three passes each -- and the workgroup
        // that finishes a block multiplies the groups' factors into u itself: the per-component statistics kernel behind takes
        // a complete u.  VB and Gaussian Rao-Blackwell PMC, from two groups on.
        if (g_split && tun().small_grouped && ks->resp_groups_split && ks->padded != 2 && d_x && K > PMC_RESP_GROUP &&
            ((kind == PMC_KIND_VB && mode == PMC_RESP_VB) || (kind == PMC_KIND_GAUSS && mode == PMC_RESP_PMC_RB)) &&
            gscale_bytes(N, K, ks) > 0) {
            hipStream_t st = (hipStream_t)stream;
            const long long nblocks = ceil_div(ceil_div(N, PMC_TILE), PMC_A_WAVES);
            const SplitPlan sp = split_plan(ks, nblocks, (int)ceil_div(K, PMC_RESP_GROUP), 0, 1);
            // (where it pays: while the pieces have compute units to spread to -- K = 32, two groups: 56 -> 40 us at 40 blocks,
            //  62 -> 65 at 391; K = 128: 213 -> 146 at 196 blocks: profiles/r06_small_batch_estep.txt)
            const long long slots = (long long)device_cus() * split_slots_per_cu(ks->dim);
            if (sp.on && sp.b1 == 0 && nblocks * 16 <= slots * ceil_div(K, PMC_RESP_GROUP)) {
                PmcArgsA a;
                std::memset(&a, 0, sizeof(a));
                a.x = d_x; a.N = N; a.dreal = D; a.pack = d_pack; a.K = K; a.max_init_zero = max_init_zero; a.mode = mode;
                a.ld = K; a.sample_w = d_sample_w; a.u = d_u; a.klds = PMC_RESP_GROUP;
                a.gscale = (double *)((char *)d_workspace + gscale_offset(N, K, ks));     // (the groups' maxima wait there)
                a.partials = (double *)d_workspace;
                a.split_complete = 1;
                {
                    Timed t(T_RESP, st, flops_pairs((double)N, K, D), 8.0 * N * (D + 2 * K));
                    const int rc0 = split_apply(a, sp, d_workspace, K, ks, st);
                    if (rc0 != PMC_OK) return rc0;
                    const hipError_t e = ks->resp_groups_split(kind, a, (unsigned)sp.grid, st);
                    if (e != hipSuccess) return hipfail(e, "k_resp_groups launch");
                }
                const int rc1 = finish_scalars((const double *)d_workspace, nblocks, d_scalars, st);
                if (rc1 != PMC_OK) return rc1;
                return sufficient_stats_impl(d_x, N, D, d_spack, K, d_u, d_stats, d_workspace, stream, kind, nullptr, d_pack);
            }
        }
        int rc = pmc_responsibilities(d_x, N, D, d_pack, K, kind, mode, max_init_zero, d_sample_w, d_latent, d_u,
                                      d_scratch, d_vsums, nullptr, nullptr, nullptr, K, d_scalars, d_workspace, stream);
        if (rc != PMC_OK) return rc;
        return sufficient_stats_impl(d_x, N, D, d_spack, K, d_u, d_stats, d_workspace, stream, kind, nullptr, d_pack);
    }
    if (!d_x) return fail(PMC_EINVAL, "pmc_estep: d_x is NULL");
    hipStream_t st = (hipStream_t)stream;
    const FusedGeom g = fused_geom(N, K, ks->dim);
    const int PSc = pmc_stats_stride_c(ks->dim);
    PmcArgsF a;
    std::memset(&a, 0, sizeof(a));
    a.x = d_x; a.N = N; a.dreal = D; a.pack = d_pack; a.K = K; a.max_init_zero = max_init_zero;
    a.qs = g.qs; a.kq = g.kq; a.cw = g.cw; a.sample_w = d_sample_w; a.ntiles = g.ntiles; a.rounds_per_wg = g.rounds_per_wg; a.reg = g.reg;
    a.shift_pack = d_shift_pack;
    a.partials = (double *)d_workspace;
    a.spartials = a.partials + (size_t)g.nchunks * K * PSc;
    a.vpartials = a.spartials + (size_t)g.grid * PMC_NSCALARS;
    if (kind == PMC_KIND_STUDENT_T && !d_vsums) return fail(PMC_EINVAL, "pmc_estep: Student-t needs d_vsums");
    hipError_t e;
    {
        Timed t(T_FUSED, st, flops_pairs((double)N, K, D) + flops_stats((double)N, K, D), 8.0 * N * D);
        e = ks->fused(kind, g.qs, a, g.grid, st);
    }
    if (e != hipSuccess) return hipfail(e, "k_estep_fused launch");
    const long long total = (long long)K * pmc_stats_stride_c(D);
    Timed tf(T_FINISH, st, 0.0, 8.0 * g.nchunks * K * PSc);
    hipLaunchKernelGGL(k_finish_stats, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, st,
                       (const double *)a.partials, (int)g.nchunks, K, D, ks->dim, d_stats, (const int *)nullptr);
    e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "k_finish_stats launch");
    if (kind == PMC_KIND_STUDENT_T) {
        hipLaunchKernelGGL(k_finish_vsums, dim3((unsigned)(2 * K)), dim3(256), 0, st, (const double *)a.vpartials,
                           (long long)g.grid * g.tpr, K, d_vsums);
        e = hipGetLastError();
        if (e != hipSuccess) return hipfail(e, "k_finish_vsums launch");
    }
    return finish_scalars(a.spartials, g.grid, d_scalars, st);
}

int pmc_estep_from_tiles(const double *d_x, int64_t N, int D, const double *d_pack, int K, int kind,
                         int max_init_zero, const double *d_sample_w, const double *d_maha_tiles, int K_tiles,
                         double *d_u, double *d_vsums, double *d_stats, double *d_scalars, void *d_workspace,
                         void *stream)
{
    TuneScope options;                                     // one snapshot of the options for the whole call
    if (N < 0 || K < 1 || K_tiles < 1 || !d_pack || !d_stats || !d_scalars || !d_workspace)
        return fail(PMC_EINVAL, "pmc_estep_from_tiles: bad N/K/pack/stats/scalars/workspace");
    if (kind != PMC_KIND_GAUSS && kind != PMC_KIND_STUDENT_T)
        return fail(PMC_EINVAL, "pmc_estep_from_tiles: kind must be GAUSS or STUDENT_T (got %d)", kind);
    if (kind == PMC_KIND_STUDENT_T && !d_vsums) return fail(PMC_EINVAL, "pmc_estep_from_tiles: Student-t needs d_vsums");
    const PmcKernelSet *ks = kernels_for(D);
    if (!ks) return fail(PMC_EINVAL, "sample dimension %d is not supported (max %d)", D, PMC_BIG_MAX_DIM);
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        hipError_t e0 = hipMemsetAsync(d_stats, 0, sizeof(double) * (size_t)K * pmc_stats_stride_c(D), st);
        if (e0 == hipSuccess && d_vsums) e0 = hipMemsetAsync(d_vsums, 0, sizeof(double) * 2 * (size_t)K, st);
        if (e0 != hipSuccess) return hipfail(e0, "hipMemsetAsync");
        return finish_scalars((const double *)d_workspace, 0, d_scalars, st);
    }
    if (!d_x || !d_maha_tiles || !d_u) return fail(PMC_EINVAL, "pmc_estep_from_tiles: d_x / d_maha_tiles / d_u is NULL");
    const long long ntiles = ceil_div(N, PMC_TILE), nblocks = ceil_div(ntiles, PMC_A_WAVES);
    double *vpartials = (double *)((char *)d_workspace + scalar_partials_bytes(N));
    PmcArgsT a;
    std::memset(&a, 0, sizeof(a));
    a.mtile = d_maha_tiles; a.N = N; a.ld = K_tiles; a.dreal = D; a.pack = d_pack; a.K = K;
    a.stride = pmc_pack_stride_c(ks->dim); a.coff = ks->dim + pmc_tri(ks->dim);
    a.max_init_zero = max_init_zero; a.sample_w = d_sample_w; a.u = d_u;
    a.vpartials = kind == PMC_KIND_STUDENT_T ? vpartials : nullptr; a.partials = (double *)d_workspace;
    {
        Timed t(T_RESP, st, 50.0 * (double)N * K, 8.0 * N * 4 * K);
        hipError_t e = pmc_launch_resp_tiles(kind, a, (unsigned)nblocks, st);
        if (e != hipSuccess) return hipfail(e, "k_resp_tiles launch");
    }
    if (kind == PMC_KIND_STUDENT_T) {
        hipLaunchKernelGGL(k_finish_vsums, dim3((unsigned)(2 * K)), dim3(256), 0, st, (const double *)vpartials,
                           ntiles, K, d_vsums);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hipfail(e, "k_finish_vsums launch");
    }
    int rc = finish_scalars((const double *)d_workspace, nblocks, d_scalars, st);
    if (rc != PMC_OK) return rc;
    return sufficient_stats_impl(d_x, N, D, d_pack, K, d_u, d_stats, d_workspace, stream, kind);
}

}  // extern "C"
