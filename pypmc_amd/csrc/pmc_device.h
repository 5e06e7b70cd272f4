// pmc_device.h -- device helpers shared by the per-dimension kernel units.
#pragma once
#include "pmc_internal.h"
#include "../../include/pmc_hip.h"

#include <cfloat>
#include <type_traits>
#include <utility>

#ifndef PMC_D
#error "compile with -DPMC_D=<dimension>"
#endif
#ifndef PMC_PADDED
#define PMC_PADDED 0
#endif

namespace {

constexpr double TINY = 2.2250738585072014e-308;  // numpy.finfo('d').tiny

typedef __attribute__((address_space(4))) const double cdouble;        // scalar-cache loads
typedef __attribute__((address_space(4))) const long long cint64;

template <int I> using ic = std::integral_constant<int, I>;
// f(ic<B>{}), ..., f(ic<E - 1>{}) in this order, as ONE flat sequence of calls (a fold over the index pack).  Not the
// recursive form f(ic<B>{}); static_for<B + 1, E>(f): the inliner works bottom-up and copies the tail of the recursion once
// per level -- quadratic in E - B, ten minutes of compile time for the 545 steps of k_mgemm<64> (forty seconds this way;
// the code that comes out is the same).
template <int B, class F, int... I> __device__ __forceinline__ void static_for_pack(F &&f, std::integer_sequence<int, I...>)
{
    (f(ic<B + I>{}), ...);
}
template <int B, int E, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) static_for_pack<B>(f, std::make_integer_sequence<int, E - B>{});
}

// ---------------------------------------------------------------------------------------------
// sample row -> registers
// ---------------------------------------------------------------------------------------------
template <int D, bool PADDED>
__device__ __forceinline__ void load_row(const double *__restrict__ x, long long n, long long N,
                                         int dreal, double (&xv)[D])
{
    if (n < N) {
        if constexpr (!PADDED) {
            const double *p = x + n * D;
#pragma unroll
            for (int j = 0; j < D; ++j) xv[j] = p[j];
        } else {
            const double *p = x + n * (long long)dreal;
#pragma unroll
            for (int j = 0; j < D; ++j) xv[j] = j < dreal ? p[j] : 0.0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) xv[j] = 0.0;
    }
}

// Pull the 64-byte lines of one component's parameters into the scalar data cache and wait.
// The pack of a K >= 16 mixture does not fit the 16 KB scalar cache, so without this every
// s_load of the component loop would pay the L2 round trip behind its own s_waitcnt (70 % of the
// requests were "miss, fill pending" in the PMC counters); with it a wavefront pays one such
// round trip per component and its ~30 real loads hit.  Touch loads beyond the last line are
// clamped onto it.  One asm statement including the wait, so no load is in flight on exit.
template <int LINES, int FIRST> __device__ __forceinline__ void touch16(cdouble *p)
{
    constexpr int L = LINES - 1;
#define PMC_OFF(i) "n"(((FIRST + (i)) < L ? (FIRST + (i)) : L) * 64)
    int dummy;
    asm volatile("s_load_dword %0, %1, %2\n s_load_dword %0, %1, %3\n s_load_dword %0, %1, %4\n"
                 "s_load_dword %0, %1, %5\n s_load_dword %0, %1, %6\n s_load_dword %0, %1, %7\n"
                 "s_load_dword %0, %1, %8\n s_load_dword %0, %1, %9\n s_load_dword %0, %1, %10\n"
                 "s_load_dword %0, %1, %11\n s_load_dword %0, %1, %12\n s_load_dword %0, %1, %13\n"
                 "s_load_dword %0, %1, %14\n s_load_dword %0, %1, %15\n s_load_dword %0, %1, %16\n"
                 "s_load_dword %0, %1, %17\n s_waitcnt lgkmcnt(0)"
                 : "=&s"(dummy)
                 : "s"(p), PMC_OFF(0), PMC_OFF(1), PMC_OFF(2), PMC_OFF(3), PMC_OFF(4), PMC_OFF(5),
                   PMC_OFF(6), PMC_OFF(7), PMC_OFF(8), PMC_OFF(9), PMC_OFF(10), PMC_OFF(11),
                   PMC_OFF(12), PMC_OFF(13), PMC_OFF(14), PMC_OFF(15)
                 : "memory");
#undef PMC_OFF
}
template <int D> __device__ __forceinline__ void touch_component(cdouble *pk)
{
    constexpr int LINES = pmc_pack_stride_c(D) * 8 / 64;
#ifdef PMC_TOUCH_BLOCKS
    static_for<0, (LINES + 15) / 16>([&](auto B) { touch16<LINES, decltype(B)::value * 16>(pk); });
#else
    // exactly LINES loads and ONE wait (the assembler repeats the line; blocks of 16 with a wait each cost a
    // scalar round trip per block -- two at D = 16 / 20, four at D = 30 -- and up to 12 redundant loads)
    int dummy;
    asm volatile(".set pmc_touch_i, 0\n .rept %2\n s_load_dword %0, %1, pmc_touch_i*64\n .set pmc_touch_i, pmc_touch_i+1\n .endr\n"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(dummy)
                 : "s"(pk), "n"(LINES)
                 : "memory");
#endif
}

// Workgroup barrier behind LDS-DMA (global_load_lds): every wavefront first waits for ITS OWN pieces to have
// landed (vmcnt), then the barrier orders them against the other wavefronts' reads.  The wait is explicit:
// __syncthreads() alone does not always get one from the compiler -- in the component loop of the D >= 32
// per-sample kernels the s_barrier came out bare and a component was occasionally (1 run in ~10, shortest
// with a 1-4 component target mixture) read before its copy had arrived.
__device__ __forceinline__ void dma_barrier()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// maha = |R (x - mu)|^2 ; R upper triangular, packed row-major in consumption order.
// Replaces bilinear_sym(inv_sigma, x - mu) (pypmc/tools/_linalg.pyx:10-39).
template <int D> __device__ __forceinline__ double mahalanobis(const double (&xv)[D], cdouble *pk)
{
    double d[D];
#pragma unroll
    for (int j = 0; j < D; ++j) d[j] = xv[j] - pk[j];
    double maha = 0.0;
    int idx = D;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double y = 0.0;
#pragma unroll
        for (int j = i; j < D; ++j) y = fma(pk[idx++], d[j], y);
        maha = fma(y, y, maha);
    }
    return maha;
}

// The same product with the scalar loads scheduled by hand (-DPMC_SP_GROUP=G, G = 1 or 2 lines of 8 coefficients per
// group): a group's s_load_dwordx16 are issued one group ahead of their use into the other half of 2 G SGPR buffers.
// Left to itself the compiler gives k_logpdf's first loop two or three buffers, but k_resp_groups' (and the target
// mixture's) loop ONE: load, s_waitcnt, eight multiply-adds, load, ... -- every scalar-cache hit's latency in the open,
// 29 times per component.  Scalar loads return out of order, so lgkmcnt(0) is the only wait there is and the distance
// between issue and wait is what one group computes.  Same operations in the same order: same bits.
typedef double sgpr8d __attribute__((ext_vector_type(8)));
template <int BYTES> __device__ __forceinline__ void sp_issue(sgpr8d &r, cdouble *p)
{
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(r) : "s"(p), "n"(BYTES));
}
__device__ __forceinline__ void sp_wait(sgpr8d &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
__device__ __forceinline__ void sp_wait(sgpr8d &a, sgpr8d &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }
// `touch`: pull the component's lines into the scalar cache first (touch_component above) -- here with the first
// group's loads already on their way and ONE wait for both (the register the touch loads land in stays reserved up to
// that wait).
template <int D, int G> __device__ __forceinline__ double mahalanobis_sp(const double (&xv)[D], cdouble *pk, bool touch)
{
    static_assert(G == 1 || G == 2, "one or two lines per group");
    constexpr int T = D + D * (D + 1) / 2, NC = (T + 7) / 8, NG = (NC + G - 1) / G;
    sgpr8d b[2][G];
    auto issue = [&](auto GI) {
        constexpr int g = decltype(GI)::value;
        static_for<0, G>([&](auto E) {
            constexpr int e = decltype(E)::value, c = g * G + e;
            if constexpr (c < NC) sp_issue<c * 64>(b[g & 1][e], pk);
        });
    };
    int landing = 0;
    auto arrive = [&](auto GI) {
        constexpr int g = decltype(GI)::value;
        if constexpr (g == 0) {
            if constexpr (G == 2 && 1 < NC)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0][0]), "+s"(b[0][1]), "+s"(landing));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0][0]), "+s"(landing));
        } else if constexpr (G == 2 && g * G + 1 < NC) sp_wait(b[g & 1][0], b[g & 1][1]);
        else sp_wait(b[g & 1][0]);
    };
    // coefficient number idx of the component (compile-time): waits for its group at the group's first element and
    // sends the next group on its way
    auto coef = [&](auto IDX) -> double {
        constexpr int idx = decltype(IDX)::value, c = idx / 8, g = c / G;
        if constexpr (idx % (8 * G) == 0) {
            // (nothing crosses: the asm statements have no latency the scheduler knows of, and it would let the
            //  arithmetic of a group drift behind the next wait)
            __builtin_amdgcn_sched_barrier(0);
            arrive(std::integral_constant<int, g>{});
            if constexpr (g + 1 < NG) issue(std::integral_constant<int, g + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        return b[g & 1][c % G][idx % 8];
    };
    issue(std::integral_constant<int, 0>{});
    if (touch) {                                          // workgroup-uniform
        constexpr int LINES = pmc_pack_stride_c(D) * 8 / 64;
        asm volatile(".set pmc_touch_i, 0\n .rept %2\n s_load_dword %0, %1, pmc_touch_i*64\n .set pmc_touch_i, pmc_touch_i+1\n .endr"
                     : "+s"(landing) : "s"(pk), "n"(LINES) : "memory");
    }
#ifdef PMC_AB_BIAS
    // A/B switch, TIMING ONLY (wrong numbers): the instruction stream of the "bias" form |R x' - b|^2 with b = R (mu - c)
    // in the place of mu and x' = x - c formed once per sample -- no subtraction per pair, no d[] registers.  What that
    // form could gain at best.
    // (the stream read in order: row i = its bias, then its D - i coefficients -- the layout a bias pack would have)
    double maha = 0.0;
    static_for<0, D>([&](auto I) {
        constexpr int i = decltype(I)::value, row = i * (D + 1) - i * (i - 1) / 2;
        double y = -coef(std::integral_constant<int, row>{});
        static_for<i, D>([&](auto J) {
            constexpr int j = decltype(J)::value;
            y = fma(coef(std::integral_constant<int, row + 1 + j - i>{}), xv[j], y);
        });
        maha = fma(y, y, maha);
    });
    return maha;
#else
    double d[D];
    static_for<0, D>([&](auto J) { constexpr int j = decltype(J)::value; d[j] = xv[j] - coef(J); });
    double maha = 0.0;
    static_for<0, D>([&](auto I) {
        constexpr int i = decltype(I)::value, base = D + i * D - i * (i - 1) / 2 - i;
        double y = 0.0;
        static_for<i, D>([&](auto J) {
            constexpr int j = decltype(J)::value;
            y = fma(coef(std::integral_constant<int, base + j>{}), d[j], y);
        });
        maha = fma(y, y, maha);
    });
    return maha;
#endif
}

// Two samples per lane (round 6 A/B, verdict r5 #3): the same coefficient stream, every coefficient feeding TWO multiply-adds --
// half the scalar loads and waits per pair, twice the registers per lane (two wavefronts per SIMD instead of four).
template <int D, int G> __device__ __forceinline__ void mahalanobis_sp2(const double (&xa)[D], const double (&xb)[D], cdouble *pk,
                                                                        bool touch, double &maha_a, double &maha_b)
{
    static_assert(G == 1 || G == 2, "one or two lines per group");
    constexpr int T = D + D * (D + 1) / 2, NC = (T + 7) / 8, NG = (NC + G - 1) / G;
    sgpr8d b[2][G];
    auto issue = [&](auto GI) {
        constexpr int g = decltype(GI)::value;
        static_for<0, G>([&](auto E) {
            constexpr int e = decltype(E)::value, c = g * G + e;
            if constexpr (c < NC) sp_issue<c * 64>(b[g & 1][e], pk);
        });
    };
    int landing = 0;
    auto arrive = [&](auto GI) {
        constexpr int g = decltype(GI)::value;
        if constexpr (g == 0) {
            if constexpr (G == 2 && 1 < NC)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0][0]), "+s"(b[0][1]), "+s"(landing));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0][0]), "+s"(landing));
        } else if constexpr (G == 2 && g * G + 1 < NC) sp_wait(b[g & 1][0], b[g & 1][1]);
        else sp_wait(b[g & 1][0]);
    };
    auto coef = [&](auto IDX) -> double {
        constexpr int idx = decltype(IDX)::value, c = idx / 8, g = c / G;
        if constexpr (idx % (8 * G) == 0) {
            __builtin_amdgcn_sched_barrier(0);
            arrive(std::integral_constant<int, g>{});
            if constexpr (g + 1 < NG) issue(std::integral_constant<int, g + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        return b[g & 1][c % G][idx % 8];
    };
    issue(std::integral_constant<int, 0>{});
    if (touch) {                                          // workgroup-uniform
        constexpr int LINES = pmc_pack_stride_c(D) * 8 / 64;
        asm volatile(".set pmc_touch_i, 0\n .rept %2\n s_load_dword %0, %1, pmc_touch_i*64\n .set pmc_touch_i, pmc_touch_i+1\n .endr"
                     : "+s"(landing) : "s"(pk), "n"(LINES) : "memory");
    }
    double da[D], db[D];
    static_for<0, D>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const double m = coef(J);
        da[j] = xa[j] - m;
        db[j] = xb[j] - m;
    });
    maha_a = 0.0;
    maha_b = 0.0;
    static_for<0, D>([&](auto I) {
        constexpr int i = decltype(I)::value, base = D + i * D - i * (i - 1) / 2 - i;
        double ya = 0.0, yb = 0.0;
        static_for<i, D>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const double c = coef(std::integral_constant<int, base + j>{});
            ya = fma(c, da[j], ya);
            yb = fma(c, db[j], yb);
        });
        maha_a = fma(ya, ya, maha_a);
        maha_b = fma(yb, yb, maha_b);
    });
}

// acc += coef[lane N of this lane's 16-lane row] * d  -- v_fmac_f64 with a DPP row broadcast on its first
// source: a coefficient held ONCE per row in a VGPR feeds the FMA of all 64 lanes like an SGPR operand
// would, at the full fp64 rate (scripts/microbench/dpp_f64.hip: 73 TFLOP/s against 68 with plain VGPR
// operands).  The broadcast operand must come from a vector load (no VALU -> DPP hazard to cover).
template <int N> __device__ __forceinline__ void fmac_bcast(double &acc, double coef, double d)
{
    static_assert(N >= 0 && N < 16, "lane within a row");
    // volatile + memory: the statements keep their order -- the order of the coefficients in the window --
    // and the coefficient loads written between them stay between them in the instruction selector
    // (MahaEngine's scheduling barriers do the same for the machine scheduler)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(coef), "v"(d), "n"(N) : "memory");
}

// max(a, b) as ONE v_max_f64: fmax() costs a second instruction (the compiler canonicalises an operand first),
// "a > b ? a : b" a compare and two v_cndmask -- 1.6 and 6.8 issue slots against 0.8 on this chip
// (scripts/microbench/fp64_oprates.hip).  A NaN operand loses, as it does against the reference's ">".
__device__ __forceinline__ double max_f64(double a, double b)
{
#ifdef PMC_LIBM_LSE            // A/B switch (scripts/tune_unit.sh): the compiler's and the library's forms
    return a > b ? a : b;
#else
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}

// r == 0 ? tiny : r for r >= +0 (variational.pyx:751-753) on the integer pipe: tiny = 0x0010000000000000 has a
// zero low word, so only the high word is selected -- v_or, v_cmp_eq_u32, v_cndmask: 1.7 issue slots against 6.8
// for the fp64 compare and its two selects.
__device__ __forceinline__ double zero_to_tiny(double r)
{
#ifdef PMC_LIBM_LSE
    return r == 0.0 ? TINY : r;
#else
    const unsigned lo = (unsigned)__double2loint(r), hi = (unsigned)__double2hiint(r);
    const unsigned h2 = ((lo | hi) == 0u) ? 0x00100000u : hi;
    return __hiloint2double((int)h2, (int)lo);
#endif
}

// log(t) for a positive, finite, normal t (1 + maha / nu, (maha + nu) / 2): the classic argument reduction
// t = 2^k m, m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), log m = f - f^2/2 + s (f^2/2 + R(s^2)) with the
// degree-7 minimax R of the freely distributable fdlibm e_log.c (error < 1 ulp), the division as a reciprocal with
// two Newton steps and one residual correction.  38 vector instructions; the device library's log is a
// double-double evaluation of 88 (most of what a Student-t pair costs beyond a Gaussian one: D = 8 +75 %, D = 20
// +28 % before).  t = +inf (a Mahalanobis form beyond 1e308: the reference's log gives +inf and the component's value
// -inf, student_t.pyx:159-164) falls through the reduction as NaN and is put right behind it: one v_cmp_class per call,
// the select only in a wavefront that holds such a lane.  NaN stays NaN.
__device__ __forceinline__ double log_pos_finite(double t);
__device__ __forceinline__ double log_pos(double t)
{
    double r = log_pos_finite(t);
    const bool pinf = __builtin_amdgcn_class(t, 0x200);                 // +infinity
    if (__builtin_expect(__any(pinf), 0)) r = pinf ? t : r;
    return r;
}
__device__ __forceinline__ double log_pos_finite(double t)
{
#ifdef PMC_LIBM_LOG
    return log(t);
#else
    double m;
    int e;
    asm("v_frexp_mant_f64 %0, %1" : "=v"(m) : "v"(t));              // m in [0.5, 1)
    asm("v_frexp_exp_i32_f64 %0, %1" : "=v"(e) : "v"(t));
    const int adj = ((unsigned)__double2hiint(m) < 0x3fe6a09eu) ? 1 : 0;     // m < sqrt(1/2): use 2 m and k - 1
    m = ldexp(m, adj);
    const double k = (double)(e - adj);
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r;
    asm("v_rcp_f64 %0, %1" : "=v"(r) : "v"(d));
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double sq = f * r;
    sq = fma(fma(-d, sq, f), r, sq);                                  // s = f / (2 + f)
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
    return fma(k, 6.93147180369123816490e-01, -((hfsq - fma(sq, hfsq + R, k * 1.90821492927058770002e-10)) - f));
#endif
}

// log of a per-sample quantity (a row sum, a normalisation): the lean form when every lane of the wavefront holds a
// positive normal number -- the rule; the library's otherwise (a row sum that underflowed to 0 must give -inf).
__device__ __forceinline__ double log_any(double t)
{
    if (__all(t >= 2.2250738585072014e-308 && t <= DBL_MAX)) return log_pos_finite(t);      // (a NaN fails the test)
    return log(t);
}

// Rows whose result must be NaN: a component value that is NaN or +inf (the reference's exp(a - max) of such a row is NaN,
// _regularize.pyx:72-81, variational.pyx:728-755).  -inf -- a Mahalanobis form beyond 1e308 -- is NOT one: exp(-inf) = 0,
// the component drops out and the others carry the row, as in the reference (student_t.pyx:159-164, gauss.pyx:151).
// exp_clamped turns a NaN argument into e = 0, so the NaN has to travel separately: one v_cmp_class per pair (in the place
// of the multiply-add 0 * v + poison of rounds 1-4, which also poisoned -inf), its lane mask OR-ed on the scalar unit.
struct RowPoison {
    unsigned long long bad = 0;
    __device__ __forceinline__ void see(double v) { bad |= __ballot(__builtin_amdgcn_class(v, 0x203)); }   // sNaN | qNaN | +inf
    __device__ __forceinline__ double value() const
    {
        return ((bad >> (threadIdx.x & 63)) & 1ull) ? __longlong_as_double(0x7ff8000000000000LL) : 0.0;
    }
};

// a_nk from maha_nk, in the reference's operation order (see enum pmc_kind).
template <int D, int KIND>
__device__ __forceinline__ double component_value(double maha, cdouble *c, double &expo)
{
    if constexpr (KIND == PMC_KIND_GAUSS) {
        return fma(-0.5, maha, c[0]);                   // gauss.pyx:151  c0 - 0.5 maha: the halving is exact, so
                                                        // one fused instruction rounds exactly as the two do
    } else if constexpr (KIND == PMC_KIND_STUDENT_T) {
        double t = maha;                                  // student_t.pyx:159-164
        t *= c[2];
        t += 1.;
        t = log_pos(t);
        t *= c[1];
        t += c[0];
        return t;
    } else {
        expo = c[0] + c[1] * maha;                        // variational.pyx:798
        return fma(0.5, c[3] - expo, c[2]);               // variational.pyx:691  (exact halving: same bits, one instruction)
    }
}

// exp(x) for x <= 0: the device library's algorithm and constants (k = rint(x log2 e), r = x - k ln 2 in two
// pieces, degree-11 polynomial, ldexp) without its overflow branch, with the underflow branch replaced by a
// clamp of the argument (ldexp rounds the subnormal results, -1075 and below give 0) and the rounding done
// with a shifter constant -- within an ulp of exp(); 16 instead of 24 vector instructions.  NaN arguments give
// 0, not NaN: callers poison the sample weight instead (below).
struct ExpConst {
    double log2e, nln2hi, nln2lo, c[9];
    __device__ __forceinline__ ExpConst()
    {
        log2e = __longlong_as_double(0x3ff71547652b82feLL);
        nln2hi = __longlong_as_double(0xbfe62e42fefa39efLL);
        nln2lo = __longlong_as_double(0xbc7abc9e3b39803fLL);
        c[0] = __longlong_as_double(0x3e5ade156a5dcb37LL);
        c[1] = __longlong_as_double(0x3e928af3fca7ab0cLL);
        c[2] = __longlong_as_double(0x3ec71dee623fde64LL);
        c[3] = __longlong_as_double(0x3efa01997c89e6b0LL);
        c[4] = __longlong_as_double(0x3f2a01a014761f6eLL);
        c[5] = __longlong_as_double(0x3f56c16c1852b7b0LL);
        c[6] = __longlong_as_double(0x3f81111111122322LL);
        c[7] = __longlong_as_double(0x3fa55555555502a1LL);
        c[8] = __longlong_as_double(0x3fc5555555555511LL);
    }
};
// (exp_clamped: the argument is already max_f64(x, -1075.0) -- callers that need the clamped value themselves)
__device__ __forceinline__ double exp_clamped(double xc, const ExpConst &E)
{
#ifdef PMC_LIBM_LSE
    return exp(xc);
#endif
    // k = rint(xc log2 e) by the shifter constant 1.5 * 2^52: one fused multiply-add leaves the rounded value in the
    // low mantissa bits -- as an integer in the low word (|k| < 2^31) and, after subtracting the shifter again, as
    // a double.  (2 instructions instead of multiply, v_rndne, v_cvt_i32; a product within half an ulp of a tie
    // may round to the other neighbour than the library's two-step rounding: r then ends up 1e-16 outside
    // [-ln2/2, ln2/2], which the polynomial does not notice.)
    const double shifted = fma(xc, E.log2e, 6755399441055744.0);
    const double k = shifted - 6755399441055744.0;
    double r = fma(k, E.nln2hi, xc);
    r = fma(k, E.nln2lo, r);
    double p = fma(E.c[0], r, E.c[1]);
#pragma unroll
    for (int i = 2; i < 9; ++i) p = fma(r, p, E.c[i]);
    p = fma(r, p, 0.5);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(p, __double2loint(shifted));
}
__device__ __forceinline__ double exp_le0(double x, const ExpConst &E) { return exp_clamped(max_f64(x, -1075.0), E); }


// v_bfi_b32: (mask & a) | (~mask & b).  Inline asm because the compiler turns the C expression back into the
// compare + v_cndmask form it was written to avoid (one compare feeding four selects: ~7 issue slots against 2,
// scripts/microbench/fp64_oprates.hip).
__device__ __forceinline__ unsigned bfi_vv(unsigned mask, unsigned a, unsigned b)
{
    unsigned r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}

// One step of the streaming log-sum-exp  log sum_k w_k exp(a_k) = m + log s  with
// m = running max, s = sum_k w_k exp(a_k - m)   (one exp per step):
//     a <= m:  s += w exp(a - m)             a > m:  s = s exp(m - a) + w,  m = a.
// Both are formed with e = exp(-|a - m|) and the sign word of a - m picks one on the
// integer pipe (a - m = +0 or a positive value with a zero high word count as "a > m": e = 1, both forms agree).
// ``w`` is the component's weight, uniform over the wavefront.  A NaN a_k adds nothing here (the clamp turns
// exp's argument into -1075); k_logpdf adds the row's poison (0 or NaN) to the result instead.
__device__ __forceinline__ void lse_step(double a, double w, double &m, double &s, const ExpConst &E)
{
#ifdef PMC_LIBM_LSE
    const double e = exp(-fabs(a - m));
    const bool gt = a > m;
    s = gt ? fma(s, e, w) : fma(w, e, s);
    m = gt ? a : m;
#else
    const double d = a - m;
    double arg;                                                      // max(-|d|, -1075): source modifiers are free
    asm("v_max_f64 %0, -|%1|, %2" : "=v"(arg) : "v"(d), "v"(-1075.0));
    const double e = exp_clamped(arg, E);
    const unsigned below = (unsigned)(__double2hiint(d) >> 31);      // all ones: a < m
    // both updates, then the sign word of a - m picks one: two multiply-adds and two v_bfi (selecting the operands
    // first would be four v_bfi and one multiply-add)
    const double up = fma(s, e, w), stay = fma(w, e, s);
    s = __hiloint2double((int)bfi_vv(below, (unsigned)__double2hiint(stay), (unsigned)__double2hiint(up)),
                         (int)bfi_vv(below, (unsigned)__double2loint(stay), (unsigned)__double2loint(up)));
    m = max_f64(a, m);
#endif
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block (PMC_A_WAVES wavefronts) reduction of NS per-lane scalars -> partials[block*PMC_NSCALARS+i]
template <int NS>
__device__ __forceinline__ void block_scalars(double (&sc)[NS], double *partials)
{
    __shared__ double red[PMC_A_WAVES][PMC_NSCALARS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double v = wave_sum(sc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < PMC_NSCALARS) {
        double v = 0.0;
        if (threadIdx.x < NS) {
#pragma unroll
            for (int w = 0; w < PMC_A_WAVES; ++w) v += red[w][threadIdx.x];
        }
        partials[(size_t)blockIdx.x * PMC_NSCALARS + threadIdx.x] = v;
    }
}
// ... of sample block `blk` (the split kernels: the workgroup that finishes a block is not workgroup `blk` of the launch)
template <int NS>
__device__ __forceinline__ void block_scalars_at(double (&sc)[NS], double *partials, long long blk)
{
    __shared__ double red[PMC_A_WAVES][PMC_NSCALARS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double v = wave_sum(sc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < PMC_NSCALARS) {
        double v = 0.0;
        if (threadIdx.x < NS) {
#pragma unroll
            for (int w = 0; w < PMC_A_WAVES; ++w) v += red[w][threadIdx.x];
        }
        partials[(size_t)blk * PMC_NSCALARS + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Pieces of a sample block (k_logpdf_split / k_resp_groups_split): what a piece leaves for the workgroup that finishes
// the block crosses compute units -- and XCDs, each with an L2 of its own.  It is written and read with AGENT-scope atomic
// accesses (relaxed: global_store / global_load with sc1, written through to / read from the level all XCDs share), and the
// block's ticket -- an agent-scope atomic too -- orders them: every thread waits until its stores are acknowledged
// (s_waitcnt vmcnt(0)), barrier, one thread draws the ticket, barrier, and only the workgroup that drew the last one goes
// on to read.  No agent-scope FENCES: a release fence writes the XCD's whole L2 back (buffer_wbl2) and an acquire fence
// invalidates it (buffer_inv sc1) -- once per piece, that evicted the parameter pack and the samples under every other
// workgroup of the XCD: the first build of this was 2-3x SLOWER than one workgroup per block from 16384 samples on
// (profiles/r06_split_fences.txt).  Nothing else a piece writes is read by the finishing workgroup.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void piece_store(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double piece_load(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true in every thread of the workgroup that drew the LAST of the block's `pieces` tickets (the counter wraps to 0 there)
__device__ __forceinline__ bool piece_ticket_is_last(unsigned *counter, int pieces)
{
    __shared__ unsigned ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's piece_stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0)
        ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)(pieces - 1)) return false;
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    asm volatile("" ::: "memory");
    return true;
}

// store of a responsibility value: written once, read by the NEXT kernel -- far more than the L2 holds in between
// (-DPMC_NT_STORES: with the non-temporal hint, an A/B switch)
__device__ __forceinline__ void store_u(double *p, double v)
{
#ifdef PMC_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

constexpr int D_ = PMC_D;
constexpr bool P_ = PMC_PADDED != 0;

}  // namespace

#define PMC_CAT4(a, b, c, d) a##b##c##d
#define PMC_UNIT_NAME(stem, d, p) PMC_CAT4(stem, d, _p, p)
#define PMC_UNIT_NAME_X(stem, d, p) PMC_UNIT_NAME(stem, d, p)
