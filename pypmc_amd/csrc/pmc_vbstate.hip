// pmc_vbstate.hip -- the K-sized half of a variational-Bayes iteration as kernels (round 6, verdict r5 #6).
//
// With the E-step's big kernels at 1.4 ms for one GPU's share of eight, what the host did between two of them -- the
// M-step's K inversions (variational.pyx:129-136, :693-697, :934-946), the digamma sums of the next E-step's expectations
// (:759-772, :800-804), the seven terms of the bound (:948-1034) and the two copies that carried K x D x D matrices to the
// host and back -- was an eighth of an iteration.  Here the posterior's hyper-parameters and the latest statistics stay on
// the device: `pmc_vb_mstep_device` -> `pmc_vb_expectations_device` -> (the pack builder and the E-step's kernels, as
// before) -> `pmc_vb_after_device` -> `pmc_vb_bound_device`; what returns per iteration is one block of 8 K + 16 doubles.
//
// Numbers: each component's W_k^-1 is factorised and inverted here by the algorithm LAPACK's dpotrf / dpotri use
// (Cholesky, inverse of the factor, its Gram matrix), not by LAPACK: results agree with the host path to rounding
// (relative 1e-13 on well-conditioned matrices), not bit for bit; psi and ln Gamma are the recurrence + asymptotic series
// below (|error| <= 2e-15 (1 + |psi|), 1e-14 (1 + |ln Gamma|): tests/test_cabi.py holds them to scipy's).  One wavefront per
// component, the matrix in LDS: K-sized work is latency, not throughput.
#include "../../include/pmc_hip.h"
#include "pmc_convert.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

extern "C" int pmc_internal_fail(int code, const char *msg);       // pmc_api.hip: sets pmc_last_error()

namespace {

constexpr double VB_TINY = 2.2250738585072014e-308;                 // numpy.finfo('d').tiny (_regularize.pyx:6-17)
constexpr double VB_LN2 = 0.69314718055994530942;
constexpr double VB_LNPI = 1.14472988584940017414;
// log(2 * pi) as the host computes it -- the logarithm of the DOUBLE 2 * pi, one ulp below the rounded ln 2 pi; the pack's
// c3 = E[ln|Lambda|] - D ln 2 pi must be the host path's bit for bit (pmc_ctx.hip::pmc_vb_estep, variational.py)
constexpr double VB_LN2PI = 0x1.d67f1c864beb4p+0;
constexpr int VB_NTERMS = 10;                                       // per-component summands of the bound
constexpr int PMC_NSCALARS = 8;                                     // (pmc_internal.h: the scalar sums in front of the statistics)

// psi(x), x > 0: psi(x) = psi(x + 1) - 1 / x up to x >= 10, then ln x - 1 / 2x - sum B_2n / (2n x^2n)
__host__ __device__ inline double vb_digamma(double x)
{
    if (!(x > 0.0)) return NAN;
    if (x > 1e300) return log(x);
    double r = 0.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0 +
                     f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + (log(x) - 0.5 / x + t);
}

// ln Gamma(x), x > 0: ln Gamma(x) = ln Gamma(x + n) - ln(x (x + 1) ... (x + n - 1)) up to x >= 10, then Stirling's series
__host__ __device__ inline double vb_lgamma(double x)
{
    if (!(x > 0.0)) return NAN;
    if (x > 1e300) return INFINITY;
    double p = 1.0;
    while (x < 10.0) {
        p *= x;
        x += 1.0;
    }
    const double i = 1.0 / x, f = i * i;
    const double t = i * (1.0 / 12.0 + f * (-1.0 / 360.0 + f * (1.0 / 1260.0 + f * (-1.0 / 1680.0 + f * (1.0 / 1188.0 +
                     f * (-691.0 / 360360.0 + f * (1.0 / 156.0)))))));
    return ((x - 0.5) * log(x) - x + 0.5 * VB_LN2PI + t) - log(p);
}

__device__ inline double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}

// s - sum_{l < n} a[l * sa] * b[l * sb], subtracted in order (the plain loop's bits) with the loads issued eight at a time:
// one wavefront per component is latency from end to end, and a loop that waits for two LDS reads per term spends
// 128 cycles on each of them
__device__ inline double dot_sub(double s, const double *a, int sa, const double *b, int sb, int n)
{
    int l = 0;
    for (; l + 8 <= n; l += 8) {
        double x[8], y[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            x[q] = a[(l + q) * sa];
            y[q] = b[(l + q) * sb];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s -= x[q] * y[q];
    }
    for (; l < n; ++l) s -= a[l * sa] * b[l * sb];
    return s;
}
__device__ inline double dot_add(double s, const double *a, int sa, const double *b, int sb, int n)
{
    int l = 0;
    for (; l + 8 <= n; l += 8) {
        double x[8], y[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            x[q] = a[(l + q) * sa];
            y[q] = b[(l + q) * sb];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s += x[q] * y[q];
    }
    for (; l < n; ++l) s += a[l * sa] * b[l * sb];
    return s;
}

// ---- M-step (variational.pyx:129-136) -----------------------------------------------------------------------
// One wavefront per component.  nu, alpha, beta (10.58, 10.60, 10.63); m (10.61); W^-1 (10.62) in the operation order of
// pypmc_amd/mix_adapt/variational.py::M_step; W = inv(W^-1) and ln|W| through the Cholesky factor.
// status = [1 + failing pivot, or 0 (K) | its value (K)]: every component writes its two slots.
__global__ __launch_bounds__(64) void k_vb_mstep(pmc_vb_fields f, int K, int D, double *status)
{
    extern __shared__ double lds[];
    const int LD = D + 1;                                           // (odd row stride: a column is conflict-free)
    double *A = lds, *X = lds + (size_t)D * LD;
    const int k = blockIdx.x, t = threadIdx.x;
    double n = f.N_comp[k];
    n = n == 0.0 ? VB_TINY : n;
    const double b0 = f.beta0[k], beta = b0 + n;
    if (t == 0) {
        f.nu[k] = f.nu0[k] + n;
        f.alpha[k] = f.alpha0[k] + n;
        f.beta[k] = beta;
    }
    double dx = 0.0;
    if (t < D) {
        const double xb = f.x_mean[(size_t)k * D + t], m0 = f.m0[(size_t)k * D + t];
        f.m[(size_t)k * D + t] = (b0 * m0 + n * xb) / beta;
        dx = xb - m0;
    }
    const double fac = b0 / (b0 + n);
    const double *S = f.S + (size_t)k * D * D, *iW0 = f.inv_W0 + (size_t)k * D * D;
#pragma unroll 4
    for (int i = 0; i < D; ++i) {
        const double dxi = __shfl(dx, i, 64);
        if (t < D) {
            double v = (dxi * dx) * fac;
            v += S[i * D + t];
            v *= n;
            v += iW0[i * D + t];
            A[i * LD + t] = v;
        }
    }
    __syncthreads();
    // W^-1 = L L^T, in place in the lower triangle: lane = row
    int bad = -1;
    double badv = 0.0, logdet = 0.0;
#ifdef VB_AB_NO_CHOL
    for (int c = 0; c < 0; ++c) {
#else
    for (int c = 0; c < D; ++c) {
#endif
        double s = 0.0;
        if (t >= c && t < D) {
            s = dot_sub(A[t * LD + c], A + t * LD, 1, A + c * LD, 1, c);
        }
        const double piv = __shfl(s, c, 64);
        if (!(piv > 0.0) || !isfinite(piv)) {
            bad = c;
            badv = piv;
            break;                                                  // (wave-uniform)
        }
        const double lcc = sqrt(piv);
        if (t == c) A[c * LD + c] = lcc;                            // (column c's old entries were read by their own lanes only)
        else if (t > c && t < D) A[t * LD + c] = s / lcc;
        __syncthreads();
    }
    if (bad < 0) {
        // ln|W^-1| / 2 = sum_c ln L_cc: the logarithms side by side (they are off the factorisation's critical path), added
        // left to right as the host adds them; the reciprocals of the diagonal for the inverse below
        const double lg = t < D ? log(A[t * LD + t]) : 0.0;
        for (int c = 0; c < D; ++c) logdet += __shfl(lg, c, 64);
    }
    if (t == 0) {
        status[k] = (double)(bad + 1);
        status[K + k] = badv;
    }
    double *W = f.W + (size_t)k * D * D;
    if (bad >= 0) {
        for (int idx = t; idx < D * D; idx += 64) W[idx] = NAN;
        if (t == 0) f.log_det_W[k] = NAN;
        return;
    }
    // X = L^-1 (lower): lane = column
#ifdef VB_AB_NO_INV
    if (t < 0) {
#else
    if (t < D) {
#endif
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
            if (i == t) {
                v = 1.0 / A[i * LD + i];
            } else if (i > t) {
                v = dot_sub(0.0, A + i * LD + t, 1, X + t * LD + t, LD, i - t) / A[i * LD + i];
            }
            X[i * LD + t] = v;
        }
    }
    __syncthreads();
    // W = X^T X: lane = column; the leading terms of the shorter of the two sums are exact zeros, so W is symmetric bit for bit
#ifdef VB_AB_NO_GRAM
    if (t < 0) {
#else
    if (t < D) {
#endif
        for (int i = 0; i < D; ++i) {
            W[i * D + t] = dot_add(0.0, X + i * LD + i, LD, X + i * LD + t, LD, D - i);
        }
    }
    if (t == 0) f.log_det_W[k] = -(2.0 * logdet);
}

// ---- chol_inv_det of K matrices (pypmc/tools/_linalg.pyx:41-95: lower factor, inverse, ln det) -----------------------
// The PMC update's K factorisations (pmc.pyx:227-244 through Gauss.update, gauss.pyx:46-57): one wavefront per matrix,
// the stages of k_vb_mstep.  out_k = [L (D x D, zeros above the diagonal) | A^-1 (D x D, symmetric bit for bit) | ln det A |
// 1 + failing pivot or 0 | its value]: everything a caller reads, one block per matrix for one copy.
__global__ __launch_bounds__(64) void k_spd_inverse(const double *A_all, int K, int D, double *out)
{
    extern __shared__ double lds[];
    const int LD = D + 1;
    double *A = lds, *X = lds + (size_t)D * LD;
    const int k = blockIdx.x, t = threadIdx.x;
    const double *src = A_all + (size_t)k * D * D;
    double *o = out + (size_t)k * (2 * (size_t)D * D + 3), *oL = o, *oI = o + (size_t)D * D, *tail = oI + (size_t)D * D;
    for (int idx = t; idx < D * D; idx += 64) A[(idx / D) * LD + idx % D] = src[idx];
    __syncthreads();
    int bad = -1;
    double badv = 0.0, logdet = 0.0;
    for (int c = 0; c < D; ++c) {
        double s = 0.0;
        if (t >= c && t < D) s = dot_sub(A[t * LD + c], A + t * LD, 1, A + c * LD, 1, c);
        const double piv = __shfl(s, c, 64);
        if (!(piv > 0.0) || !isfinite(piv)) {
            bad = c;
            badv = piv;
            break;
        }
        const double lcc = sqrt(piv);
        if (t == c) A[c * LD + c] = lcc;
        else if (t > c && t < D) A[t * LD + c] = s / lcc;
        __syncthreads();
    }
    if (bad >= 0) {
        for (int idx = t; idx < 2 * D * D; idx += 64) o[idx] = NAN;
        if (t == 0) {
            tail[0] = NAN;
            tail[1] = (double)(bad + 1);
            tail[2] = badv;
        }
        return;
    }
    const double lg = t < D ? log(A[t * LD + t]) : 0.0;
    for (int c = 0; c < D; ++c) logdet += __shfl(lg, c, 64);        // left to right, as the reference sums
    if (t < D) {
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
            if (i == t) v = 1.0 / A[i * LD + i];
            else if (i > t) v = dot_sub(0.0, A + i * LD + t, 1, X + t * LD + t, LD, i - t) / A[i * LD + i];
            X[i * LD + t] = v;
            oL[i * D + t] = i >= t ? A[i * LD + t] : 0.0;
        }
    }
    __syncthreads();
    if (t < D)
        for (int i = 0; i < D; ++i) oI[i * D + t] = dot_add(0.0, X + i * LD + i, LD, X + i * LD + t, LD, D - i);
    if (t == 0) {
        tail[0] = 2.0 * logdet;
        tail[1] = 0.0;
        tail[2] = 0.0;
    }
}

// ---- the expectations an E-step starts with (variational.pyx:759-772, :800-804) and the pack's constants --------
// E[ln|Lambda_k|] = sum_i psi((nu_k + 1 - i) / 2) + D ln 2 + ln|W_k|   (10.65)
// E[ln pi_k]      = psi(alpha_k) - psi(sum alpha)                         (10.66)
// c0 = D / beta_k, c3 = E[ln|Lambda_k|] - D ln 2 pi (enum pmc_kind, PMC_KIND_VB)
// ext = [E[ln pi] K | sum_i psi(...) + D ln 2, K] from the caller, or NULL: with the default prior nu0 = D - 1 + 1e-5 the
// sum holds psi(5e-6) = -2e5, an ulp of which is 3e-11 of every exponent -- a caller that must reproduce the reference's
// responsibilities element by element (the golden vb_* fixtures: 1e-10 on entries of 1e-79) needs the reference's psi
// (scipy's), bit for bit, not merely a psi that is as accurate.
__global__ __launch_bounds__(256) void k_vb_expect(pmc_vb_fields f, int K, int D, const double *ext, double *c0, double *c3)
{
    __shared__ double total_s;
    if (threadIdx.x == 0 && !ext) {
        double total = 0.0;
        for (int q = 0; q < K; ++q) total += f.alpha[q];            // in component order
        total_s = vb_digamma(total);
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    double s, lpi;
    if (ext) {
        lpi = ext[k];
        s = ext[K + k];
    } else {
        const double nu = f.nu[k];
        s = 0.0;
        for (int i = 1; i <= D; ++i) s += vb_digamma(0.5 * (nu + 1.0 - (double)i));
        s += D * VB_LN2;
        lpi = vb_digamma(f.alpha[k]) - total_s;
    }
    const double lam = s + f.log_det_W[k];
    f.ln_lambda[k] = lam;
    f.ln_pi[k] = lpi;
    if (c0) c0[k] = D / f.beta[k];
    if (c3) c3[k] = lam - D * VB_LN2PI;
}

// ---- behind an E-step: the converted sums into the state, the small block the host reads -----------------------
// conv = pmc_convert_stats_device's block [S0 K | M1 K D | mean K D | cov K D D | far K | scalars 8].
// small = [N_comp K (zeros -> tiny, variational.pyx:699-709) | far K | mean all finite K | S any finite K | scalars 8]
__global__ __launch_bounds__(64) void k_vb_after(const double *conv, pmc_vb_fields f, int K, int D, double *small, double *shift_prev,
                                                double *log_q_Z)
{
    const int k = blockIdx.x, t = threadIdx.x;
    const double *S0 = conv, *mean = S0 + K + (size_t)K * D, *cov = mean + (size_t)K * D, *far = cov + (size_t)K * D * D;
    const double s0 = S0[k], n = s0 == 0.0 ? VB_TINY : s0;
    int mean_ok = 1, s_any = 0;
    for (int i = t; i < D; i += 64) {
        const double v = mean[(size_t)k * D + i];
        f.x_mean[(size_t)k * D + i] = v;
        shift_prev[(size_t)k * D + i] = v;
        if (!isfinite(v)) mean_ok = 0;
    }
    for (int e = t; e < D * D; e += 64) {
        const double v = cov[(size_t)k * D * D + e];
        f.S[(size_t)k * D * D + e] = v;
        if (isfinite(v)) s_any = 1;
    }
    mean_ok = __all(mean_ok);
    s_any = __any(s_any);
    if (t == 0) {
        f.N_comp[k] = n;
        small[k] = n;
        small[K + k] = far[k];
        small[2 * K + k] = mean_ok ? 1.0 : 0.0;
        small[3 * K + k] = s_any ? 1.0 : 0.0;
    }
    if (k == 0 && t < PMC_NSCALARS) small[4 * K + t] = far[K + t];
    if (k == 0 && t == 0) *log_q_Z = far[K];
}

// conversion (pmc_convert.h) and k_vb_after in one launch: K-sized kernels cost 4-5 us each whatever they do
__global__ __launch_bounds__(256) void k_vb_convert_after(const double *stats, const double *shift, const double *scalars, double *conv,
                                                         pmc_vb_fields f, int K, int D, double *small, double *shift_prev, double *log_q_Z)
{
    pmc_convert_stats_block(stats, shift, nullptr, K, D, scalars, conv);
    __syncthreads();                                                // (this workgroup's means and covariances are written)
    const int k = blockIdx.x, t = threadIdx.x;
    const double *S0 = conv, *mean = S0 + K + (size_t)K * D, *cov = mean + (size_t)K * D, *far = cov + (size_t)K * D * D;
    const double s0 = S0[k], n = s0 == 0.0 ? VB_TINY : s0;
    int mean_ok = 1, s_any = 0;
    for (int i = t; i < D; i += 256) {
        const double v = mean[(size_t)k * D + i];
        f.x_mean[(size_t)k * D + i] = v;
        shift_prev[(size_t)k * D + i] = v;
        if (!isfinite(v)) mean_ok = 0;
    }
    for (int e = t; e < D * D; e += 256) {
        const double v = cov[(size_t)k * D * D + e];
        f.S[(size_t)k * D * D + e] = v;
        if (isfinite(v)) s_any = 1;
    }
    mean_ok = __syncthreads_and(mean_ok);
    s_any = __syncthreads_or(s_any);
    if (t == 0) {
        f.N_comp[k] = n;
        small[k] = n;
        small[K + k] = far[k];
        small[2 * K + k] = mean_ok ? 1.0 : 0.0;
        small[3 * K + k] = s_any ? 1.0 : 0.0;
    }
    // (the scalars: from the launch's argument, not from workgroup 0's copy in `conv`, which another workgroup may not see yet)
    if (k == 0 && t < PMC_NSCALARS) small[4 * K + t] = scalars ? scalars[t] : 0.0;
    if (k == 0 && t == 0) *log_q_Z = scalars ? scalars[0] : 0.0;
}

// the shifts of a second statistics pass: about the mean just found (variational.py::E_step, _stats.py)
__global__ void k_vb_newshift(const double *conv, const double *shift, int K, int D, double *out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * D) return;
    const int k = idx / D;
    const double s0 = conv[k], m1 = conv[K + idx];
    const double n = s0 == 0.0 ? VB_TINY : s0;
    out[idx] = s0 > 1e-200 ? shift[idx] + m1 / n : shift[idx];
}

// ---- the bound (variational.pyx:194-209, :948-1034; Wishart_log_B :1220-1247, Dirichlet_log_C :1269-1275) ------
__device__ inline double wishart_log_B(int D, double nu, double log_det, int t)
{
    const double g = wave_sum(t < D ? vb_lgamma(0.5 * (nu + 1.0 - (double)(t + 1))) : 0.0);
    return -0.5 * nu * log_det - 0.5 * nu * D * VB_LN2 - 0.25 * D * (D - 1) * VB_LNPI - g;
}

// per component: the summands of the seven terms, terms[k][VB_NTERMS]
__device__ inline void vb_bound_total(const double *terms, int K, int D, const double *log_q_Z, double *out, int t, bool coherent);

// (ticket != NULL: the workgroup that draws the last ticket adds the terms up -- one launch instead of two; the terms travel as
//  agent-scope relaxed atomics, written back before the ticket is drawn, as the pieces of k_logpdf_split do)
__global__ __launch_bounds__(64) void k_vb_bound_terms(pmc_vb_fields f, int K, int D, double *terms, unsigned *ticket,
                                                      const double *log_q_Z, double *out)
{
    const int k = blockIdx.x, t = threadIdx.x;
    const double *W = f.W + (size_t)k * D * D, *S = f.S + (size_t)k * D * D, *iW0 = f.inv_W0 + (size_t)k * D * D;
    double n = f.N_comp[k];
    n = n == 0.0 ? VB_TINY : n;
    const double nu = f.nu[k], nu0 = f.nu0[k], beta = f.beta[k], beta0 = f.beta0[k], alpha = f.alpha[k], alpha0 = f.alpha0[k];
    const double lam = f.ln_lambda[k], lpi = f.ln_pi[k], ldw = f.log_det_W[k], ldw0 = f.log_det_W0[k];
    double dx = 0.0, dm = 0.0;
    if (t < D) {
        const double m = f.m[(size_t)k * D + t];
        dx = f.x_mean[(size_t)k * D + t] - m;
        dm = m - f.m0[(size_t)k * D + t];
    }
    // lane j: sums over i of S_ij W_ji, inv_W0_ij W_ji, dx_i W_ij dx_j, dm_i W_ij dm_j
    double tr_sw = 0.0, tr_0 = 0.0, q_x = 0.0, q_m = 0.0;
#pragma unroll 4
    for (int i = 0; i < D; ++i) {
        const double dxi = __shfl(dx, i, 64), dmi = __shfl(dm, i, 64);
        if (t < D) {
            const double wji = W[t * D + i], wij = W[i * D + t];
            tr_sw += S[i * D + t] * wji;
            tr_0 += iW0[i * D + t] * wji;
            q_x += dxi * wij * dx;
            q_m += dmi * wij * dm;
        }
    }
    tr_sw = wave_sum(tr_sw);
    tr_0 = wave_sum(tr_0);
    q_x = wave_sum(q_x);
    q_m = wave_sum(q_m);
    const double log_b0 = wishart_log_B(D, nu0, ldw0, t), log_b = wishart_log_B(D, nu, ldw, t);
    unsigned drawn = 0;
    if (t == 0) {
        double o[VB_NTERMS];
        // (10.71) N_k (E[ln|Lambda|] - D / beta - nu (tr(S W) + dx^T W dx) - D ln 2 pi)
        o[0] = n * (lam - D / beta - nu * (tr_sw + q_x) - D * VB_LN2PI);
        o[1] = n * lpi;                                             // (10.72)
        o[2] = (alpha0 - 1.0) * lpi;                                // (10.73)
        // (10.74)
        double r = D * log(beta0 / (2. * 3.14159265358979323846));
        r += lam - D * beta0 / beta - beta0 * nu * q_m;
        r += 2 * log_b0;
        r += (nu0 - D - 1) * lam;
        r -= nu * tr_0;
        o[3] = r;
        o[4] = (alpha - 1.0) * lpi;                                 // (10.76)
        // (10.77): H[Wishart] (B.82) with E[ln|Lambda|] (B.81) = lam
        const double entropy = -log_b - 0.5 * (nu - D - 1) * lam + 0.5 * nu * D;
        o[5] = 0.5 * (lam + D * log(beta / (2. * 3.14159265358979323846))) - entropy;
        o[6] = vb_lgamma(alpha0);
        o[7] = vb_lgamma(alpha);
        o[8] = alpha0;
        o[9] = alpha;
        double *dst = terms + (size_t)k * VB_NTERMS;
        if (!ticket) {
            for (int i = 0; i < VB_NTERMS; ++i) dst[i] = o[i];
        } else {
            for (int i = 0; i < VB_NTERMS; ++i) __hip_atomic_store(dst + i, o[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!ticket) return;
    drawn = __shfl(drawn, 0, 64);
    if (drawn != (unsigned)(K - 1)) return;
    if (t == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ready for the next launch)
    vb_bound_total(terms, K, D, log_q_Z, out, t, true);
}

// the sums over the components, each in component order; out = [bound | log p(X) | log p(Z) | log p(pi) | log p(mu, Lambda) |
// log q(Z) | log q(pi) | log q(mu, Lambda)].  One wavefront; coherent: the terms were written by other workgroups of this launch.
__device__ inline void vb_bound_total(const double *terms, int K, int D, const double *log_q_Z, double *out, int t, bool coherent)
{
    double s = 0.0;
    if (t < VB_NTERMS) {
        int k = 0;
        for (; k + 8 <= K; k += 8) {                                // (loads eight at a time, the sum in component order)
            double v8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double *src = terms + (size_t)(k + q) * VB_NTERMS + t;
                v8[q] = coherent ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v8[q];
        }
        for (; k < K; ++k) {
            const double *src = terms + (size_t)k * VB_NTERMS + t;
            s += coherent ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
        }
    }
    double v[VB_NTERMS];
    for (int i = 0; i < VB_NTERMS; ++i) v[i] = __shfl(s, i, 64);
    if (t != 0) return;
    const double p_X = 0.5 * v[0], p_Z = v[1];
    const double p_pi = (vb_lgamma(v[8]) - v[6]) + v[2];
    const double p_ml = 0.5 * v[3];
    const double q_Z = *log_q_Z;
    const double q_pi = v[4] + (vb_lgamma(v[9]) - v[7]);
    const double q_ml = -0.5 * K * D + v[5];
    double b = p_X;
    b += p_Z;
    b += p_pi;
    b += p_ml;
    b -= q_Z;
    b -= q_pi;
    b -= q_ml;
    out[0] = b;
    out[1] = p_X;
    out[2] = p_Z;
    out[3] = p_pi;
    out[4] = p_ml;
    out[5] = q_Z;
    out[6] = q_pi;
    out[7] = q_ml;
}

int vfail(int code, const char *msg) { return pmc_internal_fail(code, msg); }

int launched(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return PMC_OK;
    char buf[200];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    return vfail(PMC_EHIP, buf);
}

bool fields_ok(const pmc_vb_fields *f)
{
    return f && f->alpha0 && f->beta0 && f->nu0 && f->m0 && f->inv_W0 && f->log_det_W0 && f->alpha && f->beta && f->nu && f->m &&
           f->W && f->log_det_W && f->ln_lambda && f->ln_pi && f->N_comp && f->x_mean && f->S;
}

}  // namespace

extern "C" {

double pmc_host_digamma(double x) { return vb_digamma(x); }
double pmc_host_lgamma(double x) { return vb_lgamma(x); }

int pmc_vb_max_dim(void) { return 64; }

int pmc_vb_mstep_device(int K, int D, const pmc_vb_fields *f, double *d_status, void *stream)
{
    if (K < 1 || D < 1 || D > 64 || !fields_ok(f) || !d_status) return vfail(PMC_EINVAL, "pmc_vb_mstep_device: bad argument (1 <= D <= 64)");
    hipLaunchKernelGGL(k_vb_mstep, dim3((unsigned)K), dim3(64), sizeof(double) * 2 * (size_t)D * (D + 1), (hipStream_t)stream, *f, K, D,
                       d_status);
    return launched("k_vb_mstep launch");
}

int64_t pmc_spd_inverse_len(int K, int D)
{
    return (K < 1 || D < 1) ? (int64_t)vfail(PMC_EINVAL, "pmc_spd_inverse_len: bad K / D") : (int64_t)K * (2 * (int64_t)D * D + 3);
}

int pmc_spd_inverse_device(int K, int D, const double *d_A, double *d_out, void *stream)
{
    if (K < 1 || D < 1 || D > 64 || !d_A || !d_out) return vfail(PMC_EINVAL, "pmc_spd_inverse_device: bad argument (1 <= D <= 64)");
    hipLaunchKernelGGL(k_spd_inverse, dim3((unsigned)K), dim3(64), sizeof(double) * 2 * (size_t)D * (D + 1), (hipStream_t)stream, d_A, K, D,
                       d_out);
    return launched("k_spd_inverse launch");
}

int pmc_vb_mstep_status(int K, const double *h_status)
{
    if (K < 1 || !h_status) return vfail(PMC_EINVAL, "pmc_vb_mstep_status: bad argument");
    for (int k = 0; k < K; ++k)
        if (h_status[k] != 0.0) {
            char buf[200];
            snprintf(buf, sizeof(buf), "M-step: W^-1 of component %d is not positive definite (pivot %d = %g)", k, (int)h_status[k] - 1,
                     h_status[K + k]);
            return vfail(PMC_ENOTPOSDEF, buf);
        }
    return PMC_OK;
}

int pmc_vb_expectations_device(int K, int D, const pmc_vb_fields *f, const double *d_psi_parts, double *d_c0, double *d_c3, void *stream)
{
    if (K < 1 || D < 1 || !fields_ok(f)) return vfail(PMC_EINVAL, "pmc_vb_expectations_device: bad argument");
    hipLaunchKernelGGL(k_vb_expect, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *f, K, D, d_psi_parts, d_c0,
                       d_c3);
    return launched("k_vb_expect launch");
}

int64_t pmc_vb_small_len(int K) { return K < 1 ? (int64_t)vfail(PMC_EINVAL, "pmc_vb_small_len: bad K") : 4 * (int64_t)K + PMC_NSCALARS; }

int pmc_vb_after_device(int K, int D, const double *d_conv, const pmc_vb_fields *f, double *d_small, double *d_shift_prev,
                        double *d_log_q_Z, void *stream)
{
    if (K < 1 || D < 1 || !d_conv || !fields_ok(f) || !d_small || !d_shift_prev || !d_log_q_Z)
        return vfail(PMC_EINVAL, "pmc_vb_after_device: bad argument");
    hipLaunchKernelGGL(k_vb_after, dim3((unsigned)K), dim3(64), 0, (hipStream_t)stream, d_conv, *f, K, D, d_small, d_shift_prev, d_log_q_Z);
    return launched("k_vb_after launch");
}

int pmc_vb_convert_after_device(int K, int D, const double *d_stats, const double *d_shift, const double *d_scalars, double *d_conv,
                                const pmc_vb_fields *f, double *d_small, double *d_shift_prev, double *d_log_q_Z, void *stream)
{
    if (K < 1 || D < 1 || !d_stats || !d_shift || !d_conv || !fields_ok(f) || !d_small || !d_shift_prev || !d_log_q_Z)
        return vfail(PMC_EINVAL, "pmc_vb_convert_after_device: bad argument");
    hipLaunchKernelGGL(k_vb_convert_after, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, d_stats, d_shift, d_scalars, d_conv, *f, K,
                       D, d_small, d_shift_prev, d_log_q_Z);
    return launched("k_vb_convert_after launch");
}

int pmc_vb_newshift_device(int K, int D, const double *d_conv, const double *d_shift, double *d_out, void *stream)
{
    if (K < 1 || D < 1 || !d_conv || !d_shift || !d_out) return vfail(PMC_EINVAL, "pmc_vb_newshift_device: bad argument");
    const int n = K * D;
    hipLaunchKernelGGL(k_vb_newshift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_conv, d_shift, K, D, d_out);
    return launched("k_vb_newshift launch");
}

int64_t pmc_vb_bound_scratch_len(int K) { return K < 1 ? (int64_t)vfail(PMC_EINVAL, "pmc_vb_bound_scratch_len: bad K") : (int64_t)K * VB_NTERMS + 8; }

int pmc_vb_bound_device(int K, int D, const pmc_vb_fields *f, const double *d_log_q_Z, double *d_scratch, double *d_out, void *stream)
{
    if (K < 1 || D < 1 || D > 64 || !fields_ok(f) || !d_log_q_Z || !d_scratch || !d_out)
        return vfail(PMC_EINVAL, "pmc_vb_bound_device: bad argument (1 <= D <= 64)");
    // one launch: the workgroup that draws the last ticket (the counter behind the terms: zero between launches) adds up
    hipLaunchKernelGGL(k_vb_bound_terms, dim3((unsigned)K), dim3(64), 0, (hipStream_t)stream, *f, K, D, d_scratch,
                       (unsigned *)(d_scratch + (size_t)K * VB_NTERMS), d_log_q_Z, d_out);
    return launched("k_vb_bound_terms launch");
}

}  // extern "C"
