// pmc_propose.hip -- device side of MixtureDensity.propose (pypmc/density/mixture.pyx:159-212 with
// Gauss.propose gauss.pyx:50-52,159-163 and LocalStudentT.propose student_t.pyx:49-55), compiled once
// per sample dimension.
//
// The component counts are drawn on the HOST with the caller's generator (rng.multinomial -- they
// and the derived origin indices stay bit-exact with the reference); the device fills the samples:
// one lane = one sample n, its component k found from the exclusive prefix offsets of the counts,
//     x_n = mu_k + L_k z            z ~ N(0,1)^D          (Gauss)
//     x_n = mu_k + L_k z sqrt(nu_k / c),  c ~ chi^2(nu_k)  (Student-t)
// Random numbers: Philox4x32-10, counter = (sample index, draw index), key = seed -- stateless, so
// the stream of a sample does not depend on the launch geometry or on how samples are sharded over
// GPUs (every rank passes its global sample offset).  Normals by Box-Muller from 53-bit uniforms (lean log,
// sincospi: the angle is a multiple of pi by construction);
// chi-square(nu) = 2 Gamma(nu/2) by Marsaglia-Tsang rejection.  Sample values are statistically
// (not bitwise) equivalent to numpy's MT19937 stream, as SURVEY section 7 ("RNG parity") states.
#include "pmc_device.h"

namespace {

struct Philox {
    unsigned k0, k1;
    __device__ __forceinline__ static void round(unsigned (&c)[4], unsigned k0, unsigned k1)
    {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    }
    // 128 random bits for counter (n, draw)
    __device__ __forceinline__ void operator()(unsigned long long n, unsigned draw, unsigned (&out)[4]) const
    {
        unsigned c[4] = {(unsigned)n, (unsigned)(n >> 32), draw, 0x9E3779B9u};
        unsigned a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            round(c, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = c[i];
    }
};

// two uniforms: u1 in (0,1], u2 in [0,1), 53 bits each
__device__ __forceinline__ void uniforms(const unsigned (&r)[4], double &u1, double &u2)
{
    const unsigned long long a = ((unsigned long long)r[0] << 32 | r[1]) >> 11;
    const unsigned long long b = ((unsigned long long)r[2] << 32 | r[3]) >> 11;
    u1 = ((double)a + 1.0) * (1.0 / 9007199254740992.0);
    u2 = (double)b * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void normal_pair(const Philox &g, unsigned long long n, unsigned draw, double &z0,
                                            double &z1)
{
    unsigned r[4];
    g(n, draw, r);
    double u1, u2;
    uniforms(r, u1, u2);
#ifdef PMC_PROPOSE_LIBM
    const double rad = sqrt(-2.0 * log(u1));
    double s, c;
    sincos(6.283185307179586476925286766559 * u2, &s, &c);
#else
    // u1 is a positive normal number (>= 2^-53): the lean log; the angle 2 pi u2 as a multiple of pi needs no
    // range reduction against pi at all
    const double rad = sqrt(-2.0 * log_pos_finite(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
#endif
    z0 = rad * c;
    z1 = rad * s;
}

// chi^2(nu) = 2 * Gamma(nu/2): Marsaglia & Tsang (2000); draws start at index `draw0`
__device__ __forceinline__ double chi_square(const Philox &g, unsigned long long n, unsigned draw0, double nu)
{
    double a = 0.5 * nu, boost = 1.0;
    unsigned draw = draw0;
    if (a < 1.0) {                                        // Gamma(a) = Gamma(a+1) U^(1/a)
        unsigned r[4];
        g(n, draw++, r);
        double u1, u2;
        uniforms(r, u1, u2);
        boost = pow(u1, 1.0 / a);
        a += 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double result = d;                                    // fallback after 64 rejections (p < 1e-60)
    for (int it = 0; it < 64; ++it) {
        double z, zz;
        normal_pair(g, n, draw++, z, zz);
        unsigned r[4];
        g(n, draw++, r);
        double u1, u2;
        uniforms(r, u1, u2);
        const double t = 1.0 + c * z;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        if (log(u1) < 0.5 * z * z + d - d * v + d * log(v)) {
            result = d * v;
            break;
        }
    }
    return 2.0 * result * boost;
}

constexpr int PROPOSE_OFFS_LDS = 1025;   // prefix offsets of up to 1024 components sit in LDS (8 KB)
#if PMC_D > 0
constexpr int PCH = 16;          // coordinates staged per pass
constexpr int PPITCH = PCH + 1;  // row pitch of the staging buffer in doubles (odd: conflict-free column writes)

// x_i = mu_i + scale * sum_{j <= i} L_ij z_j, coordinates in passes of PCH through the wavefront's LDS stage so
// that they leave in 128-byte row segments (4 rows per store instruction) instead of 64 scattered 8-byte
// stores per coordinate.  The factor is ALWAYS read through the scalar cache (the coefficient is an SGPR operand of the
// multiply-add instead of a 64-lane vector load of one address): samples arrive ordered by component, so all but K - 1
// wavefronts draw from one component; a wavefront that straddles a boundary walks its distinct components one after the
// other, the lanes of the others idle (rounds 2-4 gave such wavefronts a second copy of the loop with per-lane vector
// loads).  At D >= 40 the kernel spills (256 registers at two wavefronts per SIMD, 65-71 more wanted: the D / 2 Box-Muller
// pairs and the D (D + 1) / 2 hoisted scalar loads); serialising the pairs or the rows' loads by artificial dependencies
// changes neither the spills nor the time, and one wavefront per SIMD without spills is slower (round 3) -- left as it is.
template <int D, bool PADDED>
__device__ __forceinline__ void affine_out(const double *chol, const double *mus, int k, int dreal, const double (&z)[D],
                                           double scale, double *st, int lane, double *__restrict__ out, int rows)
{
#pragma unroll
    for (int c0 = 0; c0 < D; c0 += PCH) {
        if (PADDED && c0 >= dreal) break;
        unsigned long long todo = __ballot(1);
        while (todo) {                                               // (one round for all but K - 1 wavefronts)
            const int first = __ffsll((long long)todo) - 1;
            const int kf = __builtin_amdgcn_readlane(k, first);
            const bool mine = k == kf;
            todo &= ~__ballot(mine);
            cdouble *L = (cdouble *)(chol + (size_t)kf * dreal * dreal);
            cdouble *mu = (cdouble *)(mus + (size_t)kf * dreal);
#pragma unroll
            for (int i = c0; i < c0 + PCH && i < D; ++i) {
                double acc = 0.0;
                if (!PADDED || i < dreal) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) acc = fma(L[i * dreal + j], z[j], acc);
                    acc = mu[i] + acc * scale;
                }
                if (mine) st[lane * PPITCH + (i - c0)] = acc;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wavefront's LDS writes have landed
        const int ncol = (dreal - c0 < PCH) ? dreal - c0 : PCH;  // columns of this pass
        if (ncol == PCH) {
            // lane = (row within a group of 4, column): 16 store instructions of 4 x 128 contiguous bytes
#pragma unroll
            for (int r0 = 0; r0 < 64; r0 += 4) {
                const int r = r0 + (lane >> 4), c = lane & 15;
                if (r < rows) out[(long long)r * dreal + c0 + c] = st[r * PPITCH + c];
            }
        } else {
            for (int e = lane; e < rows * ncol; e += 64) {
                const int r = e / ncol, c = e - r * ncol;
                out[(long long)r * dreal + c0 + c] = st[r * PPITCH + c];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // the reads are done before the next pass overwrites
    }
}

// (two wavefronts per SIMD: the unrolled Box-Muller pairs take every register they are given -- 316 at D = 40
//  without a bound, one wavefront per SIMD; bounded to 256: 5.3 -> 4.2 ms per 1.25e7 samples; to 128 it spills: 6.2)
#ifndef PMC_PROPOSE_WAVES
#define PMC_PROPOSE_WAVES 2
#endif
template <int D, bool PADDED>
__global__ __launch_bounds__(256, PMC_PROPOSE_WAVES) void k_propose(const PmcArgsP a)
{
    __shared__ double stage[4][64 * PPITCH];
    __shared__ long long offs[PROPOSE_OFFS_LDS];
    // the K + 1 prefix offsets -> LDS, one coalesced load per workgroup: the binary search below is log2 K DEPENDENT loads,
    // a memory latency each when they go to global memory (7 of them at K = 128: ~10 % of a wavefront's life with two
    // wavefronts per SIMD to hide it; K = 4 ran 30 % faster per sample than K = 128 at D = 40)
    const bool in_lds = a.K + 1 <= PROPOSE_OFFS_LDS;
    if (in_lds) {
        for (int t = threadIdx.x; t <= a.K; t += 256) offs[t] = a.offsets[t];
        __syncthreads();
    }
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n0 = n - lane;                               // first sample of this wavefront
    if (n0 >= a.N) return;                                       // wave-uniform
    const bool valid = n < a.N;
    const int dreal = PADDED ? a.dreal : D;
    // component of sample n: offsets[k] <= n < offsets[k+1]  (binary search, K+1 entries)
    const long long nn = valid ? n : a.N - 1;
    int lo = 0, hi = a.K;
    if (in_lds) {
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offs[mid] <= nn) lo = mid; else hi = mid;
        }
    } else {
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.offsets[mid] <= nn) lo = mid; else hi = mid;
        }
    }
    const int k = lo;
    const Philox g = {(unsigned)a.seed, (unsigned)(a.seed >> 32)};
    const unsigned long long gn = (unsigned long long)(a.first_sample + nn);  // global sample index

    double z[D];
#pragma unroll
    for (int j = 0; j < D; j += 2) {
        double z0, z1;
        normal_pair(g, gn, (unsigned)(j >> 1), z0, z1);
        z[j] = z0;
        if (j + 1 < D) z[j + 1] = z1;
    }
    double scale = 1.0;
    if (a.dof != nullptr) {
        const double nu = a.dof[k];
        scale = sqrt(nu / chi_square(g, gn, 0x10000u, nu));      // student_t.pyx:55
    }
    const int rows = (int)((a.N - n0 < 64) ? a.N - n0 : 64);     // samples of this wavefront
    double *__restrict__ out = a.x + n0 * (long long)dreal;
    affine_out<D, PADDED>(a.chol, a.mu, k, dreal, z, scale, stage[wave], lane, out, rows);
    if (a.origin != nullptr && valid) a.origin[n] = k;
}

#else   // PMC_D == 0: the run-time-dimension unit (D > PMC_MAX_DIM)

// x = mu_k + scale L_k z without D registers per lane: the lane's normals are written into its own row of the output
// first and transformed IN PLACE, 16 coordinates at a time from the last block to the first -- block I needs
// z_j for j < 16 (I + 1) only, all still intact, and its result overwrites z of block I.  16 accumulators and 16
// staged z per lane; L through pointer type P (the scalar cache when the wavefront draws from one component).
template <class P>
__device__ __forceinline__ void affine_in_place(P L, P mu, int D, double scale, double *row)
{
    const int G = (D + 15) >> 4;
    for (int I = G - 1; I >= 0; --I) {
        double acc[16];
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) acc[ii] = 0.0;
        for (int J = 0; J <= I; ++J) {
            double z[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {              // (unconditional load, selected afterwards: no branch)
                const int j = 16 * J + jj;
                const double v = row[j < D ? j : D - 1];
                z[jj] = j < D ? v : 0.0;
            }
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) {
                const int i = 16 * I + ii;
                const int ic_ = i < D ? i : D - 1;                  // rows beyond D: computed on row D - 1, discarded
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int j = 16 * J + jj;
                    const int jc = j < D ? j : D - 1;               // columns beyond D: z = 0
                    // the strict upper triangle is never read (mixture.pyx only multiplies by the lower factor)
                    if (J < I || jj <= ii) acc[ii] = fma(L[(long long)ic_ * D + jc], (J < I || jj <= ii) ? z[jj] : 0.0, acc[ii]);
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
            const int i = 16 * I + ii;
            if (i < D) row[i] = mu[i] + acc[ii] * scale;
        }
    }
}

__global__ __launch_bounds__(256) void k_propose_big(const PmcArgsP a)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const long long n0 = n - lane;
    if (n0 >= a.N) return;                                       // wave-uniform
    const bool valid = n < a.N;
    const int D = a.dreal;
    const long long nn = valid ? n : a.N - 1;
    int lo = 0, hi = a.K;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.offsets[mid] <= nn) lo = mid; else hi = mid;
    }
    const int k = lo;
    const Philox g = {(unsigned)a.seed, (unsigned)(a.seed >> 32)};
    const unsigned long long gn = (unsigned long long)(a.first_sample + nn);
    double scale = 1.0;
    if (a.dof != nullptr) {
        const double nu = a.dof[k];
        scale = sqrt(nu / chi_square(g, gn, 0x10000u, nu));      // student_t.pyx:55
    }
    const int kfirst = __builtin_amdgcn_readfirstlane(k);
    const bool uniform = __all(k == kfirst);
    if (valid) {
        double *row = a.x + n * (long long)D;
        for (int j = 0; j < D; j += 2) {
            double z0, z1;
            normal_pair(g, gn, (unsigned)(j >> 1), z0, z1);
            row[j] = z0;
            if (j + 1 < D) row[j + 1] = z1;
        }
        if (uniform)
            affine_in_place((cdouble *)(a.chol + (size_t)kfirst * D * D), (cdouble *)(a.mu + (size_t)kfirst * D), D, scale, row);
        else
            affine_in_place(a.chol + (size_t)k * D * D, a.mu + (size_t)k * D, D, scale, row);
        if (a.origin != nullptr) a.origin[n] = k;
    }
}
#endif

}  // namespace

#if PMC_D > 0
extern "C" hipError_t PMC_UNIT_NAME_X(pmc_launch_propose_d, PMC_D, PMC_PADDED)(const PmcArgsP &a, unsigned grid,
                                                                              hipStream_t st)
{
    hipLaunchKernelGGL((k_propose<D_, P_>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}
#else
extern "C" hipError_t pmc_launch_propose_big(const PmcArgsP &a, unsigned grid, hipStream_t st)
{
    hipLaunchKernelGGL(k_propose_big, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}
#endif
