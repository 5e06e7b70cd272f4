#!/usr/bin/env python3
"""Ozaki-style slicing of the PMC statistics (verdict r5 #4): how many int8 slice products does

    T[k][m] = sum_n u_nk z_nm,   z_n = the monomials of d = x_n - c up to degree 2   (pmc.pyx:188-222 as one matrix product:
                                                                                       k_stats_gemm, pmc_stats.hip)

need for 1e-11?  The scheme (Ozaki et al., error-free transformation for matrix multiplication; the int8 variant): per
block of L samples and per column, a shared exponent and 7-bit signed digits,

    u_nk = 2^e_k sum_s 2^(-7 (s + 1)) U_s[n, k],      z_nm = 2^f_m sum_t 2^(-7 (t + 1)) Z_t[n, m],     U_s, Z_t int8,

every product U_s^T Z_t accumulated EXACTLY in int32 (L <= 2^31 / 127^2 = 133 k samples), the products with s + t < T kept,
scaled and added in fp64.  Emulated here with integer arithmetic in numpy against a long-double loop on config 5's
shape (D = 40, proposal of overlapping components, importance-weighted responsibilities), error normalised as the
parity tests normalise the statistics: N_k relative; first moments against sigma_i N_k; second moments against
sigma_i sigma_j N_k.

    python scripts/microbench/ozaki_accuracy.py
"""
import numpy as np

D, K, N, L = 40, 8, 60000, 60000                 # (|products| < 127^2 N < 2^53: the fp64 BLAS products below are exact integers)
rs = np.random.RandomState(5)
mu = rs.normal(0, 0.6, (K, D))
A = rs.normal(size=(D, D))
cov = A.dot(A.T) / D + 0.5 * np.eye(D)
Lc = np.linalg.cholesky(cov)
comp = rs.randint(0, K, N)
x = mu[comp] + rs.normal(size=(N, D)).dot(Lc.T) * 1.2
# responsibilities of the (equal-weight, shared-covariance) mixture times importance weights of wide spread
inv = np.linalg.inv(cov)
d = x[:, None, :] - mu[None]
a = -0.5 * np.einsum('nki,ij,nkj->nk', d, inv, d)
rho = np.exp(a - a.max(axis=1, keepdims=True))
rho /= rho.sum(axis=1, keepdims=True)
w = np.exp(rs.normal(0, 1.5, N))
u = w[:, None] * rho                                         # N x K
c = 0.5 * (mu.min(axis=0) + mu.max(axis=0))
dc = x - c
il, jl = np.tril_indices(D)
z = np.concatenate([np.ones((N, 1)), dc, dc[:, il] * dc[:, jl]], axis=1)   # N x 861
M = z.shape[1]

ref = (u.astype(np.longdouble).T @ z.astype(np.longdouble)).astype(np.float64)      # K x M, long double accumulation
f64 = u.T @ z

S0 = ref[:, 0]
sig = np.sqrt(np.diag(cov))
scale = np.concatenate([[1.0], sig, sig[il] * sig[jl]])[None, :] * S0[:, None]


def slices(v, nsl):
    """per column: exponent e (max |v| < 2^e), digits D_s int8 with v = 2^e sum_s 2^(-7 (s + 1)) D_s, rounding to nearest"""
    e = np.ceil(np.log2(np.abs(v).max(axis=0) * (1 + 2.0 ** -40)))
    t = v / 2.0 ** e                                         # |t| < 1
    out = []
    for _ in range(nsl):
        t = t * 128.0
        q = np.rint(t)
        q = np.clip(q, -127, 127)
        out.append(q)                                         # (integers held in fp64)
        t = t - q
    return e, out


for T in (5, 6, 7, 8, 9):
    eu, U = slices(u, T)
    ez, Z = slices(z, T)
    acc = np.zeros((K, M))
    nprod = 0
    for s in range(T):
        for t in range(T - s):
            P = U[s].T @ Z[t]                                # exact integers (|P| < 127^2 N < 2^53 here; < 2^31 per L = 133 k)
            acc += P * 2.0 ** (-7.0 * (s + t + 2))
            nprod += 1
    got = acc * 2.0 ** eu[:, None] * 2.0 ** ez[None, :]
    err = np.abs(got - ref) / scale
    print("T = %d: %2d int8 products, max normalised error %.2e  (N_k %.1e, means %.1e, second moments %.1e)"
          % (T, nprod, err.max(), err[:, 0].max(), err[:, 1:1 + D].max(), err[:, 1 + D:].max()))
e64 = np.abs(f64 - ref) / scale
print("plain fp64 accumulation (numpy):        max normalised error %.2e" % e64.max())
