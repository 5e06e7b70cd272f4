"""A backend for the CPU-only test-suite: the same interface as pypmc_amd.backend.HipBackend,
answered by the CPU oracle (oracle/).  TEST INFRASTRUCTURE -- it lets ``-m "not gpu"`` tests
exercise the host-side logic of the front-end (parameter handling, M-step, pruning, convergence
control, sharding + all-reduce) where no GPU exists.  The package itself never imports it."""
import numpy as np

from oracle import oracle as orc
from pypmc_amd._lib import NSCALARS, PMC_KIND_GAUSS, PMC_KIND_STUDENT_T, PMC_KIND_VB

TINY = np.finfo('d').tiny


class OracleBackend(object):
    name = "oracle"

    # plumbing: "device" arrays are numpy arrays
    def asdevice(self, a, dtype=None):
        return np.ascontiguousarray(a, dtype=np.float64 if dtype is None else np.int64)

    def tohost(self, t):
        return np.asarray(t)

    def zeros(self, shape, dtype=None):
        return np.zeros(shape)

    def empty(self, shape, dtype=None):
        return np.empty(shape)

    def stats_len(self, K, D):
        return NSCALARS + K * (1 + D + D * (D + 1) // 2) + 2 * K

    # operations
    def _individual(self, x, cs):
        N = len(x)
        ind = np.zeros((N, cs.K))
        if cs.kind == PMC_KIND_GAUSS:
            orc.mixture_multi_evaluate(0, x, cs.weight, cs.mu, cs.precision, cs.c0, individual=ind,
                                       components=list(range(cs.K)))
        elif cs.kind == PMC_KIND_STUDENT_T:
            orc.mixture_multi_evaluate(1, x, cs.weight, cs.mu, cs.precision, cs.c0, cs.c1, cs.c2,
                                       individual=ind, components=list(range(cs.K)))
        else:
            raise ValueError(cs.kind)
        return ind

    def _lse(self, ind, w, max_init_zero):
        if max_init_zero:       # a dead all-zero column with weight 0 joins the row maximum
            ind = np.hstack((ind, np.zeros((len(ind), 1))))
            w = np.concatenate((w, [0.]))
        return orc.logsumexp2D(ind, w)

    def logpdf(self, x, comps, want_out=True, individual=None, want_individual=False,
               max_init_zero=False, log_target=None, sample_w=None, want_scalars=False, pack=None,
               out=None):
        x = self.asdevice(x)
        N = len(x)
        ind = self._individual(x, comps)
        lse = self._lse(ind, comps.weight, max_init_zero)
        if out is not None:
            out[:] = lse
            lse, want_out = out, True
        if individual is None and want_individual:
            individual = np.empty((N, comps.ld))
        if individual is not None:
            individual[:, comps.column] = ind
        weights, scalars = None, None
        sc = np.zeros(NSCALARS)
        if log_target is not None:
            tmp = np.asarray(log_target, dtype=float) - lse
            with np.errstate(over='ignore'):
                weights = np.exp(tmp)
            sc[0] = weights.sum()
            nz = weights != 0
            sc[1] = (weights[nz] * tmp[nz]).sum()
            sc[2] = (weights ** 2).sum()
            sc[4] = np.count_nonzero(np.isinf(weights) & ~np.isinf(tmp))
        sc[3] = (np.asarray(sample_w) * lse).sum() if sample_w is not None else lse.sum()
        if want_scalars:
            scalars = sc
        return dict(out=lse if want_out else None, individual=individual, weights=weights,
                    scalars=scalars)

    def weight_sums(self, w):
        w = np.asarray(w, dtype=float).reshape(-1)
        sc = np.zeros(NSCALARS)
        nz = w != 0
        sc[0], sc[1], sc[2] = w.sum(), (w[nz] * np.log(w[nz])).sum(), (w ** 2).sum()
        return sc

    def weighted_moments(self, x, w):
        x, w = np.asarray(x, dtype=float), np.asarray(w, dtype=float).reshape(-1)
        shift = x[0].copy()
        d = x - shift
        return float(w.sum()), w.dot(d), np.einsum('n,ni,nj->ij', w, d, d), shift, float((w ** 2).sum())

    def logsumexp2d(self, a, w):
        return orc.logsumexp2D(np.asarray(a, dtype=float), np.asarray(w, dtype=float))

    def combine_weights(self, q, counts, t, omega, n_total, log_scale):
        q = np.asarray(q, dtype=float)                       # T x N here, N x T in the reference
        with np.errstate(all='ignore'):
            out = orc.combine_weights_run(np.ascontiguousarray(q.T), counts, t, omega, n_total, log_scale)
        return out, np.array([float(np.count_nonzero(~np.isfinite(out)))])

    def estep(self, x, comps, mode, max_init_zero=False, sample_w=None, latent=None,
              want_r=False, want_log_rho=False, want_exponent=False, pack=None, out=None, shift=None):
        x = self.asdevice(x)
        N, D = x.shape
        K = comps.K
        sw = np.ones(N) if sample_w is None else np.asarray(sample_w, dtype=float)
        sc = np.zeros(NSCALARS)
        vs = np.zeros((K, 2))
        r_pub = lr_pub = ex_pub = None
        if comps.kind == PMC_KIND_VB:
            beta, nu = D / comps.c0, comps.c1
            ln_pi, ln_lambda = comps.c2, comps.c3 + D * np.log(2. * np.pi)
            res = orc.vb_estep(x, sample_w, comps.mu, comps.precision, beta, nu, ln_pi, ln_lambda)
            r = res["r"]
            u = sw[:, None] * r
            sc[0] = res["expectation_log_q_Z"]
            r_pub, lr_pub, ex_pub = r, res["log_rho"], res["expectation_gauss_exponent"]
        else:
            ind = self._individual(x, comps)
            lse = self._lse(ind, comps.weight, max_init_zero)
            if mode == 2:
                r = (np.asarray(latent)[:, None] == comps.column[None, :]).astype(float)
            else:
                r = np.exp(ind) * comps.weight[None, :] / (np.exp(lse) + TINY)[:, None]
            u = sw[:, None] * r
            sc[3] = (sw * lse).sum()
            r_pub = r
            if comps.kind == PMC_KIND_STUDENT_T:
                d = x[:, None, :] - comps.mu[None, :, :]
                maha = np.einsum('nki,kij,nkj->nk', d, comps.precision, d)
                dof = comps.c3
                gamma = (dof[None, :] + D) / (dof[None, :] + maha)
                vs[:, 0] = u.sum(axis=0)
                vs[:, 1] = (u * np.log(.5 * (maha + dof[None, :]))).sum(axis=0)
                u = u * gamma
        d = x[:, None, :] - (comps.mu if shift is None else np.asarray(shift, dtype=float).reshape(K, D))[None, :, :]
        S0 = u.sum(axis=0)
        M1 = np.einsum('nk,nki->ki', u, d)
        M2 = np.einsum('nk,nki,nkj->kij', u, d, d)
        il, jl = np.tril_indices(D)
        body = np.concatenate((S0[:, None], M1, M2[:, il, jl]), axis=1)
        flat = np.concatenate((sc, body.ravel(), vs.ravel()))
        if out is not None:
            out[:] = flat
            flat = out

        def spread(a):
            full = np.zeros((N, comps.ld))
            full[:, comps.column] = a
            return full
        return dict(stats=flat, r=spread(r_pub) if want_r else None,
                    log_rho=spread(lr_pub) if want_log_rho and lr_pub is not None else None,
                    exponent=spread(ex_pub) if want_exponent and ex_pub is not None else None)
