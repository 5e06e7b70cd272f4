"""(M-)PMC updates of Gaussian and Student-t mixture proposals [Cap+08, Kil+09, HOD12] with the
responsibilities and all N-sized reductions on the GPU (reference: pypmc/mix_adapt/pmc.pyx).

One kernel pass produces rho_nk (Rao-Blackwellised, pmc.pyx:23-43, or the latent one-hot form,
:45-51) -- for Student-t also gamma_nk (:602-610) -- directly in the layout of the statistics
kernel, which reduces  sum u | sum u d | sum u d d^T  with d = x - mu_k^old and u = w rho [gamma].
The host turns these K-sized sums into the reference's alpha, mu, Sigma (and the constant of the
degree-of-freedom condition), applies ``component.update`` with the LinAlgError fall-back, and
does the root finding and pruning.  With torch.distributed initialised the sums are all-reduced
(pypmc_amd.parallel) and every rank performs the same update.
"""
import logging
from copy import deepcopy

import numpy as np
from scipy.optimize import brentq
from scipy.special import digamma

from .. import parallel
from .._lib import PMC_RESP_PMC_RB, PMC_RESP_PMC_LATENT
from ..backend import get_backend
from ..density.gauss import Gauss
from ..density.student_t import StudentT
from ..density.mixture import MixtureDensity, component_set
from ._stats import convert_stats, split_stats, centred_moments, shift_is_far, regularize
from ..tools._linalg import single_threaded_blas, chol_inv_det_batch

logger = logging.getLogger(__name__)


def _prepare_pmc_update(samples, weights, latent, mincount, density, rb, copy, backend, mahalanobis=None,
                        responsibilities=None, student=False):
    """Argument checks, live-component bookkeeping and the device pass
    (reference: pmc.pyx:53-118).  Returns density, live_components (after ``mincount`` pruning),
    the indices the statistics were computed for, the host statistics, the weight normalisation
    and the renormalisation flag."""
    need_renormalize = False
    if copy:
        density = deepcopy(density)
    N_local = len(samples)
    on_device = lambda a: a is not None and not isinstance(a, (np.ndarray, list, tuple)) and hasattr(a, 'device')
    if weights is not None:
        if not on_device(weights):                  # device tensors (device-resident loops) pass through
            weights = np.asarray(weights)
        assert len(weights.shape) == 1, 'Weights must be one-dimensional.'
        assert len(weights) == N_local, \
            "Number of weights (%s) does not match the number of samples (%s)." % (len(weights), N_local)
        local_norm = float(weights.sum())
    else:
        local_norm = float(N_local)
    K = len(density)

    if latent is None:
        if mincount > 0:
            raise ValueError('`mincount` must be 0 if `latent` is not provided!')
        if not rb:
            raise ValueError('`rb` must be True if `latent` is not provided!')
        count = None
    elif on_device(latent):
        import torch
        count = torch.bincount(latent, minlength=K)[:K].cpu().numpy().astype(np.float64)
    else:
        latent = np.asarray(latent)
        count = np.histogram(latent, bins=K, range=(0, K))[0].astype(np.float64)

    live_components = [k for k in range(K) if density.weights[k] != 0]
    stat_components = list(live_components)

    be = get_backend(backend)
    D = density.dim
    tail0 = np.concatenate(([local_norm], count if count is not None else np.zeros(0)))

    def exchange(flat):
        """ONE exchange: statistics | weight normalisation | latent histogram travel in the same buffer
        (on the device when the statistics are, so RCCL reduces it in place)"""
        nstat = int(flat.shape[0])
        if parallel.active():
            joined = be.zeros(nstat + len(tail0))
            joined[:nstat] = flat
            joined[nstat:] = be.asdevice(tail0)
            joined = be.tohost(parallel.all_reduce_sum(joined))
            return joined[:nstat], joined[nstat:]
        return be.tohost(flat), tail0

    cs = None
    blocks = bool(live_components) and not rb and count is not None and int(count.sum()) == N_local \
        and _is_sorted(latent)
    if not rb and parallel.active():
        # which form a rank would take depends on its own shard (an empty one counts as sorted); the general form
        # may answer with a second exchange (far shift, below), so the ranks must agree: blocks only if all can
        blocks = parallel.all_reduce_scalars(0.0 if blocks else 1.0)[0] == 0.0
    if blocks:
        # non-Rao-Blackwell update of samples that arrive ordered by generating component (what
        # propose(trace=True) / run(trace_sort=True) deliver): component k only sees its own
        # contiguous block of samples, so the cost is N x D^2 instead of N x K x D^2 -- the
        # "faster" variant of pmc.pyx:153 is actually faster here too
        flat = _latent_blocks_estep(be, samples, weights, latent, density, live_components, count, K, D)
        nlive = len(live_components)
    elif live_components:
        cs = component_set(density.components, density.weights, live_components, K)
        if cs is None:
            raise TypeError('``density`` must have only Gauss or only StudentT components')
        mode = PMC_RESP_PMC_RB if rb else PMC_RESP_PMC_LATENT
        # dead components' all-zero columns take part in the row maximum (pmc.pyx:24-34)
        if responsibilities is not None and rb:
            # the weighting pass left u = w rho of these very samples, weights and parameters: statistics only
            full = component_set(density.components, density.weights)
            # (pruned components have no columns: the pass formed responsibilities for ITS live components, in order)
            why = 'density' if (list(getattr(responsibilities, 'live', range(K))) != list(live_components) or
                                responsibilities.N != N_local) else responsibilities.mismatch(full, weights, samples)
            if why == 'samples':
                # the same density and weights but another storage behind ``samples`` (a copy of the run, a history that
                # was reallocated by a later append -- advice r4): nothing proves the values are stale, nothing proves
                # they are not; the update forms its responsibilities itself
                logger.info("``responsibilities`` belong to another sample array: recomputed")
                res = be.estep(samples, cs, mode, max_init_zero=len(live_components) < K, sample_w=weights)
            elif why is not None:
                raise ValueError('``responsibilities`` were not formed with this density, these samples and weights')
            else:
                res = be.estep_from_u(samples, cs, responsibilities)
        elif mahalanobis is not None and rb:
            # the weighting pass kept maha_nk of these very samples: rho without a second evaluation
            full = component_set(density.components, density.weights)
            if mahalanobis.N != N_local or not mahalanobis.matches(full):
                raise ValueError('``mahalanobis`` was not computed with this density on these samples')
            res = be.estep_from_tiles(samples, cs, mahalanobis, max_init_zero=len(live_components) < K,
                                      sample_w=weights)
        else:
            res = be.estep(samples, cs, mode, max_init_zero=len(live_components) < K,
                           sample_w=weights, latent=None if rb else latent)
        flat = res["stats"]
        nlive = len(live_components)
    else:
        flat, nlive = be.zeros(be.stats_len(1, D)), 0

    flat, tail = exchange(flat)
    weight_normalization = float(tail[0])
    shift = np.array([density.components[k].mu for k in stat_components]).reshape(len(stat_components), D)
    # split / far-shift test / centring in one host call of the library (_stats.convert_stats: the numpy functions'
    # operations in their order); Student-t: the covariance is normalised by sum w rho, the mean by sum w rho gamma
    n_cov = 'vsum0' if student else None
    stats = convert_stats(flat, nlive, D, shift, n_cov) if nlive else None
    if nlive and stats[5]:
        # a weighted mean far from its proposal component (the first iterations of a badly placed proposal): the
        # one-pass moments about mu_k would cancel; second pass about the mean just found -- the reference's own
        # order (mean first, then the covariance about it: pmc.pyx:188-222, :612-632).  Decided on the all-reduced
        # sums, so every rank takes the same branch.
        S0 = stats[1]
        shift = np.where((S0 > 1e-200)[:, None], shift + stats[2] / regularize(S0.copy())[:, None], shift)
        if cs is None:
            again = _latent_blocks_estep(be, samples, weights, latent, density, live_components, count, K, D, shift)
        else:
            again = be.estep(samples, cs, mode, max_init_zero=len(live_components) < K, sample_w=weights,
                             latent=None if rb else latent, shift=shift)["stats"]
        flat, tail = exchange(again)
        stats = convert_stats(flat, nlive, D, shift, n_cov)

    if count is not None:
        count = tail[1:]
        # prune components that generated fewer than ``mincount`` samples -- AFTER rho was
        # computed; the list is edited while it is iterated exactly as in pmc.pyx:110-116, which
        # skips the element following a removed one
        for k in live_components:
            if count[k] < mincount:
                live_components.remove(k)
                density.weights[k] = 0.
                need_renormalize = True
                logger.warning("Component %i died because of too few (%i) samples." % (k, count[k]))

    return density, live_components, stat_components, stats, weight_normalization, need_renormalize, shift


def _sharded_update(samples, density, weights, latent, rb, mincount, copy, student):
    """``samples`` is a ``pypmc_amd.devices.ShardedSamples``: the N-sized part of the update runs on all devices of its
    group (pmc_pmc_update_stats: every device its block, the statistics added in device order, the far-shift second
    pass and the reference's normalisations inside); what is left here is the K-sized bookkeeping of pmc.pyx:53-118.
    ``latent``: None, a host array, or 'origin' (the generating components kept with generated samples).
    Returns density, live components, dict k -> (alpha, mu, sigma, dof constant), renormalise?"""
    need_renormalize = False
    if copy:
        density = deepcopy(density)
    K = len(density)
    use_origin = isinstance(latent, str) and latent == 'origin'
    if latent is None:
        if mincount > 0:
            raise ValueError('`mincount` must be 0 if `latent` is not provided!')
        if not rb:
            raise ValueError('`rb` must be True if `latent` is not provided!')
        count = None
    elif use_origin:
        if samples.counts is None:
            raise ValueError("latent='origin' needs samples a DeviceGroup generated")
        count = samples.counts.astype(np.float64)
    else:
        latent = np.asarray(latent)
        count = np.histogram(latent, bins=K, range=(0, K))[0].astype(np.float64)
    if weights is not None and not hasattr(weights, 'samples'):
        weights = np.asarray(weights)
        assert len(weights.shape) == 1, 'Weights must be one-dimensional.'
        assert len(weights) == len(samples), \
            "Number of weights (%s) does not match the number of samples (%s)." % (len(weights), len(samples))
    live_components = [k for k in range(K) if density.weights[k] != 0]
    stat_components = list(live_components)
    res = samples.group.pmc_update_stats(density, samples, weights, latent, rb)
    if count is not None:
        for k in live_components:                                  # (edited while iterated, as pmc.pyx:110-116)
            if count[k] < mincount:
                live_components.remove(k)
                density.weights[k] = 0.
                need_renormalize = True
                logger.warning("Component %i died because of too few (%i) samples." % (k, count[k]))
    return density, live_components, stat_components, res, need_renormalize


def _is_sorted(a):
    """non-decreasing?  (numpy array or device tensor)"""
    if len(a) < 2:
        return True
    return bool((a[1:] >= a[:-1]).all())


def _latent_blocks_estep(be, samples, weights, latent, density, live_components, count, K, D, shift=None):
    """Statistics of the latent (one-hot) responsibilities when the samples are ordered by
    component: one single-component pass per live component over its own block.  Returns the same
    flat layout as ``backend.estep`` for the live components.  ``shift`` (live x D): moments about these
    points instead of the components' means (the second pass of a far shift)."""
    from .._lib import NSCALARS
    x = be.asdevice(samples)
    lat = be.asdevice(latent, getattr(getattr(be, 'torch', None), 'int64', None))
    w = be.asdevice(weights) if weights is not None else None
    nlive = len(live_components)
    ps = 1 + D + D * (D + 1) // 2
    flat = be.zeros(be.stats_len(nlive, D))
    offsets = np.concatenate(([0], np.cumsum(count))).astype(np.int64)
    for i, k in enumerate(live_components):
        a, b = int(offsets[k]), int(offsets[k + 1])
        if a == b:
            continue
        cs = component_set(density.components, density.weights, [k], K)
        if cs is None:
            raise TypeError('``density`` must have only Gauss or only StudentT components')
        one = be.estep(x[a:b], cs, PMC_RESP_PMC_LATENT, sample_w=None if w is None else w[a:b],
                       latent=lat[a:b], shift=None if shift is None else shift[i:i + 1])["stats"]
        flat[NSCALARS + i * ps:NSCALARS + (i + 1) * ps] = one[NSCALARS:NSCALARS + ps]
        flat[NSCALARS + nlive * ps + 2 * i:NSCALARS + nlive * ps + 2 * i + 2] = one[NSCALARS + ps:NSCALARS + ps + 2]
    return flat


def _linalg_backend(density, sig):
    """the GPU backend, if the batch of factorisations is large enough to go there (K = 128, D = 40: 1.4 ms of LAPACK
    against 0.5 ms for upload + kernel + download); None: LAPACK on the host"""
    try:
        be = get_backend(getattr(density.components[0], '_backend', None))
    except Exception:
        return None
    if not hasattr(be, 'chol_inv_det_batch') or sig.size < getattr(be, 'DEVICE_LINALG_FROM', 1 << 62):
        return None
    return be


def _apply_updates(density, live_components, new_params, need_renormalize, stacked=None):
    """``component.update`` with the reference's fall-back: a LinAlgError restores the old
    parameters and zeroes the component's weight (pmc.pyx:227-244, :713-737).

    The K factorisations are done as one batch first (tools._linalg.chol_inv_det_batch); only if one of
    them fails does the update go component by component, so that exactly the failing ones fall back."""
    with single_threaded_blas():                  # K small factorisations: thread pool = overhead
        live = list(live_components)
        batch = None
        if len(live) > 1:
            try:
                if stacked is not None and list(stacked[0]) == live and stacked[1].dtype == np.float64 and stacked[1].ndim == 3:
                    sig = stacked[1]                              # (the statistics' covariances are one array already)
                else:
                    sig = np.array([np.asarray(new_params[k][1][1], dtype=np.float64) for k in live])
                if sig.ndim == 3 and all(float(new_params[k][1][2]) > 0. for k in live if len(new_params[k][1]) == 3):
                    done = None
                    be = _linalg_backend(density, sig)
                    if be is not None:
                        done = be.chol_inv_det_batch(sig)              # on the device (K D^2 large: backend.DEVICE_LINALG_FROM)
                    if done is None:
                        done = chol_inv_det_batch(sig, check_symmetric=False)   # centred_moments mirrors
                    batch = (sig,) + tuple(done)
            except np.linalg.LinAlgError:
                batch = None
        for i, k in enumerate(live):
            component = density.components[k]
            alpha_k, args = new_params[k]
            density.weights[k] = alpha_k
            if batch is not None:
                mu = np.array(args[0], dtype=float).reshape(-1)
                component._assign(mu, batch[0][i].copy(), batch[1][i], batch[2][i], float(batch[3][i]), *args[2:])
                continue
            old = (component.mu, component.sigma) + ((component.dof,) if len(args) == 3 else ())
            try:
                component.update(*args)
            except np.linalg.LinAlgError:
                logger.warning("Could not update component %i --> weight is set to zero." % k)
                component.update(*old)
                density.weights[k] = 0.
                need_renormalize = True
        if batch is not None and len(live) == len(density.components) and live == list(range(len(live))):
            # the factors and inverses as they are: one array each, the components' own arrays their rows
            from ..density.mixture import register_stacked
            register_stacked(density.components, batch[1], batch[2])
    if need_renormalize:
        density.normalize()
    return density


def gaussian_pmc(samples, density, weights=None, latent=None, rb=True, mincount=0, copy=True,
                 backend=None, mahalanobis=None, responsibilities=None):
    """Adapt a Gaussian mixture ``density`` to the (weighted) ``samples`` it proposed
    (reference: pmc.pyx:120-246, same signature and semantics).

    ``mahalanobis`` (extension): what ``ImportanceSampler.run_device(..., keep_mahalanobis=True)`` returned for
    these samples -- the Rao-Blackwellised update then reuses the Mahalanobis forms of the weighting pass
    instead of evaluating the proposal a second time, as the reference does (pmc.pyx:23-43).
    ``responsibilities`` (extension): what ``run_device(..., prepare_update=True)`` returned -- u = w rho [gamma] left
    behind by the weighting pass itself; the update is then the statistics kernel alone.  Both refuse any other
    density, sample set or weights than the ones they were formed with (``ValueError``)."""
    assert samples is not None
    if hasattr(samples, 'group') and hasattr(samples, 'shards'):      # ShardedSamples of a DeviceGroup (several GPUs)
        density, live, _, res, renorm = _sharded_update(samples, density, weights, latent, rb, mincount, copy, False)
        new = {k: (res["alpha"][k], (res["mu"][k], res["sigma"][k])) for k in live}
        return _apply_updates(density, live, new, renorm)
    if isinstance(samples, np.ndarray):          # device-resident tensors pass through untouched
        samples = np.ascontiguousarray(samples, dtype=np.float64)
    density, live, stat_comps, stats, norm, renorm, shift = \
        _prepare_pmc_update(samples, weights, latent, mincount, density, rb, copy, backend, mahalanobis,
                            responsibilities)
    if stat_comps:
        _, S0, _, mu, cov, _, _, _ = stats                       # mu, cov: pmc.pyx:194-204 / :213-222 (_stats.centred_moments)
        alpha = S0 / norm                                         # :191-193
        pos = {k: i for i, k in enumerate(stat_comps)}
        new = {k: (alpha[pos[k]], (mu[pos[k]], cov[pos[k]])) for k in live}
        return _apply_updates(density, live, new, renorm, stacked=(stat_comps, cov))
    return _apply_updates(density, live, {}, renorm)


DOF_BATCH_FROM = 16        # live components from which student_t_pmc solves all degree-of-freedom conditions at once


def _dof_condition(const):
    """First-order condition for nu, [HOD12] eq. (16): const + log(nu/2) - psi(nu/2) = 0
    (reference: pmc.pyx:478-497)."""
    return lambda nu: const + np.log(.5 * nu) - digamma(.5 * nu)


def _trigamma(x):
    """psi'(x), x > 0, to ~1e-12 (the slope of a Newton step, not a result): six steps of psi'(x) = psi'(x + 1) + 1 / x^2,
    then the asymptotic series (scipy's polygamma goes through a zeta function: 33 us per 128 values)"""
    x = np.asarray(x, dtype=np.float64)
    acc = np.zeros_like(x)
    for _ in range(6):
        acc += 1. / (x * x)
        x = x + 1.
    i = 1. / x
    i2 = i * i
    return acc + i * (1. + i * (.5 + i * (1. / 6. + i2 * (-1. / 30. + i2 * (1. / 42. + i2 * (-1. / 30.))))))


def _solve_dofs(const, mindof, maxdof, start=None):
    """The roots of ``_dof_condition(const[i])`` in [mindof, maxdof] for all i at once (round 6): safeguarded Newton steps in
    log nu on the whole vector instead of one ``brentq`` per component (K = 128: 3.2 ms of scalar root findings -> 0.2 ms).
    The condition decreases in nu (1 / nu < psi'(nu / 2) / 2), so a sign at an end decides a clamp exactly as the reference's
    handling of brentq's ValueError does (pmc.pyx:700-710).  Roots are iterated until the step is below 1e-15 of log nu;
    brentq stops within its xtol = 2e-12 of the root -- the two agree to 2e-12 absolutely (and to the conditioning of the root,
    1e-9 at nu = 1e3, where the condition is flat).  Returns None if a constant is not finite or the iteration does not settle
    (the caller's loop and its error handling take over)."""
    const = np.asarray(const, dtype=np.float64)
    if not np.isfinite(const).all():
        return None
    f = lambda nu: const + np.log(.5 * nu) - digamma(.5 * nu)
    lo, hi = np.full(const.shape, float(mindof)), np.full(const.shape, float(maxdof))
    f_lo, f_hi = f(lo), f(hi)
    if not (np.isfinite(f_lo).all() and np.isfinite(f_hi).all()):
        return None
    aconst = np.abs(const)
    below, above = f_lo < 0., f_hi > 0.                    # (no root inside: the condition is negative / positive throughout)
    work = ~(below | above)
    exact_lo, exact_hi = work & (f_lo == 0.), work & (f_hi == 0.)
    t_lo, t_hi = np.log(lo), np.log(hi)
    t = .5 * (t_lo + t_hi)
    if start is not None:                                  # (the previous iteration's dof: a few steps from the new one)
        ts = np.log(np.asarray(start, dtype=np.float64))
        t = np.where(np.isfinite(ts) & (ts > t_lo) & (ts < t_hi), ts, t)
    settled = ~work
    for _ in range(80):
        nu = np.exp(t)
        lg, ps = np.log(.5 * nu), digamma(.5 * nu)
        ft = const + lg - ps
        noise = 4e-16 * (aconst + np.abs(lg) + np.abs(ps))  # the condition is zero to the rounding of its three terms
        pos = ft > 0.                                      # the condition decreases: a positive value moves the lower end up
        t_lo = np.where(work & pos, t, t_lo)
        t_hi = np.where(work & ~pos, t, t_hi)
        dfdt = 1. - .5 * nu * _trigamma(.5 * nu)          # d/dt of log(nu/2) - psi(nu/2) with nu = e^t: negative
        t_new = t - np.where(dfdt < 0., ft / np.where(dfdt < 0., dfdt, -1.), 0.)
        outside = ~((t_new > t_lo) & (t_new < t_hi)) | (dfdt >= 0.)
        t_new = np.where(outside, .5 * (t_lo + t_hi), t_new)
        settled = settled | (np.abs(ft) <= noise) | (np.abs(t_new - t) <= 1e-15 * np.maximum(1., np.abs(t))) | \
            (t_hi - t_lo <= 2e-16 * np.maximum(1., np.abs(t_lo)))
        t = np.where(settled, t, t_new)
        if settled.all():
            break
    else:
        return None
    dof = np.exp(t)
    dof = np.where(below | exact_lo, mindof, dof)
    dof = np.where(above | exact_hi, maxdof, dof)
    return np.clip(dof, mindof, maxdof)


def student_t_pmc(samples, density, weights=None, latent=None, rb=True, dof_solver_steps=100,
                  mindof=1e-5, maxdof=1e3, mincount=0, copy=True, backend=None, mahalanobis=None,
                  responsibilities=None):
    """Adapt a Student-t mixture ``density`` (means, covariances and -- unless
    ``dof_solver_steps`` is 0 -- degrees of freedom) to the (weighted) ``samples`` it proposed
    (reference: pmc.pyx:499-739, same signature and semantics; ``mahalanobis``, ``responsibilities``: see
    ``gaussian_pmc``)."""
    assert samples is not None
    sharded = hasattr(samples, 'group') and hasattr(samples, 'shards')
    if sharded:
        density, live, stat_comps, res, renorm = _sharded_update(samples, density, weights, latent, rb, mincount, copy, True)
        stats = None
    else:
        if isinstance(samples, np.ndarray):
            samples = np.ascontiguousarray(samples, dtype=np.float64)
        density, live, stat_comps, stats, norm, renorm, shift = \
            _prepare_pmc_update(samples, weights, latent, mincount, density, rb, copy, backend, mahalanobis,
                                responsibilities, student=True)
    D = density.dim
    new = {}
    if sharded and stat_comps:
        # the devices' sums were turned into alpha, mu, sigma and the dof constant by the library (pmc_pmc_update_stats)
        alpha = res["alpha"][stat_comps]
        mu, cov = res["mu"][stat_comps], res["sigma"][stat_comps]
        const = res["dof_const"][stat_comps]
    if stat_comps:
        # S0g = sum w rho gamma, V1 = sum w rho; mean: normalised by sum w rho gamma; covariance: by sum w rho
        # (pmc.pyx:620-630; _stats.centred_moments with S0_cov = V1)
        if not sharded:
            _, S0g, _, mu, cov, _, V1, V2 = stats
            alpha = V1 / norm
        old_dof = np.array([density.components[k].dof for k in stat_comps])
        if dof_solver_steps and not sharded:
            # sum_n w_n (xi + delta)_nk of pmc.pyx:659-679 assembled from the device sums:
            #   rho (log(.5(b+nu)) - psi(.5(D+nu)))      -> V2 - psi(.5(D+nu)) V1
            #   (1-rho)(log(.5 nu) - psi(.5 nu))          -> (W - V1)(log(.5 nu) - psi(.5 nu))
            #   rho (D+nu)/(b+nu)                         -> S0g
            #   (1-rho)                                   -> W - V1
            total = V2 - digamma(.5 * (D + old_dof)) * V1 \
                + (norm - V1) * (np.log(.5 * old_dof) - digamma(.5 * old_dof)) + S0g + (norm - V1)
            const = 1. - total / norm
        pos = {k: i for i, k in enumerate(stat_comps)}
        # many components: all roots at once (brentq always converges within 50 steps on this range, so a cap of 50 or
        # more never takes its "not converged" branch; below that, and for a handful of components, the reference's loop)
        batch_dof = _solve_dofs(const, mindof, maxdof, old_dof) if (dof_solver_steps >= 50 and len(live) >= DOF_BATCH_FROM) else None
        for k in live:
            i = pos[k]
            if batch_dof is not None:
                new[k] = (alpha[i], (mu[i], cov[i], float(batch_dof[i])))
                continue
            if dof_solver_steps:
                condition = _dof_condition(const[i])
                try:
                    dof = brentq(condition, mindof, maxdof, maxiter=dof_solver_steps)
                except RuntimeError:                      # not converged
                    logger.warning("``dof`` solver for component %i did not converge." % k)
                    dof = density.components[k].dof
                except ValueError as error:
                    # same sign at both ends; the condition decreases with nu (pmc.pyx:700-710)
                    if condition(mindof) < 0.:
                        dof = mindof
                    elif condition(maxdof) > 0.:
                        dof = maxdof
                    else:
                        raise RuntimeError('``dof`` adaptation for component %i raised an error.' % k, error)
            else:
                dof = density.components[k].dof
            new[k] = (alpha[i], (mu[i], cov[i], dof))
    return _apply_updates(density, live, new, renorm, stacked=(stat_comps, np.asarray(cov)) if stat_comps else None)


class PMC(object):
    """EM driver: repeated PMC updates on the same samples until the log-likelihood converges
    (reference: pmc.pyx:248-476, same constructor and ``run``)."""

    def __init__(self, samples, density, weights=None, latent=None, rb=True, mincount=0,
                 backend=None, **kwargs):
        assert samples is not None
        self._backend = backend
        if weights is not None:
            self.weights = np.asarray(weights)
            assert len(self.weights.shape) == 1, 'Weights must be one-dimensional.'
            assert len(self.weights) == len(samples), \
                "Number of weights (%s) does not match the number of samples (%s)." % (len(weights), len(samples))
        else:
            self.weights = None
        if latent is None:
            if mincount > 0:
                raise ValueError('`mincount` must be 0 if `latent` is not provided!')
            if not rb:
                raise ValueError('`rb` must be True if `latent` is not provided!')
        wrong = '``density`` must be a ``pypmc_amd.density.mixture.MixtureDensity`` with ' \
                '``pypmc_amd.density.gauss.Gauss`` or ``pypmc_amd.density.student_t.StudentT`` components'
        if not isinstance(density, MixtureDensity):
            raise TypeError(wrong)
        first = type(density.components[0])
        if issubclass(first, StudentT):
            self.pmc, required = student_t_pmc, StudentT
        elif issubclass(first, Gauss):
            self.pmc, required = gaussian_pmc, Gauss
        else:
            raise TypeError(wrong)
        for c in density.components:
            if not isinstance(c, required) or (required is Gauss and isinstance(c, StudentT)):
                raise TypeError(wrong)
        self.density = deepcopy(density)
        self.samples = np.ascontiguousarray(samples, dtype=np.float64)
        self.latent = latent
        self.rb = rb
        self.mincount = mincount
        self.additional_args = kwargs
        # global normalisation of the importance weights (sum over all ranks)
        local = self.weights.sum() if self.weights is not None else float(len(self.samples))
        self._norm = parallel.all_reduce_scalars(local)[0]
        if self.weights is not None:
            self.normalized_weights = self.weights / self._norm
        be = get_backend(self._backend)
        self._samples_dev = be.asdevice(self.samples)
        # (the weights too: every update and every log-likelihood of run() would upload the N numbers again)
        self._weights_dev = be.asdevice(self.weights) if self.weights is not None and hasattr(be, 'torch') else self.weights

    def log_likelihood(self):
        """sum_n wbar_n log q(x_n), eq. (5) of [Cap+08] (reference: pmc.pyx:367-391); the sum is
        reduced on the device by the log-pdf kernel."""
        be = get_backend(self._backend)
        cs = component_set(self.density.components, self.density.weights)
        res = be.logpdf(self._samples_dev, cs, want_out=False, sample_w=self._weights_dev, want_scalars=True)
        total = be.tohost(parallel.all_reduce_sum(res["scalars"]))[3]
        return float(total / self._norm)

    def run(self, iterations=1000, prune=0., rel_tol=1e-10, abs_tol=1e-5, verbose=False):
        """Iterate updates until the log-likelihood converges; returns the iteration count or
        None (reference: pmc.pyx:393-476 -- same convergence rules)."""
        old_K = None
        bound = None
        for i in range(1, iterations + 1):
            if old_K == len(self.density):
                old_bound = bound
            else:
                old_bound = self.log_likelihood()
                logger.info('New bound=%g, K=%i' % (old_bound, len(self.density)))
            self.pmc(self._samples_dev, self.density,
                     self._weights_dev, self.latent, self.rb, mincount=self.mincount, copy=False,
                     backend=self._backend, **self.additional_args)
            bound = self.log_likelihood()
            if logger.isEnabledFor(logging.INFO):               # (formatting K weights costs more than a K-sized kernel)
                logger.info('After update %d: bound=%.15g, K=%i, component_weights=%s'
                            % (i, bound, len(self.density), self.density.weights))
            if bound < old_bound:
                logger.warning('Bound decreased from %g to %g' % (old_bound, bound))
            if bound == old_bound:
                return i
            diff = bound - old_bound
            if diff > 0:
                if abs(bound) < abs_tol:
                    if abs(diff) < abs_tol:
                        return i
                elif abs(diff / bound) < rel_tol:
                    return i
            old_K = len(self.density)                       # K *before* pruning
            self.density.prune(prune)
            self.density.normalize()
        return None
