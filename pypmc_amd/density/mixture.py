"""Mixture densities with the fused log-pdf / log-sum-exp kernel behind them
(reference: pypmc/density/mixture.pyx)."""
from copy import deepcopy

from collections import OrderedDict

import numpy as np

from .base import ProbabilityDensity
from .gauss import Gauss
from .student_t import StudentT
from ..backend import ComponentSet, get_backend


_SETS = OrderedDict()        # the last few ComponentSets built, with the parameter pack the backend uploaded for them
_SETS_MAX = 8


# What a batched K-sized update left behind (mix_adapt.pmc._apply_updates): the K Cholesky factors and inverses as ONE array
# each, of which the components' own arrays are the rows.  The next pack / the next device-side propose take those arrays
# instead of gathering 2 x K x D x D numbers out of K objects again (K = 128, D = 40: 0.2-0.3 ms each).  Keyed by the components'
# parameter stamps in order; used only while every component's array still IS its row (``.base is`` the stacked array).
_STACKED = OrderedDict()
_STACKED_MAX = 4


def register_stacked(components, cholesky, inverse):
    _STACKED[tuple(c._stamp for c in components)] = (cholesky, inverse)
    while len(_STACKED) > _STACKED_MAX:
        _STACKED.popitem(last=False)


def _stacked(components, attribute):
    """the stacked array behind ``attribute`` ('cholesky_sigma' / 'inv_sigma') of these components, or None"""
    hit = _STACKED.get(tuple(c._stamp for c in components))
    if hit is None:
        return None
    arr = hit[0 if attribute == 'cholesky_sigma' else 1]
    owner = arr if arr.base is None else arr.base           # (numpy names the owning array as the base of a row)
    if len(arr) != len(components) or any(getattr(c, attribute).base is not owner for c in components):
        return None
    return arr


def component_set(components, weights, columns=None, ld=None):
    """ComponentSet of homogeneous Gauss / StudentT ``components`` (None if they are of another
    or of mixed type).  ``columns`` selects a subset, ``ld`` is the total component count.

    Evaluating the same mixture again -- every ``ImportanceSampler.run`` between two proposal updates, every
    ``multi_evaluate`` of a target mixture -- finds its set here and with it the pack already on the device
    (K Cholesky factors of the precision matrices and an upload otherwise: 0.6 ms at K = 128, D = 40).  The key
    is the components' parameter stamps (``update`` renews them; they are process-local and renewed when a
    component is unpickled), the means and the weights themselves: both are plain arrays the reference lets
    callers change in place.  Only complete mixtures are kept: subsets (``components=``, the live components of
    a PMC update, the single-component sets of the latent-blocks update) come and go with every call and would
    only push the sets worth keeping out.  ``clear_component_cache()`` (also called by ``HipBackend.release``)
    drops the sets and with them the device packs."""
    comps = list(components)
    if not comps:
        return None
    first = type(comps[0])
    if first not in (Gauss, StudentT) or any(type(c) is not first for c in comps):
        return None
    idx = list(range(len(comps))) if columns is None else list(columns)
    if columns is not None and idx == list(range(len(comps))) and ld in (None, len(comps)):
        columns = None                                       # every component, in order: the complete mixture (cached)
    sel = [comps[k] for k in idx]
    mu = np.array([c.mu for c in sel], dtype=np.float64).reshape(len(sel), -1)
    wts = np.asarray(weights, dtype=np.float64)[idx]
    total = len(comps) if ld is None else ld
    cacheable = columns is None and total == len(comps)
    if cacheable:
        key = (first, tuple(c._stamp for c in sel), mu.tobytes(), wts.tobytes(), total)
        hit = _SETS.get(key)
        if hit is not None:
            _SETS.move_to_end(key)
            return hit
    consts = np.array([c._kernel_constants() for c in sel], dtype=np.float64).reshape(len(sel), 4)
    inv = _stacked(sel, 'inv_sigma') if cacheable else None
    cs = ComponentSet(first.kind, mu, inv if inv is not None else np.array([c.inv_sigma for c in sel], dtype=np.float64),
                      consts[:, 0], consts[:, 1], consts[:, 2], consts[:, 3],
                      weight=wts, column=idx, ld=total)
    if cacheable:
        _SETS[key] = cs
        if len(_SETS) > _SETS_MAX:
            _SETS.popitem(last=False)
    return cs


def clear_component_cache():
    """Forget the cached ComponentSets (and the device parameter packs kept with them)."""
    _SETS.clear()
    _STACKED.clear()


class MixtureDensity(ProbabilityDensity):
    """sum_k w_k q_k(x) over component densities q_k (reference: mixture.pyx:21-59).

    ``components`` are deep-copied, ``weights`` normalised.  Host state is authoritative; the
    device parameter pack is rebuilt from it for every evaluation."""

    def __init__(self, components, weights=None, backend=None):
        self._backend = backend
        self.components = [deepcopy(c) for c in components]
        assert self.components, "Must have at least one component!"
        self.dim = self.components[0].dim
        np.testing.assert_equal([c.dim for c in self.components], [self.dim] * len(self.components))
        if weights is None:
            self.weights = np.ones(len(self.components))
        else:
            self.weights = np.array(weights, dtype=float)
            assert len(self.weights) == len(self.components)
        self.normalize()

    def __len__(self):
        K = len(self.components)
        assert K == len(self.weights)
        return K

    def normalize(self):
        """Scale the weights to sum to one (in place)."""
        self.weights /= self.weights.sum()

    def normalized(self):
        return bool(np.allclose(self.weights.sum(), 1.0))

    def prune(self, threshold=0.0):
        """Drop components with weight <= ``threshold``; returns [(index, component, weight), ...]
        from the highest index down (reference: mixture.pyx:66-94)."""
        removed = []
        for k in range(len(self.weights) - 1, -1, -1):
            if self.weights[k] <= threshold:
                removed.append((k, self.components.pop(k), self.weights[k]))
        self.weights = np.delete(self.weights, [r[0] for r in removed])
        return removed

    # -- evaluation ---------------------------------------------------------------------------
    def evaluate(self, x, individual=False):
        """log q(x) of one point [and the component log-densities] (reference: mixture.pyx:101-110)."""
        x = np.asarray(x, dtype=np.float64).reshape(1, -1)
        ind = np.empty((1, len(self)))
        res = self.multi_evaluate(x, individual=ind)
        return (float(res[0]), ind[0]) if individual else float(res[0])

    def multi_evaluate(self, x, out=None, individual=None, components=None):
        """log q(x_n) for all rows of ``x`` (N x D); optionally the N x K matrix of component
        log-densities in ``individual``; with ``components`` only those columns of ``individual``
        are computed and nothing is returned (reference: mixture.pyx:112-156, same assertions)."""
        assert x.shape[1] == self.dim, \
            "The points in ``x`` have the wrong dimension (%i instead of %i)" % (x.shape[1], self.dim)
        N, K = len(x), len(self)
        if individual is not None:
            assert len(x) == len(individual), \
                "For the provided ``x``, ``individual`` must have shape %s" % ((N, K),)
            assert individual.shape[1] == K, \
                "For the provided ``x``, ``individual`` must have shape %s" % ((N, K),)
        be = get_backend(self._backend)
        x = np.ascontiguousarray(x, dtype=np.float64)

        if components is not None:
            assert out is None, 'If ``components`` is not None, ``out`` must be None.'
            components = list(components)
            if individual is None:
                individual = np.empty((N, K))       # reference allocates and discards it too
            cs = component_set(self.components, self.weights, components, K)
            if cs is not None and components:
                dev = be.zeros((N, K))
                be.logpdf(x, cs, want_out=False, individual=dev)
                individual[:, components] = be.tohost(dev)[:, components]
            else:
                for k in components:
                    self.components[k].multi_evaluate(x, individual[:, k])
            return None

        if out is not None:
            assert len(out) == len(x), '``out`` must have length %i' % (len(x))
        cs = component_set(self.components, self.weights)
        if cs is not None:
            res = be.logpdf(x, cs, want_out=True, want_individual=individual is not None)
            if individual is not None:
                individual[:] = be.tohost(res["individual"])
            result = be.tohost(res["out"])
        else:
            # foreign component types: their own multi_evaluate, then the log-sum-exp kernel
            ind = individual if individual is not None else np.empty((N, K))
            for k, c in enumerate(self.components):
                c.multi_evaluate(x, ind[:, k])
            result = be.tohost(be.logsumexp2d(ind, self.weights))
        if out is None:
            return result
        out[:] = result
        return out

    # -- sampling -------------------------------------------------------------------------------
    def propose(self, N=1, rng=np.random.mtrand, trace=False, shuffle=True, device=False, out=None):
        """N samples.  Component counts come from ``rng.multinomial(N, weights)``; with
        ``trace`` the generating component of every sample is returned too (samples then stay
        ordered by component).  Reference: mixture.pyx:159-212 -- including its quirk that the
        components draw from their default generator, not from ``rng``.

        ``device=True`` (extension): the counts are still drawn on the host with ``rng`` (so counts
        and origins are bit-exact for a given generator state), the samples are generated on the
        GPU (pmc_propose, Philox stream seeded from ``rng``) and returned as device tensors --
        nothing N-sized crosses PCIe.  ``out``: optional N x dim device buffer to generate into
        (e.g. the run a DeviceHistory just opened)."""
        if trace and shuffle:
            raise ValueError('Either ``shuffle`` or ``trace`` must be ``False``!')
        counts = rng.multinomial(N, self.weights)
        if device:
            return self._propose_device(counts, rng, trace, shuffle, out=out)
        samples = np.empty((N, self.dim))
        start = 0
        for comp, n in zip(self.components, counts):
            if n != 0:
                samples[start:start + n] = comp.propose(n)
            start += n
        if trace:
            return samples, np.repeat(np.arange(len(self.components)), counts)
        if shuffle:
            rng.shuffle(samples)
        return samples

    def _propose_device(self, counts, rng, trace, shuffle, first_sample=0, out=None):
        comps = self.components
        first = type(comps[0])
        if first not in (Gauss, StudentT) or any(type(c) is not first for c in comps):
            raise TypeError('device-side propose needs only Gauss or only StudentT components')
        be = get_backend(self._backend)
        seed = int(rng.randint(0, 2 ** 31 - 1)) | (int(rng.randint(0, 2 ** 31 - 1)) << 32)
        chol = _stacked(comps, 'cholesky_sigma')
        x, origin = be.propose(np.array([c.mu for c in comps]),
                               chol if chol is not None else np.array([c.cholesky_sigma for c in comps]),
                               np.array([c.dof for c in comps]) if first is StudentT else None,
                               counts, seed, first_sample=first_sample, want_origin=trace,
                               out=None if (shuffle and not trace) else out)
        if trace:
            return x, origin
        if shuffle:
            import torch
            g = torch.Generator(device=x.device).manual_seed(seed & (2 ** 63 - 1))
            perm = torch.randperm(len(x), device=x.device, generator=g)
            x = torch.index_select(x, 0, perm, out=out) if out is not None else x[perm].contiguous()
        return x


def create_gaussian_mixture(means, covs, weights=None):
    """MixtureDensity of Gauss components (reference: mixture.pyx:214-250)."""
    assert len(means) == len(covs), \
        'Number of means (%i) does not match number of covariances (%i)' % (len(means), len(covs))
    return MixtureDensity([Gauss(m, c) for m, c in zip(means, covs)], weights)


def recover_gaussian_mixture(mixture):
    """(means, covs, weights) arrays of a Gaussian mixture (reference: mixture.pyx:252-282)."""
    means = np.array([c.mu for c in mixture.components]).reshape(len(mixture), mixture.dim)
    covs = np.array([c.sigma for c in mixture.components]).reshape(len(mixture), mixture.dim, mixture.dim)
    return means, covs, np.array(mixture.weights)


def create_t_mixture(means, covs, dofs, weights=None):
    """MixtureDensity of StudentT components (reference: mixture.pyx:284-323)."""
    assert (len(means) == len(covs)) and (len(means) == len(dofs)), \
        'Number of ``means`` (%i), ``covs`` (%i) and ``dofs`` (%i) do not match.' % (len(means), len(covs), len(dofs))
    return MixtureDensity([StudentT(m, c, d) for m, c, d in zip(means, covs, dofs)], weights)


def recover_t_mixture(mixture):
    """(means, covs, dofs, weights) arrays of a Student-t mixture (reference: mixture.pyx:325-350)."""
    means, covs, weights = recover_gaussian_mixture(mixture)
    return means, covs, np.array([c.dof for c in mixture.components], dtype=float), weights
