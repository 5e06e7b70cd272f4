"""Importance sampling with the proposal density and the weights evaluated on the GPU
(reference: pypmc/sampler/importance_sampling.py)."""
from copy import deepcopy

import numpy as np

from ..backend import get_backend
from ..density.base import ProbabilityDensity
from ..tools._history import History, DeviceHistory
from ..tools.indicator import merge_function_with_indicator


def calculate_expectation(samples, weights, f):
    """sum_n w_n f(x_n) / sum_n w_n  (reference: importance_sampling.py:13-44); host loop over
    the user's callable."""
    assert len(samples) == len(weights), \
        "The number of samples (got %i) must equal the number of weights (got %i)." % (len(samples), len(weights))
    total, norm = 0., 0.
    for w, x in zip(weights, samples):
        norm += w
        total += w * f(x)
    return total / norm


def _on_device(a):
    """True for a device tensor (anything array-like that is not a numpy array / sequence)."""
    return hasattr(a, 'device') and not isinstance(a, np.ndarray)


def _sharded(samples):
    from ..devices import ShardedSamples
    return isinstance(samples, ShardedSamples)


def _moments(samples, weights, backend):
    assert len(samples) == len(weights), \
        "The number of samples (got %i) must equal the number of weights (got %i)." % (len(samples), len(weights))
    be = get_backend(backend)
    x = samples if _on_device(samples) else np.ascontiguousarray(samples, dtype=np.float64)
    return be.weighted_moments(x, weights)


def calculate_mean(samples, weights, backend=None):
    """Weighted sample mean (reference: importance_sampling.py:46-61), reduced on the GPU by the
    statistics kernel."""
    if _sharded(samples):                     # a DeviceGroup's sample set: every device its block, one ordered sum
        return samples.group.weighted_moments(samples, weights, want_cov=False)[0]
    S0, M1, _, shift, _ = _moments(samples, weights, backend)
    return shift + M1 / S0


def calculate_covariance(samples, weights, backend=None):
    """Weighted sample covariance with the (sum w)^2 / ((sum w)^2 - sum w^2) correction
    (reference: importance_sampling.py:63-83), reduced on the GPU by the statistics kernel."""
    if _sharded(samples):
        return samples.group.weighted_moments(samples, weights)[1]
    S0, M1, M2, _, Q = _moments(samples, weights, backend)
    dbar = M1 / S0
    return S0 * S0 / (S0 * S0 - Q) * (M2 / S0 - np.outer(dbar, dbar))


class ImportanceSampler(object):
    """Weighted samples of ``target`` (a callable returning log P(x)) drawn from ``proposal``
    (reference: importance_sampling.py:132-236; same constructor, attributes and ``run``).

    Per run the proposal's log-density, w = exp(log P - log q) and the three sums behind
    perplexity / ESS come from one fused kernel launch; the user's ``target`` stays a host
    callable evaluated per sample (or per batch, if it is the ``evaluate`` method of a density
    offering ``multi_evaluate``).

    ``device=True`` (extension): ``samples`` / ``weights`` / ``target_values`` are
    :class:`DeviceHistory` objects — ``run`` proposes on the GPU straight into the store, weights
    there, and only a target that is a host callable sees host copies of the samples.  Indexing
    the histories still yields (lazy) host arrays, ``.device(i)`` the device views.

    ``devices=[0, 1, 2, 3]`` (extension; a list of ordinals or a ``pypmc_amd.devices.DeviceGroup``): the GPUs of this
    node behind this ONE process.  ``run`` draws the component counts from ``rng`` on the host (bit-exact counts and
    origins), generates contiguous blocks of the samples on the devices, weights them there -- in one pass with a
    Gauss / Student-t mixture target, through a host copy of the samples with any other callable -- and records
    samples, weights and target values in the (host) histories as the reference does.  ``last_run`` is the sharded
    sample set with its weights still on the devices: ``gaussian_pmc(sampler.last_run, proposal,
    weights=sampler.last_run.weights)`` adapts the proposal without moving anything N-sized again.  The samples of a
    run arrive ordered by generating component (the weights do not depend on the order; the reference shuffles)."""

    def __init__(self, target, proposal, indicator=None, prealloc=0, save_target_values=False,
                 rng=np.random.mtrand, backend=None, device=False, devices=None):
        self._backend = backend
        self.device = bool(device)
        self._group = None
        self.last_run = None
        if devices is not None:
            if self.device:
                raise ValueError('``device=True`` (one GPU, device-resident histories) and ``devices=[...]`` (several GPUs '
                                 'behind this process) are two modes: choose one')
            from ..devices import DeviceGroup
            self._group = DeviceGroup.of(devices)
        self.proposal = deepcopy(proposal)
        self.rng = rng
        self._batch_target = None
        # batch shortcut only for this package's own densities: a foreign class's multi_evaluate
        # may mean something else, and the reference calls target(x) once per sample
        owner = getattr(target, '__self__', None)
        if indicator is None and isinstance(owner, ProbabilityDensity) \
                and getattr(target, '__name__', '') == 'evaluate' \
                and getattr(target, '__func__', None) is getattr(type(owner), 'evaluate', None):
            self._batch_target = owner.multi_evaluate
        self.target = merge_function_with_indicator(target, indicator, -np.inf)
        if self.device:
            def store(dim):
                return DeviceHistory(dim, prealloc, backend)
        else:
            def store(dim):
                return History(dim, prealloc)
        self.target_values = store(1) if save_target_values else None
        self.weights = store(1)
        self.samples = store(proposal.dim)
        self.last_weight_sums = None      # (sum w, sum w log w, sum w^2) of the latest run

    # ``last_run`` is a handle to sample buffers on the devices: a copy.deepcopy of the sampler shares it (ShardedSamples
    # copies by reference), a pickle leaves it behind -- the histories on the host hold the same samples and weights
    def __getstate__(self):
        state = dict(self.__dict__)
        state['last_run'] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        memo[id(self)] = new
        for key, value in self.__dict__.items():
            new.__dict__[key] = value if key in ('last_run', 'rng', '_group') else deepcopy(value, memo)
        return new

    def clear(self):
        """Forget samples, weights and target values; the proposal is untouched."""
        self.samples.clear()
        self.weights.clear()
        if self.target_values is not None:
            self.target_values.clear()

    def run(self, N=1, trace_sort=False):
        """Draw N samples and weight them.  With ``trace_sort`` (mixture proposals) the samples
        are ordered by generating component and that component index is returned per sample."""
        if N == 0:
            return 0
        if self._group is not None:
            return self._run_group(N, trace_sort)
        if self.device:
            res = self.run_device(N, trace_sort=trace_sort, store=True)
            return get_backend(self._backend).tohost(res["origin"]) if trace_sort else None
        if trace_sort:
            this_samples, origin = self._get_samples(N, trace_sort=True)
            self._calculate_weights(this_samples, N)
            return origin
        this_samples = self._get_samples(N, trace_sort=False)
        self._calculate_weights(this_samples, N)

    def _run_group(self, N, trace_sort, store=True):
        """one run over the devices of ``self._group`` (see the class docstring)"""
        from ..density.mixture import MixtureDensity
        g, prop = self._group, self.proposal
        if not isinstance(prop, MixtureDensity):
            raise TypeError('``devices=[...]`` needs a MixtureDensity proposal (Gauss or StudentT components)')
        counts = self.rng.multinomial(N, prop.weights)                     # mixture.pyx:192
        # the Philox seed of the device stream: two draws from the caller's generator -- ``randint`` of the legacy
        # numpy.random API (RandomState, the module itself: what the reference takes, importance_sampling.py:146) or
        # ``integers`` of a numpy Generator (advice r5).  This consumes the generator differently from the reference's
        # run(), which draws every sample from it: the streams of a device run and a host run are not comparable.
        draw = getattr(self.rng, 'integers', None) or self.rng.randint
        seed = int(draw(0, 2 ** 31 - 1)) | (int(draw(0, 2 ** 31 - 1)) << 32)
        run = g.generate(prop, counts, seed)
        tgt = getattr(self._batch_target, '__self__', None)
        mixture_target = isinstance(tgt, MixtureDensity)
        if mixture_target:
            try:
                g.mixture(tgt)
            except TypeError:
                mixture_target = False                                      # foreign component types: the host path
        host_x = run.host() if (store or not mixture_target) else None
        if mixture_target:
            res = g.importance_weights(prop, run, target=tgt, want_weights=store,
                                       want_log_target=store and self.target_values is not None)
        else:
            res = g.importance_weights(prop, run, log_target=self._target_values(host_x, N), want_weights=store)
        self.last_weight_sums = res["sums"]
        self.last_run = run
        if store:
            self.samples.append(N)[:] = host_x
            self.weights.append(N)[:, 0] = res["weights"]
            if self.target_values is not None:
                self.target_values.append(N)[:, 0] = res["log_target"]
        return np.repeat(np.arange(len(prop.components)), counts) if trace_sort else None

    def run_device(self, N, trace_sort=False, target_density=None, store=False, keep_mahalanobis=False,
                   prepare_update=False):
        """Extension for device-resident loops (BASELINE config 5): propose N samples ON THE GPU,
        weight them there and return ``dict(samples, weights, origin, weight_sums)`` of device
        tensors.  With a mixture target (``target_density``, default: the object whose ``evaluate``
        was given as ``target``) nothing N-sized crosses PCIe; any other target is called on a host
        copy of the samples and only its N log-values are uploaded.  ``store=True`` (needs
        ``device=True`` at construction) generates into / records in the DeviceHistory objects.
        ``keep_mahalanobis=True`` additionally returns ``mahalanobis``: the Mahalanobis forms of the samples
        under the proposal's components, kept on the device for ``gaussian_pmc / student_t_pmc(...,
        mahalanobis=...)`` (8 K bytes per sample).
        ``prepare_update=True`` (Gauss / Student-t mixture proposal and mixture target): the weighting pass also leaves the
        Rao-Blackwellised responsibilities of the update that follows, ``responsibilities`` in the result, for
        ``gaussian_pmc(samples, proposal, weights, ..., responsibilities=...)`` -- no responsibility kernel runs at
        all (likewise ``student_t_pmc``); where that form does not apply (dead components, D > 64) ``mahalanobis`` is
        returned instead."""
        from ..density.mixture import MixtureDensity, component_set
        be = get_backend(self._backend)
        if store and not self.device:
            raise ValueError('store=True needs ImportanceSampler(..., device=True)')
        tgt = target_density if target_density is not None else getattr(self._batch_target, '__self__', None)
        out = self.samples.append(N) if store else None
        origin = None
        if trace_sort:
            x, origin = self.proposal.propose(N, self.rng, trace=True, shuffle=False, device=True, out=out)
        else:
            x = self.proposal.propose(N, self.rng, device=True, out=out)
        prop_set = component_set(self.proposal.components, self.proposal.weights)
        if isinstance(tgt, MixtureDensity):
            # mixture target: log P, log q, the weights and the perplexity sums in one pass over x
            # prepare_update: the pass leaves u = w rho itself where that form applies, else it keeps the Mahalanobis
            # forms for the update -- decided BEFORE the pass (advice r3: it used to run twice in the second case)
            emit = prepare_update and not keep_mahalanobis and getattr(be, "can_emit", lambda c: True)(prop_set)
            res = be.importance_weights(x, prop_set, component_set(tgt.components, tgt.weights),
                                        want_log_target=store and self.target_values is not None,
                                        keep=keep_mahalanobis or (prepare_update and not emit), emit=emit)
            log_target = res["log_target"]
        else:
            log_target = be.asdevice(self._target_values(be.tohost(x), N))
            res = be.logpdf(x, prop_set, want_out=False, log_target=log_target, want_scalars=True,
                            keep=keep_mahalanobis)
        sc = be.tohost(res["scalars"])
        if sc[4] > 0:
            raise OverflowError('math range error')
        self.last_weight_sums = (float(sc[0]), float(sc[1]), float(sc[2]))
        if store:
            self.weights.append(N)[:, 0] = res["weights"]
            if self.target_values is not None:
                self.target_values.append(N)[:, 0] = log_target
        return dict(samples=x, weights=res["weights"], origin=origin, weight_sums=self.last_weight_sums,
                    mahalanobis=res.get("tiles"), responsibilities=res.get("responsibilities"))

    def _get_samples(self, N, trace_sort):
        this_run = self.samples.append(N)
        if trace_sort:
            this_run[:], origin = self.proposal.propose(N, self.rng, trace=True, shuffle=False)
            return this_run, origin
        this_run[:] = self.proposal.propose(N, self.rng)
        return this_run

    def _target_values(self, x, N):
        if self._batch_target is not None:
            return np.asarray(self._batch_target(x), dtype=np.float64).reshape(N)
        t = np.empty(N)
        for i in range(N):
            v = self.target(x[i])
            t[i] = v.item() if np.ndim(v) != 0 else v
        return t

    def _calculate_weights(self, this_samples, N):
        """w_n = exp(log P(x_n) - log q(x_n))  (reference: importance_sampling.py:197-215)."""
        this_weights = self.weights.append(N)[:, 0]
        log_target = self._target_values(this_samples, N)
        if self.target_values is not None:
            self.target_values.append(N)[:, 0] = log_target
        be = get_backend(self._backend)
        from ..density.mixture import MixtureDensity, component_set
        cs = None
        if isinstance(self.proposal, MixtureDensity):
            cs = component_set(self.proposal.components, self.proposal.weights)
        elif hasattr(self.proposal, '_component_set'):
            cs = self.proposal._component_set()
        if cs is not None:
            res = be.logpdf(np.ascontiguousarray(this_samples), cs, want_out=False,
                            log_target=log_target, want_scalars=True)
            sc = be.tohost(res["scalars"])
            if sc[4] > 0:
                raise OverflowError('math range error')      # math.exp, importance_sampling.py:207
            this_weights[:] = be.tohost(res["weights"])
            self.last_weight_sums = (float(sc[0]), float(sc[1]), float(sc[2]))
        else:
            # foreign proposal type: its own multi_evaluate, exponentiation on the host
            log_q = self.proposal.multi_evaluate(this_samples)
            with np.errstate(over='raise'):
                try:
                    this_weights[:] = np.exp(log_target - log_q)
                except FloatingPointError:
                    raise OverflowError('math range error')
            self.last_weight_sums = None


def _proposal_component_set(prop):
    from ..density.mixture import MixtureDensity, component_set
    if isinstance(prop, MixtureDensity):
        return component_set(prop.components, prop.weights)
    if hasattr(prop, '_component_set'):
        return prop._component_set()
    return None


def combine_weights(samples, weights, proposals, backend=None):
    """Deterministic-mixture weights [Cor+12] of T importance-sampling runs with different
    proposals (reference: importance_sampling.py:238-371).  Every log q_l(x^t_n) comes from the
    log-pdf kernel straight into row l of a T x N_t device matrix, and one launch of
    ``pmc_combine_weights`` per run turns it into the combined weights (log-scale branch if all
    weights are positive, else the linear branch).  Returns a History with one run per proposal;
    if any of the inputs is a device tensor (e.g. ``sampler.samples.device(i)``) everything stays
    on the GPU and the result is a DeviceHistory."""
    samples, weights = list(samples), list(weights)
    assert len(samples) == len(weights), \
        "Got %i importance-sampling runs but %i weights" % (len(samples), len(weights))
    assert len(samples) == len(proposals), \
        "Got %i importance-sampling runs but %i proposal densities" % (len(samples), len(proposals))
    T = len(proposals)
    counts = np.empty(T)
    on_device = any(_on_device(a) for a in samples + weights)
    for t in range(T):
        if not _on_device(samples[t]):
            samples[t] = np.asarray(samples[t])
        assert samples[t].ndim == 2, '``samples[%i]`` is not matrix like.' % t
        dim = samples[0].shape[-1]
        assert samples[t].shape[-1] == dim, \
            "Dimension of samples[0] (%i) does not match the dimension of samples[%i] (%i)" \
            % (dim, t, samples[t].shape[-1])
        counts[t] = len(samples[t])
        if not _on_device(weights[t]):
            weights[t] = np.asarray(weights[t])
        assert counts[t] == len(weights[t]), \
            'Length of weights[%i] (%i) does not match length of samples[%i] (%i)' \
            % (t, len(weights[t]), t, counts[t])
    n_total = int(counts.sum())
    be = get_backend(backend)
    combined = DeviceHistory(1, n_total, backend) if on_device else History(1, n_total)
    use_log = all(bool((w > 0.0).all()) for w in weights)
    nonfinite, total = 0., 0.
    for t in range(T):
        n_t = int(counts[t])
        x = be.asdevice(samples[t])
        q = be.empty((T, n_t))                    # q[l, n] = log q_l(x^t_n)
        host_x = None
        for l, prop in enumerate(proposals):
            cs = _proposal_component_set(prop)
            if cs is not None:
                be.logpdf(x, cs, out=q[l])
            else:                                 # foreign density type: its own multi_evaluate
                host_x = be.tohost(x) if host_x is None else host_x
                q[l] = be.asdevice(prop.multi_evaluate(host_x))
        w, flag = be.combine_weights(q, counts, t, be.asdevice(weights[t]).reshape(n_t), n_total, use_log)
        run = combined.append(n_t)
        run[:, 0] = w if on_device else be.tohost(w)
        nonfinite += float(be.tohost(flag)[0])
        total += float(be.tohost(be.weight_sums(w))[0])
    if use_log:
        assert total > 0, 'Sum of weights <=0 (%g)' % total
    assert nonfinite == 0, 'Encountered inf or nan mixture weights'
    return combined
