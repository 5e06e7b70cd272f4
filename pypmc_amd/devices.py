"""Several GPUs of one node behind ONE Python process: the sharded front-end over libpmc_hip's handle layer
(include/pmc_ctx.h, ``pmc_init_devices``).

pypmc's callers are single Python processes (pypmc/examples/pmc.py:53-73, examples/variational.py:54-61); its only
multi-process code gathers whole sample histories on one rank with mpi4py (pypmc/tools/parallel_sampler.py:58-66).
``DeviceGroup`` gives such a caller the GPUs of its node without a launcher: the library owns contiguous shards of the
sample array (device order = row order), runs every N-sized call on all devices at once -- one host thread, one stream and
one scratch set per device -- and adds the K-sized vectors in device order on the first device, so results are
bit-reproducible and equal to the ordered sum of the per-shard results.  No IPC, no RCCL, no ``torchrun``:

    GaussianInference(data, components=K, devices=[0, 1, 2, 3])
    ImportanceSampler(target.evaluate, proposal, devices=[0, 1, 2, 3])
    gaussian_pmc(sampler.last_run, proposal, weights=sampler.last_run.weights)

(``pypmc_amd.parallel`` -- one process per GPU under ``torchrun``, RCCL -- stays the way to span several nodes.)  The same
ordinal may be listed more than once: virtual shards on one GPU, which is how a one-GPU box tests and profiles this path.
Everything here is ctypes over the C ABI; arrays in and out are numpy (host) arrays in the reference's conventions.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import HipLibraryError, PMC_KIND_GAUSS, PMC_KIND_STUDENT_T

__all__ = ["DeviceGroup", "ShardedSamples", "ShardedMixture", "ShardedWeights"]

_dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
_ip = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_int64))
_c64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)


class ShardedWeights(object):
    """The importance weights a weighting pass left on the devices, next to the samples they belong to
    (``run.weights``): hand it to ``gaussian_pmc / student_t_pmc / calculate_mean(...)`` and nothing N-sized crosses
    the bus again.  ``host()`` copies them out."""

    def __init__(self, samples):
        self.samples = samples

    def host(self):
        return self.samples.host_weights()

    def __len__(self):
        return len(self.samples)


class ShardedSamples(object):
    """N x D samples resident on the devices of a ``DeviceGroup`` in contiguous blocks (``shards()``)."""

    def __init__(self, group, handle, N, D, counts=None):
        self.group, self._h, self.N, self.dim = group, handle, int(N), int(D)
        self.counts = None if counts is None else np.asarray(counts, dtype=np.int64)   # generated: per component
        self.has_weights = False
        self.has_sample_weights = False
        self._host_w = None
        group._live_samples.add(self)                      # (weakly: DeviceGroup.close() frees what is still alive)

    # A sample set on the devices is a handle to buffers the library owns: copies of a front-end object that holds one
    # (copy.deepcopy of a sampler after a run, of a GaussianInference) SHARE it -- it is never written after its creation
    # apart from the weights of the latest weighting pass -- and the one Python object frees it once.  It does not pickle:
    # the front-ends drop it from their pickled state and upload again on first use (advice r5).
    def __deepcopy__(self, memo):
        return self

    def __reduce__(self):
        raise TypeError("ShardedSamples live on the devices of a DeviceGroup and cannot be pickled; pickle .host() instead")

    def __len__(self):
        return self.N

    @property
    def shape(self):
        return (self.N, self.dim)

    @property
    def weights(self):
        if not self.has_weights:
            raise ValueError("no importance weights on the devices (DeviceGroup.importance_weights first)")
        return ShardedWeights(self)

    def shards(self):
        """[(device ordinal, first row, row count), ...] in device order"""
        lib, out = self.group.lib, []
        for p in range(len(self.group.devices)):
            b, c = C.c_int64(), C.c_int64()
            dev = _lib.check(lib.pmc_samples_shard(self._h, p, C.byref(b), C.byref(c)), "pmc_samples_shard")
            out.append((dev, b.value, c.value))
        return out

    def host(self):
        """the samples as an N x D host array"""
        x = np.empty((self.N, self.dim))
        _lib.check(self.group.lib.pmc_samples_download(self._h, _dp(x)), "pmc_samples_download")
        return x

    def origin(self):
        """generating component per sample (generated samples only; sorted)"""
        o = np.empty(self.N, dtype=np.int64)
        _lib.check(self.group.lib.pmc_samples_origin(self._h, _ip(o)), "pmc_samples_origin")
        return o

    def set_sample_weights(self, w):
        """sample weights of the VB E-step that stay on the devices next to the samples (``None`` removes them):
        ``DeviceGroup.vb_estep(samples, None, ...)`` then uses them without another upload"""
        w = None if w is None else _c64(w).reshape(self.N)
        _lib.check(self.group.lib.pmc_samples_set_sample_weights(self._h, _dp(w)), "pmc_samples_set_sample_weights")
        self.has_sample_weights = w is not None

    def host_weights(self):
        if self._host_w is None:
            raise ValueError("the weights of this run were not copied to the host (want_weights=False)")
        return self._host_w

    def free(self):
        if self._h is not None and self.group._ctx is not None:
            self.group.lib.pmc_samples_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # pragma: no cover
            pass


class ShardedMixture(object):
    """A Gauss / StudentT ``MixtureDensity`` as the devices of a ``DeviceGroup`` hold it (a copy of the parameter pack on
    every device).  The host density stays authoritative: ``update(density)`` after it changed."""

    def __init__(self, group, density):
        self.group, self._h = group, None
        fam, K, D, arrays = self._arrays(density)
        self.family, self.K, self.dim = fam, K, D
        h = C.c_void_p()
        _lib.check(group.lib.pmc_mixture_create(group._ctx, fam, K, D, *[_dp(a) for a in arrays], C.byref(h)),
                   "pmc_mixture_create")
        self._h = h
        self._key = self._fingerprint(arrays)

    # (a device-side copy of a host density, kept by its group: shared by copies, rebuilt from the density -- never pickled)
    def __deepcopy__(self, memo):
        return self

    def __reduce__(self):
        raise TypeError("a ShardedMixture is rebuilt from its host density (DeviceGroup.mixture) and cannot be pickled")

    @staticmethod
    def _arrays(density):
        from .density.gauss import Gauss
        from .density.student_t import StudentT
        comps = density.components
        first = type(comps[0])
        if first not in (Gauss, StudentT) or any(type(c) is not first for c in comps):
            raise TypeError("a DeviceGroup evaluates mixtures of only Gauss or only StudentT components")
        student = first is StudentT
        w = _c64(density.weights)
        mu = _c64([c.mu for c in comps])
        inv = _c64([c.inv_sigma for c in comps])
        ln = _c64([c.log_normalization for c in comps])
        dof = _c64([c.dof for c in comps]) if student else None
        return (PMC_KIND_STUDENT_T if student else PMC_KIND_GAUSS), len(comps), density.dim, (w, mu, inv, ln, dof)

    @staticmethod
    def _fingerprint(arrays):
        return tuple(None if a is None else a.tobytes() for a in arrays)

    def update(self, density):
        """bring the devices' copy up to date with the host density (no-op if nothing changed)"""
        fam, K, D, arrays = self._arrays(density)
        if (fam, K, D) != (self.family, self.K, self.dim):
            raise ValueError("the mixture changed its family / component count / dimension: make a new ShardedMixture")
        key = self._fingerprint(arrays)
        if key != self._key:
            _lib.check(self.group.lib.pmc_mixture_update(self._h, *[_dp(a) for a in arrays]), "pmc_mixture_update")
            self._key = key
        return self

    def free(self):
        if self._h is not None and self.group._ctx is not None:
            self.group.lib.pmc_mixture_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # pragma: no cover
            pass


class DeviceGroup(object):
    """``DeviceGroup([0, 1, 2, 3])``: one handle-layer context over these devices (``None``: the list in the environment
    variable ``PMC_HIP_DEVICES``, or every visible device).  Raises ``HipLibraryError`` without the library or a device
    -- there is no CPU fallback."""

    def __init__(self, devices=None):
        self.lib = _lib.load()
        self._ctx = None
        ids = [] if devices is None else [int(d) for d in devices]
        arr = (C.c_int * max(len(ids), 1))(*ids)
        h = C.c_void_p()
        _lib.check(self.lib.pmc_init_devices(len(ids), arr if ids else None, C.byref(h)), "pmc_init_devices")
        self._ctx = h
        n = self.lib.pmc_ctx_device_count(h)
        out = (C.c_int * n)()
        self.lib.pmc_ctx_devices(h, out, n)
        self.devices = list(out)
        self._mixtures = {}
        import weakref
        self._live_samples = weakref.WeakSet()             # ShardedSamples of this group that have not been freed
        self._live_states = weakref.WeakSet()              # VBStates on this group's context

    # a group is a process-wide handle: densities / samplers that hold one are deep-copied by the front-end; a pickled
    # one becomes a NEW group over the same device ordinals where it is loaded
    def __deepcopy__(self, memo):
        return self

    def __reduce__(self):
        return (DeviceGroup, (list(self.devices),))

    def close(self):
        if self._ctx is not None:
            for m in list(self._mixtures.values()):
                m.free()
            self._mixtures = {}
            # sample sets that are still alive (GaussianInference._samples, sampler.last_run): pmc_shutdown does not own
            # their device buffers, and once the context is gone ShardedSamples.free() can no longer return them (advice r5)
            for state in list(self._live_states):
                state.close()
            for smp in list(self._live_samples):
                smp.free()
            ctx, self._ctx = self._ctx, None
            self.lib.pmc_shutdown(ctx)

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @staticmethod
    def of(devices):
        """a DeviceGroup from what a front-end class was given as ``devices=`` (a group passes through)"""
        return devices if isinstance(devices, DeviceGroup) else DeviceGroup(devices)

    # ------------------------------------------------------------------ the K-sized state of a VB fit (backend.VBState)
    def ctx(self):
        return self._ctx

    @staticmethod
    def _timed(name, fn, *args):
        return fn(*args)

    def vb_state_supported(self, D):
        return 1 <= D <= min(int(self.lib.pmc_vb_max_dim()), int(self.lib.pmc_max_compiled_dim()))

    def vb_state(self, K, D):
        """prior, posterior and latest sums of a GaussianInference on this group's FIRST device (pmc_vb_state over a context of
        several devices: per E-step the others get the posterior in one peer copy, the statistics come back in device order)"""
        from .backend import VBState
        state = VBState(self, K, D)
        self._live_states.add(state)                           # (close() destroys what is still alive before the context goes)
        return state

    # ------------------------------------------------------------------ handles
    def configure(self, key, value):
        _lib.check(self.lib.pmc_ctx_configure(self._ctx, key.encode(), float(value)), "pmc_ctx_configure")

    def kernel_timing(self, on=True):
        _lib.check(self.lib.pmc_ctx_timing_enable(self._ctx, int(bool(on))), "pmc_ctx_timing_enable")

    def kernel_timings(self):
        """{kernel: dict(calls, ms, flops, bytes)} since the last call; calls / flops / bytes added over the devices, ms
        of the slowest device (they run side by side)"""
        buf = (_lib.Timing * 32)()
        n = C.c_int(0)
        _lib.check(self.lib.pmc_ctx_get_timings(self._ctx, C.cast(buf, C.c_void_p), 32, C.byref(n)), "pmc_ctx_get_timings")
        return {buf[i].name.decode(): dict(calls=buf[i].calls, ms=buf[i].ms, flops=buf[i].flops, bytes=buf[i].bytes)
                for i in range(min(n.value, 32))}

    def upload(self, x):
        """N x D host array -> ShardedSamples (contiguous blocks, device order = row order)"""
        x = _c64(x)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        h = C.c_void_p()
        _lib.check(self.lib.pmc_samples_upload(self._ctx, _dp(x), x.shape[0], x.shape[1], C.byref(h)), "pmc_samples_upload")
        return ShardedSamples(self, h, x.shape[0], x.shape[1])

    def mixture(self, density):
        """the devices' copy of ``density`` (kept per density object and refreshed when its parameters changed)"""
        m = self._mixtures.get(id(density))
        if m is not None and m._owner() is density and m._h is not None:
            try:
                return m.update(density)
            except ValueError:
                m.free()
        import weakref
        m = ShardedMixture(self, density)
        m._owner = weakref.ref(density)
        if len(self._mixtures) > 64:                      # (densities that went away)
            for key in [k for k, v in self._mixtures.items() if v._owner() is None]:
                self._mixtures.pop(key).free()
        self._mixtures[id(density)] = m
        return m

    def generate(self, density, counts, seed, first_sample=0):
        """MixtureDensity.propose(N, trace=True, shuffle=False) on the devices (mixture.pyx:159-212): ``counts`` from the
        caller's generator (bit-exact counts and origins), the samples from the Philox stream ``seed`` counted by the
        global row -- the same numbers whatever the number of devices."""
        mix = self.mixture(density)
        counts = np.ascontiguousarray(counts, dtype=np.int64).reshape(mix.K)
        chol = _c64([c.cholesky_sigma for c in density.components])
        h = C.c_void_p()
        _lib.check(self.lib.pmc_samples_generate(self._ctx, mix._h, _dp(chol), _ip(counts), C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                                 int(first_sample), C.byref(h)), "pmc_samples_generate")
        return ShardedSamples(self, h, int(counts.sum()), density.dim, counts=counts)

    # ------------------------------------------------------------------ operations
    def logpdf(self, density, samples, want_individual=False):
        """MixtureDensity.multi_evaluate (mixture.pyx:112-156): log q (N) [, the N x K component values]"""
        mix = self.mixture(density)
        out = np.empty(samples.N)
        ind = np.empty((samples.N, mix.K)) if want_individual else None
        _lib.check(self.lib.pmc_mix_logpdf(mix._h, samples._h, _dp(out), _dp(ind)), "pmc_mix_logpdf")
        return (out, ind) if want_individual else out

    def importance_weights(self, proposal, samples, log_target=None, target=None, want_weights=True, want_log_target=False):
        """ImportanceSampler._calculate_weights (importance_sampling.py:197-215) on all devices: w = exp(log P - log q) with
        log P from the host (``log_target``, N) or from a second mixture (``target``, evaluated in the same pass).  The
        weights stay with ``samples`` on the devices (``samples.weights``).  Returns dict(weights, log_target, sums) --
        sums = (sum w, sum w log w, sum w^2) over all samples: perplexity and ESS follow from them."""
        q = self.mixture(proposal)
        t = self.mixture(target) if target is not None else None
        lt = _c64(log_target).reshape(samples.N) if log_target is not None else None
        w = np.empty(samples.N) if want_weights else None
        lto = np.empty(samples.N) if (want_log_target and t is not None) else None
        sums = np.empty(3)
        rc = self.lib.pmc_is_weights(q._h, samples._h, _dp(lt), t._h if t is not None else None, _dp(w), _dp(lto), _dp(sums))
        if rc < 0 and "math range error" in _lib.last_error():
            raise OverflowError('math range error')               # math.exp, importance_sampling.py:207
        _lib.check(rc, "pmc_is_weights")
        samples.has_weights = True
        samples._host_w = w
        return dict(weights=w, log_target=lto if t is not None else lt, sums=(float(sums[0]), float(sums[1]), float(sums[2])))

    def vb_estep(self, samples, sample_w, m, W, nu, beta, ln_pi, ln_lambda, shift=None, want_nk=False):
        """GaussianInference.E_step (variational.pyx:116-127) over all devices, the reference's conventions:
        dict(N_comp, x_mean_comp, S, log_q_Z[, r, log_rho])."""
        m = _c64(m)
        K, D = m.shape
        Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
        r = np.empty((samples.N, K)) if want_nk else None
        lr = np.empty((samples.N, K)) if want_nk else None
        sw = _c64(sample_w).reshape(samples.N) if sample_w is not None else None
        sh = _c64(shift).reshape(K, D) if shift is not None else None
        _lib.check(self.lib.pmc_vb_estep(self._ctx, samples._h, _dp(sw), K, _dp(m), _dp(_c64(W)), _dp(_c64(nu)), _dp(_c64(beta)),
                                         _dp(_c64(ln_pi)), _dp(_c64(ln_lambda)), _dp(sh), _dp(Nk), _dp(xbar), _dp(S), _dp(elq),
                                         _dp(r), _dp(lr)), "pmc_vb_estep")
        return dict(N_comp=Nk, x_mean_comp=xbar, S=S, log_q_Z=float(elq[0]), r=r, log_rho=lr)

    def pmc_update_stats(self, density, samples, weights=None, latent=None, rb=True):
        """the N-sized part of gaussian_pmc / student_t_pmc (pmc.pyx:120-246, :499-739) over all devices: dict(alpha, mu,
        sigma, dof_const, loglik, norm) for the live components (rows of dead ones are left zero).  ``weights``: None,
        a host array, or ``samples.weights`` (left on the devices by ``importance_weights``); ``latent``: None, a host
        int array, or the string 'origin' (the generating components ``generate`` kept with the samples)."""
        mix = self.mixture(density)
        K, D = mix.K, mix.dim
        alpha, mu, sigma = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D))
        dofc = np.zeros(K) if mix.family == PMC_KIND_STUDENT_T else None
        ll, norm = np.zeros(1), np.zeros(1)
        on_dev = isinstance(weights, ShardedWeights)
        if on_dev and weights.samples is not samples:
            raise ValueError("these device weights belong to another sample set")
        hw = None if (weights is None or on_dev) else _c64(weights).reshape(samples.N)
        use_origin = isinstance(latent, str) and latent == 'origin'
        hl = None if (latent is None or use_origin) else np.ascontiguousarray(latent, dtype=np.int64).reshape(samples.N)
        if not rb and latent is None:
            raise ValueError('`rb` must be True if `latent` is not provided!')
        _lib.check(self.lib.pmc_pmc_update_stats(self._ctx, mix._h, samples._h, _dp(hw), int(on_dev), _ip(hl), int(bool(rb)),
                                                 _dp(alpha), _dp(mu), _dp(sigma), _dp(dofc), _dp(ll), _dp(norm)),
                   "pmc_pmc_update_stats")
        return dict(alpha=alpha, mu=mu, sigma=sigma, dof_const=dofc, loglik=float(ll[0]), norm=float(norm[0]))

    def weighted_moments(self, samples, weights=None, want_cov=True):
        """calculate_mean / calculate_covariance (importance_sampling.py:46-83) over all devices: (mean, cov)"""
        D = samples.dim
        mean, cov = np.empty(D), (np.empty((D, D)) if want_cov else None)
        on_dev = isinstance(weights, ShardedWeights)
        if on_dev and weights.samples is not samples:
            raise ValueError("these device weights belong to another sample set")
        hw = None if (weights is None or on_dev) else _c64(weights).reshape(samples.N)
        _lib.check(self.lib.pmc_weighted_moments(self._ctx, samples._h, _dp(hw), int(on_dev), _dp(mean), _dp(cov)),
                   "pmc_weighted_moments")
        return mean, cov
