"""Device backend: libpmc_hip.so driven through ctypes, with PyTorch-ROCm used only as plumbing
(device memory, the current HIP stream, and -- in pypmc_amd.parallel -- torch.distributed/RCCL).

``HipBackend`` is the only backend the package ships.  The front-end classes take a ``backend``
argument solely so that the CPU-only test-suite can inject a checker from ``tests/``; the
default is always the HIP path and it raises ``HipLibraryError`` when the library or the GPU is
missing.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (HipLibraryError, NotPositiveDefinite, PMC_KIND_GAUSS, PMC_KIND_STUDENT_T,  # noqa: F401
                   NSCALARS)

__all__ = ["ComponentSet", "Responsibilities", "HipBackend", "get_backend", "set_default_backend", "HipLibraryError",
           "NotPositiveDefinite"]


class ComponentSet(object):
    """Host description of K mixture components as the kernels consume them
    (see ``enum pmc_kind`` in include/pmc_hip.h for the meaning of c0..c3).

    ``column[k]`` is the column of component k in N x ``ld`` row-major outputs, so a subset of
    a K_total-component mixture is described by its live components with ``ld = K_total``.
    """

    def __init__(self, kind, mu, precision, c0=None, c1=None, c2=None, c3=None, weight=None,
                 column=None, ld=None):
        self.kind = int(kind)
        self.mu = np.ascontiguousarray(mu, dtype=np.float64)
        assert self.mu.ndim == 2
        self.K, self.D = self.mu.shape
        self.precision = np.ascontiguousarray(precision, dtype=np.float64).reshape(self.K, self.D, self.D)

        def vec(v, default):
            if v is None:
                return np.full(self.K, default, dtype=np.float64)
            v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.K)
            return v
        self.c0, self.c1, self.c2, self.c3 = vec(c0, 0.), vec(c1, 0.), vec(c2, 0.), vec(c3, 0.)
        self.weight = vec(weight, 1.)
        if column is None:
            column = np.arange(self.K)
        self.column = np.ascontiguousarray(column, dtype=np.int32).reshape(self.K)
        self.ld = int(ld) if ld is not None else int(self.column.max()) + 1
        assert self.ld > int(self.column.max())


class MahaTiles(object):
    """Mahalanobis forms maha_nk a weighting pass kept on the device (library tile-major layout), together with
    what they belong to: the sample count and the ComponentSet -- i.e. the parameter state -- they were evaluated
    with.  ``gaussian_pmc / student_t_pmc(..., mahalanobis=tiles)`` use them only for that very mixture."""

    def __init__(self, data, N, comps):
        self.data, self.N, self.K, self.comps = data, int(N), int(comps.K), comps

    def matches(self, comps_full):
        """True if ``comps_full`` (complete mixture) has the means and precisions these values were computed
        with (the weights and the normalisations do not enter maha_nk)."""
        c = self.comps
        return comps_full is c or (comps_full.K == c.K and np.array_equal(comps_full.mu, c.mu) and
                                   np.array_equal(comps_full.precision, c.precision))


class Responsibilities(object):
    """u_nk = w_n rho_nk of a Gaussian proposal on the device (tile-major), left behind by a weighting pass that knew
    the update follows (``importance_weights(..., emit=True)``), together with what they belong to.
    ``gaussian_pmc(..., responsibilities=...)`` reduces them to the statistics without any responsibility kernel."""

    def __init__(self, data, N, comps, weights, vsums=None, gscale=None, samples=None, live=None):
        self.data, self.N, self.K, self.comps, self.weights = data, int(N), int(comps.K), comps, weights
        # a mixture with pruned components (weight 0): the columns are its LIVE components, in ascending order -- what the
        # update forms responsibilities for (pmc.pyx:98-103); ``comps`` stays the complete mixture the pass evaluated
        self.live = list(range(comps.K)) if live is None else [int(k) for k in live]
        self.K = len(self.live)
        self.vsums = vsums            # Student-t: the 2 K sums of the degree-of-freedom condition (device)
        self.gscale = gscale          # per-(sample, 16 components) factors u is still to be multiplied with (ABI 2)
        # what the values were formed on: the sample tensor's storage, the weight tensor's modification counter
        self._samples_key = self._key(samples)
        self._weights_version = getattr(weights, "_version", None)

    @staticmethod
    def _key(t):
        # storage and shape -- and, for a tensor that owns its storage, its modification counter (in-place edits, a
        # recycled allocation).  A run of a DeviceHistory is a VIEW whose counter moves with every later append to the
        # history: there the address and the shape are all there is.
        if not hasattr(t, "data_ptr"):
            return None
        own = getattr(t, "_base", None) is None
        return (t.data_ptr(), tuple(t.shape), getattr(t, "_version", None) if own else None)

    def host_matrix(self, be):
        """u as an N x K host array (the tile-major values times their groups' factors)"""
        tile = be.tile
        nt = (self.N + tile - 1) // tile
        t = be.tohost(self.data)[:nt * self.K * tile].reshape(nt, self.K, tile)
        if self.gscale is not None:
            ng = (self.K + 15) // 16
            f = be.tohost(self.gscale)[:nt * ng * tile].reshape(nt, ng, tile)
            t = t * np.repeat(f, 16, axis=1)[:, :self.K, :]
        return np.concatenate([t[i].T for i in range(nt)])[:self.N] if nt else np.zeros((0, self.K))

    def mismatch(self, comps_full, weights, samples=None):
        """None for the very mixture (means, precisions, component weights, normalisations), the very sample weights
        (the importance weights of that pass: the same tensor, not modified in place since) and -- when ``samples`` is
        a device tensor -- the very sample array these values were formed with; else which of them differs:
        'density', 'weights' or 'samples'"""
        c = self.comps
        same = comps_full is c or (comps_full.K == c.K and comps_full.kind == c.kind and
                                   np.array_equal(comps_full.mu, c.mu) and np.array_equal(comps_full.precision, c.precision)
                                   and np.array_equal(comps_full.weight, c.weight) and np.array_equal(comps_full.c0, c.c0)
                                   and np.array_equal(comps_full.c3, c.c3))
        if not same:
            return 'density'
        if weights is not self.weights or getattr(weights, "_version", None) != self._weights_version:
            return 'weights'
        key = self._key(samples)
        if key is None or self._samples_key is None:
            return None
        if key[:2] != self._samples_key[:2]:
            return 'samples'
        return None if (key[2] is None or self._samples_key[2] is None or key[2] == self._samples_key[2]) else 'samples'

    def matches(self, comps_full, weights, samples=None):
        return self.mismatch(comps_full, weights, samples) is None


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class HipBackend(object):
    """MI355X backend.  All array results are torch CUDA tensors (float64)."""

    name = "hip"

    def __init__(self, device=None):
        self.lib = _lib.load()
        try:
            import torch
        except Exception as exc:  # pragma: no cover
            raise HipLibraryError("PyTorch (ROCm) is required for device memory: %s" % exc)
        self.torch = torch
        if not torch.cuda.is_available():
            raise HipLibraryError("no HIP device visible (torch.cuda.is_available() is False); "
                                  "pypmc_amd has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        arch = C.create_string_buffer(256)
        _lib.check(self.lib.pmc_device_arch(self.device.index or 0, arch, 256), "pmc_device_arch")
        self.arch = arch.value.decode()
        if not self.arch.startswith("gfx950"):
            raise HipLibraryError("kernels are built for gfx950 (MI355X); device reports %r" % self.arch)
        self.tile = self.lib.pmc_tile()
        self._ws = {}            # launch stream -> scratch (calls on different streams may overlap)
        self._bufs = {}
        self.profile = None      # set to a list to collect (entry point, start, end) event triples
        self._ctx = None         # a handle-layer context on this device (include/pmc_ctx.h), made on first use

    # Densities keep a reference to their backend and are deep-copied / pickled by the front-end
    # (MixtureDensity copies its components, ImportanceSampler its proposal): a backend is a
    # process-wide handle, never duplicated, and re-created from the default when unpickled.
    def __deepcopy__(self, memo):
        return self

    def __reduce__(self):
        return (get_backend, ())

    # ------------------------------------------------------------------ plumbing
    def _timed(self, name, fn, *args):
        """Call a C-ABI entry point, bracketed by events on the launch stream when profiling."""
        if self.profile is None:
            return fn(*args)
        start = self.torch.cuda.Event(enable_timing=True)
        end = self.torch.cuda.Event(enable_timing=True)
        start.record()
        rc = fn(*args)
        end.record()
        self.profile.append((name, start, end))
        return rc

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def configure(self, key, value):
        """pmc_configure: library options ("stats_common_shift_min_k", "stats_common_shift_min_n",
        "stats_common_shift_limit")."""
        _lib.check(self.lib.pmc_configure(key.encode(), float(value)), "pmc_configure")

    def option(self, key):
        """pmc_option_get: the current value of a library option"""
        v = C.c_double(0.)
        _lib.check(self.lib.pmc_option_get(key.encode(), C.byref(v)), "pmc_option_get")
        return v.value

    def option_default(self, key):
        """pmc_option_default: the built-in value of a library option"""
        v = C.c_double(0.)
        _lib.check(self.lib.pmc_option_default(key.encode(), C.byref(v)), "pmc_option_default")
        return v.value

    def reset_option(self, key):
        self.configure(key, self.option_default(key))

    def kernel_timing(self, on=True):
        """Switch the library's own kernel timing on / off (pmc_timing_enable: HIP events on the launch
        stream around every hot kernel)."""
        _lib.check(self.lib.pmc_timing_enable(int(bool(on))), "pmc_timing_enable")

    def kernel_timings(self):
        """{kernel: dict(calls, ms, flops, bytes)} since the last call (pmc_get_timings; synchronises)."""
        buf = (_lib.Timing * 16)()
        n = C.c_int(0)
        _lib.check(self.lib.pmc_get_timings(C.cast(buf, C.c_void_p), 16, C.byref(n)), "pmc_get_timings")
        return {buf[i].name.decode(): dict(calls=buf[i].calls, ms=buf[i].ms, flops=buf[i].flops, bytes=buf[i].bytes)
                for i in range(min(n.value, 16))}

    def maha_gemm_report(self, N, K, D):
        """pmc_maha_gemm_report for the current stream's workspace: dict(norms, refused, workgroups) of the last call
        that took the matrix-product form of the Mahalanobis forms with this shape (None: the shape does not take it)"""
        if int(self.lib.pmc_maha_gemm_tiles(N, K, D)) <= 0:
            return None
        ws = self._workspace(N, K, D)
        norms = np.zeros(3)
        refused, wgs = C.c_int64(0), C.c_int64(0)
        _lib.check(self.lib.pmc_maha_gemm_report(self._p(ws), N, K, D, self._stream(), _dptr(norms), C.byref(refused),
                                                 C.byref(wgs)), "pmc_maha_gemm_report")
        return dict(norms=norms, refused=int(refused.value), workgroups=int(wgs.value))

    def asdevice(self, a, dtype=None):
        """numpy array / torch tensor -> contiguous tensor on this device."""
        torch = self.torch
        dtype = dtype or torch.float64
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        np_dtype = {torch.float64: np.float64, torch.int64: np.int64}[dtype]
        a = np.ascontiguousarray(a, dtype=np_dtype)
        if not a.flags.writeable:                       # e.g. a DeviceHistory host copy
            a = a.copy()
        return torch.from_numpy(a).to(self.device)

    def tohost(self, t):
        return t.detach().cpu().numpy()

    def empty(self, shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.torch.float64, device=self.device)

    def zeros(self, shape, dtype=None):
        return self.torch.zeros(shape, dtype=dtype or self.torch.float64, device=self.device)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def _workspace(self, N, K, D):
        need = _lib.check(self.lib.pmc_workspace_bytes(N, K, D), "pmc_workspace_bytes")
        key = self.torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._ws[key] = self.torch.empty(int(need), dtype=self.torch.uint8, device=self.device)
        return ws

    def _tilebuf(self, name, N, K):
        need = _lib.check(self.lib.pmc_tile_buffer_len(N, K), "pmc_tile_buffer_len")
        key = (name, self.torch.cuda.current_stream(self.device).cuda_stream)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < need:
            buf = self.torch.empty(int(max(need, 1)), dtype=self.torch.float64, device=self.device)
            self._bufs[key] = buf
        return buf

    def release(self):
        """Drop cached scratch buffers and the parameter packs cached with the front-end's ComponentSets."""
        self._ws = {}
        self._bufs = {}
        from .density.mixture import clear_component_cache
        clear_component_cache()

    def pack(self, comps):
        """ComponentSet -> device parameter pack (host Cholesky in pmc_pack_components).  A ComponentSet is
        not modified after construction, so its pack is built and uploaded once and kept with it."""
        cached = getattr(comps, "_pack", None)
        if cached is not None and cached[0] is self:
            return cached[1]
        comps._pack = (self, self._build_pack(comps))
        return comps._pack[1]

    # K-sized linear algebra of a PMC update where it is big enough to matter (K = 128, D = 40: 1.4 ms of LAPACK and
    # 0.4 ms of host Cholesky per iteration, with the GPU idle): from this many matrix elements on, on the device
    DEVICE_LINALG_FROM = 60000

    def chol_inv_det_batch(self, sig):
        """tools._linalg.chol_inv_det_batch (lower factors, inverses, log determinants of K symmetric positive definite
        matrices) on the device: pmc_spd_inverse_device -- potrf / potri's algorithm, one wavefront per matrix; agrees
        with LAPACK to rounding.  Raises numpy.linalg.LinAlgError if a matrix does not factorise (callers fall back to
        the per-component loop, as with the host batch).  None where it does not apply (D > 64)."""
        sig = np.ascontiguousarray(sig, dtype=np.float64)
        K, D = sig.shape[0], sig.shape[1]
        if sig.ndim != 3 or sig.shape[2] != D or D > int(self.lib.pmc_vb_max_dim()):
            return None
        if not np.isfinite(sig).all():
            raise np.linalg.LinAlgError('array must not contain infs or NaNs')
        d_in = self.torch.from_numpy(sig).to(self.device)
        n = int(self.lib.pmc_spd_inverse_len(K, D))
        d_out = self.empty(n)
        _lib.check(self.lib.pmc_spd_inverse_device(K, D, self._p(d_in), self._p(d_out), self._stream()), "pmc_spd_inverse_device")
        # (3.3 MB at K = 128, D = 40: through torch's cached pinned allocator, 0.13 ms instead of 0.35 from pageable memory)
        pinned = self.torch.empty(n, dtype=self.torch.float64, pin_memory=True)
        pinned.copy_(d_out, non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        out = pinned.numpy().reshape(K, 2 * D * D + 3)
        if (out[:, 2 * D * D + 1] != 0).any() or not np.isfinite(out[:, 2 * D * D]).all():
            raise np.linalg.LinAlgError('a matrix of the batch is not positive definite (or its determinant is not finite)')
        return (np.ascontiguousarray(out[:, :D * D]).reshape(K, D, D), np.ascontiguousarray(out[:, D * D:2 * D * D]).reshape(K, D, D),
                out[:, 2 * D * D].copy())

    def _build_pack_device(self, comps):
        """the parameter pack built by the device's builder (pmc_pack_components_device: the host builder's bits) from
        ONE upload of the raw parameters; None where the device builder does not apply"""
        K, D = comps.K, comps.D
        if D > int(self.lib.pmc_max_compiled_dim()):
            return None
        KD, KDD = K * D, K * D * D
        raw = np.empty(KD + KDD + 5 * K + (K + 1) // 2 + 1)
        raw[:KD] = comps.mu.reshape(-1)
        raw[KD:KD + KDD] = comps.precision.reshape(-1)
        o = KD + KDD
        for i, v in enumerate((comps.c0, comps.c1, comps.c2, comps.c3, comps.weight)):
            raw[o + i * K:o + (i + 1) * K] = v
        col = raw[o + 5 * K:o + 5 * K + (K + 1) // 2].view(np.int32)
        col[:K] = comps.column
        d = self.torch.from_numpy(raw).to(self.device)
        stride = _lib.check(self.lib.pmc_pack_stride(D), "pmc_pack_stride")
        pack, status = self.empty(K * stride), self.empty(2 * K)
        base, f8 = d.data_ptr(), 8
        at = lambda off: C.c_void_p(base + f8 * off)
        rc = self.lib.pmc_pack_components_device(K, D, at(0), at(KD), at(o), at(o + K), at(o + 2 * K), at(o + 3 * K), at(o + 4 * K),
                                                 at(o + 5 * K), self._p(pack), self._p(status), None, None, self._stream())
        if rc < 0:
            return None                                        # (a layout the device builder does not serve: the host's)
        st = status.cpu().numpy()
        _lib.check(self.lib.pmc_pack_status(K, _dptr(st)), "pmc_pack_components_device")
        return pack

    def _build_pack(self, comps):
        if comps.K * comps.D * comps.D >= self.DEVICE_LINALG_FROM:
            pack = self._build_pack_device(comps)
            if pack is not None:
                return pack
        stride = _lib.check(self.lib.pmc_pack_stride(comps.D), "pmc_pack_stride")
        host = np.empty(comps.K * stride, dtype=np.float64)
        _lib.check(self.lib.pmc_pack_components(
            comps.K, comps.D, _dptr(comps.mu), _dptr(comps.precision), _dptr(comps.c0),
            _dptr(comps.c1), _dptr(comps.c2), _dptr(comps.c3), _dptr(comps.weight),
            comps.column.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(host)), "pmc_pack_components")
        return self.torch.from_numpy(host).to(self.device)

    def _means_pack(self, mu, K, D):
        """pmc_pack_means: a device pack that carries only the K x D shifts of the statistics kernel."""
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(K, D)
        stride = _lib.check(self.lib.pmc_pack_stride(D), "pmc_pack_stride")
        host = np.empty(K * stride, dtype=np.float64)
        _lib.check(self.lib.pmc_pack_means(K, D, _dptr(mu), _dptr(host)), "pmc_pack_means")
        return self.torch.from_numpy(host).to(self.device)

    # ------------------------------------------------------------------ operations
    def logpdf(self, x, comps, want_out=True, individual=None, want_individual=False,
               max_init_zero=False, log_target=None, sample_w=None, want_scalars=False, pack=None,
               out=None, keep=False):
        """pmc_mixture_logpdf.  ``x`` N x D device tensor.  Returns dict(out, individual, weights,
        scalars) of device tensors (None where not requested).  ``out``: optional contiguous
        N-vector on the device to receive log q (e.g. a row of combine_weights' q matrix)."""
        x = self.asdevice(x)
        N, D = x.shape
        assert D == comps.D, "sample dimension %d != component dimension %d" % (D, comps.D)
        pack = self.pack(comps) if pack is None else pack
        if out is not None:
            assert out.shape == (N,) and out.is_contiguous() and out.dtype == self.torch.float64
        elif want_out:
            out = self.empty(N)
        if individual is None and want_individual:
            individual = self.empty((N, comps.ld))
        if individual is not None:
            assert individual.shape == (N, comps.ld) and individual.is_contiguous()
        lt = self.asdevice(log_target).reshape(N) if log_target is not None else None
        weights = self.empty(N) if lt is not None else None
        sw = self.asdevice(sample_w).reshape(N) if sample_w is not None else None
        scalars = self.zeros(NSCALARS) if want_scalars else None
        # (always a workspace: with it the library may take its matrix-product form of the Mahalanobis forms)
        ws = self._workspace(N, comps.K, D)
        tiles = self._new_tiles(N, comps) if keep else None
        _lib.check(self._timed(
            "pmc_mixture_logpdf[K=%d]" % comps.K, self.lib.pmc_mixture_logpdf_keep,
            self._p(x), N, D, self._p(pack), comps.K, comps.kind, int(bool(max_init_zero)),
            self._p(out), self._p(individual), comps.ld, self._p(lt), self._p(weights), self._p(sw),
            self._p(scalars), self._p(ws), self._p(tiles.data) if keep else None, self._stream()), "pmc_mixture_logpdf")
        return dict(out=out, individual=individual, weights=weights, scalars=scalars, tiles=tiles)

    def _new_tiles(self, N, comps):
        """Buffer for the Mahalanobis forms a kept weighting pass leaves behind (see MahaTiles)."""
        assert comps.ld == comps.K and bool((comps.column == np.arange(comps.K)).all()), \
            "Mahalanobis forms are kept for complete mixtures only"
        n = int(self.lib.pmc_maha_tiles_size(N, comps.K))
        return MahaTiles(self.empty(max(n, 1)), N, comps)

    def importance_weights(self, x, comps, target, sample_w=None, want_out=False, want_log_target=False,
                           pack=None, target_pack=None, keep=False, emit=False):
        """pmc_importance_weights: w = exp(log P - log q) for a mixture target P (``target``) and proposal
        q (``comps``) in one pass over ``x``.  Returns dict(weights, scalars, out, log_target[, tiles | responsibilities]).
        ``emit`` (Gauss / Student-t proposal, compiled dimensions, every component alive): the pass also leaves
        u = w rho [gamma] for the update (``Responsibilities``; pmc_importance_weights_emit)."""
        if emit and not keep and sample_w is None and self.can_emit(comps):
            return self._importance_weights_emit(x, comps, target, want_out, want_log_target, pack, target_pack)
        x = self.asdevice(x)
        N, D = x.shape
        assert D == comps.D == target.D, "sample / proposal / target dimensions differ"
        pack = self.pack(comps) if pack is None else pack
        target_pack = self.pack(target) if target_pack is None else target_pack
        out = self.empty(N) if want_out else None
        lt = self.empty(N) if want_log_target else None
        weights = self.empty(N)
        sw = self.asdevice(sample_w).reshape(N) if sample_w is not None else None
        scalars = self.zeros(NSCALARS)
        ws = self._workspace(N, max(comps.K, target.K), D)
        tiles = self._new_tiles(N, comps) if keep else None
        _lib.check(self._timed(
            "pmc_importance_weights[K=%d+%d]" % (comps.K, target.K), self.lib.pmc_importance_weights_keep,
            self._p(x), N, D, self._p(pack), comps.K, comps.kind, self._p(target_pack), target.K, target.kind,
            self._p(out), self._p(lt), self._p(weights), self._p(sw), self._p(scalars), self._p(ws),
            self._p(tiles.data) if keep else None, self._stream()), "pmc_importance_weights")
        return dict(weights=weights, scalars=scalars, out=out, log_target=lt, tiles=tiles)

    @staticmethod
    def can_emit(comps):
        """does the emitting form of the weighting pass apply to this proposal?  (Gauss / Student-t, a compiled
        dimension, the complete mixture, at least one component alive -- pruned components, weight 0, are sorted behind
        the live ones and get no columns: round 6) -- callers that need the update's inputs either way ask BEFORE the
        pass and keep the Mahalanobis forms instead where it does not"""
        w = comps.weight
        return comps.kind in (PMC_KIND_GAUSS, PMC_KIND_STUDENT_T) and comps.D <= 64 and comps.ld == comps.K \
            and bool((w >= 0).all()) and bool(np.isfinite(w).all()) and bool((w != 0).any())

    def _importance_weights_emit(self, x, comps, target, want_out, want_log_target, pack, target_pack):
        x = self.asdevice(x)
        N, D = x.shape
        assert D == comps.D == target.D, "sample / proposal / target dimensions differ"
        live = np.nonzero(comps.weight != 0)[0]
        Kl = len(live)
        if Kl < comps.K:
            # pruned components (pmc.pyx:109-117 leaves them in the mixture with weight 0): the pass evaluates them -- they
            # take part in log q's row maximum -- but the update forms no responsibilities for them (pmc.pyx:98-103).  The
            # pack is sorted live components first (pmc_importance_weights_emit_live): u gets the Kl live columns only
            order = np.concatenate([live, np.nonzero(comps.weight == 0)[0]])
            sorted_set = ComponentSet(comps.kind, comps.mu[order], comps.precision[order], comps.c0[order], comps.c1[order],
                                      comps.c2[order], comps.c3[order], weight=comps.weight[order], column=comps.column[order],
                                      ld=comps.ld)
            pack = self.pack(sorted_set)
        else:
            pack = self.pack(comps) if pack is None else pack
        target_pack = self.pack(target) if target_pack is None else target_pack
        out = self.empty(N) if want_out else None
        lt = self.empty(N) if want_log_target else None
        weights = self.empty(N)
        scalars = self.zeros(NSCALARS)
        ws = self._workspace(N, max(comps.K, target.K), D)
        u = self.empty(max(int(self.lib.pmc_tile_buffer_len(N, Kl)), 1))           # the caller's: outlives this call
        gscale = self.empty(max(int(self.lib.pmc_gscale_len(N, Kl)), 1))
        vsums = self.zeros(2 * Kl) if comps.kind == PMC_KIND_STUDENT_T else None
        _lib.check(self._timed(
            "pmc_importance_weights_emit[K=%d+%d]" % (comps.K, target.K), self.lib.pmc_importance_weights_emit_live,
            self._p(x), N, D, self._p(pack), comps.K, Kl, comps.kind, self._p(target_pack), target.K, target.kind,
            self._p(out), self._p(lt), self._p(weights), self._p(scalars), self._p(ws), self._p(u), self._p(gscale),
            self._p(vsums), self._stream()), "pmc_importance_weights_emit_live")
        return dict(weights=weights, scalars=scalars, out=out, log_target=lt, tiles=None,
                    responsibilities=Responsibilities(u, N, comps, weights, vsums, gscale, samples=x,
                                                      live=live if Kl < comps.K else None))

    def estep_from_u(self, x, comps, resp, out=None):
        """pmc_estep_from_u: the statistics of responsibilities a weighting pass left behind (``Responsibilities``).
        Same return value as ``estep``."""
        x = self.asdevice(x)
        N, D = x.shape
        assert resp.N == N and resp.K == comps.K and D == comps.D, "responsibilities belong to another sample set / mixture"
        # (``comps``: the components the columns stand for -- the LIVE components of the mixture the pass evaluated)
        K = comps.K
        ps = int(self.lib.pmc_stats_stride(D))
        nflat = NSCALARS + K * ps + 2 * K
        flat = out if out is not None else self.zeros(nflat)
        assert flat.numel() == nflat
        flat[:NSCALARS] = 0.
        if resp.vsums is not None:
            flat[NSCALARS + K * ps:] = resp.vsums
        if resp.gscale is not None:
            _lib.check(self._timed(
                "pmc_estep_from_u", self.lib.pmc_estep_from_u_grouped, self._p(x), N, D, self._p(self.pack(comps)), K,
                comps.kind, self._p(resp.data), self._p(resp.gscale), self._p(flat[NSCALARS:]),
                self._p(self._workspace(N, K, D)), self._stream()), "pmc_estep_from_u_grouped")
        else:
            _lib.check(self._timed(
                "pmc_estep_from_u", self.lib.pmc_estep_from_u, self._p(x), N, D, self._p(self.pack(comps)), K, comps.kind,
                self._p(resp.data), self._p(flat[NSCALARS:]), self._p(self._workspace(N, K, D)), self._stream()),
                "pmc_estep_from_u")
        return dict(stats=flat, r=None, log_rho=None, exponent=None)

    def weight_sums(self, w):
        """(sum w, sum w log w [zeros masked], sum w^2) as a device tensor of NSCALARS doubles."""
        w = self.asdevice(w).reshape(-1)
        N = w.shape[0]
        scalars = self.zeros(NSCALARS)
        ws = self._workspace(max(N, 1), 1, 1)
        _lib.check(self.lib.pmc_weight_sums(self._p(w), N, self._p(scalars), self._p(ws),
                                            self._stream()), "pmc_weight_sums")
        return scalars

    def propose(self, mu, chol, dof, counts, seed, first_sample=0, want_origin=True, out=None):
        """pmc_propose: samples of a Gauss / Student-t mixture for host-drawn component ``counts``.
        mu K x D, chol K x D x D (lower Cholesky factors of the covariances), dof K or None.
        Returns (x N x D, origin N int64 or None) as device tensors, ordered by component."""
        torch = self.torch
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        K, D = mu.shape
        counts = np.asarray(counts, dtype=np.int64).reshape(K)
        offsets = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        N = int(offsets[-1])
        d_mu = self.asdevice(mu)
        d_chol = self.asdevice(np.ascontiguousarray(chol, dtype=np.float64).reshape(K, D, D))
        d_dof = self.asdevice(np.ascontiguousarray(dof, dtype=np.float64).reshape(K)) if dof is not None else None
        d_off = self.asdevice(offsets, torch.int64)
        if out is not None:
            assert tuple(out.shape) == (N, D) and out.is_contiguous() and out.dtype == torch.float64
            x = out
        else:
            x = self.empty((N, D))
        origin = self.empty(N, torch.int64) if want_origin else None
        _lib.check(self._timed(
            "pmc_propose", self.lib.pmc_propose, self._p(d_mu), self._p(d_chol), self._p(d_dof),
            self._p(d_off), K, D, N, int(first_sample), C.c_uint64(int(seed) & (2 ** 64 - 1)),
            self._p(x), self._p(origin), self._stream()), "pmc_propose")
        return x, origin

    def logsumexp2d(self, a, w):
        """row-wise log sum_k w_k exp(a_nk) of an N x K matrix (device tensor result)."""
        a = self.asdevice(a)
        w = self.asdevice(w).reshape(-1)
        N, K = a.shape
        assert w.shape[0] == K
        out = self.empty(N)
        _lib.check(self.lib.pmc_logsumexp2d(self._p(a), self._p(w), N, K, self._p(out),
                                            self._stream()), "pmc_logsumexp2d")
        return out

    def combine_weights(self, q, counts, t, omega, n_total, log_scale):
        """pmc_combine_weights: deterministic-mixture weights of run ``t``; ``q`` T x N with
        q[l, n] = log q_l(x^t_n).  Returns (weights N, number of non-finite results) on the device."""
        q = self.asdevice(q)
        T, N = q.shape
        counts = self.asdevice(np.asarray(counts, dtype=np.float64)).reshape(T)
        omega = self.asdevice(omega).reshape(N)
        out = self.empty(N)
        flag = self.zeros(1)
        _lib.check(self.lib.pmc_combine_weights(self._p(q), N, T, self._p(counts), int(t), self._p(omega),
                                                float(n_total), int(bool(log_scale)), self._p(out),
                                                self._p(flag), self._stream()), "pmc_combine_weights")
        return out, flag

    def estep(self, x, comps, mode, max_init_zero=False, sample_w=None, latent=None,
              want_r=False, want_log_rho=False, want_exponent=False, pack=None, out=None, shift=None):
        """Responsibilities (pmc_responsibilities) followed by the sufficient statistics
        (pmc_sufficient_stats) of the same samples.

        ``shift`` (K x D): take the moments about these points instead of the components' means -- the second
        pass of mix_adapt when a mean turned out far from its component (``_stats.shift_is_far``).

        Returns dict(stats = [scalars(NSCALARS) | K*stats_stride | K*2 Student-t sums] one flat
        device tensor (so that a multi-GPU caller all-reduces a single buffer), r, log_rho,
        exponent).
        """
        torch = self.torch
        x = self.asdevice(x)
        N, D = x.shape
        assert D == comps.D
        K = comps.K
        pack = self.pack(comps) if pack is None else pack
        sw = self.asdevice(sample_w).reshape(N) if sample_w is not None else None
        lat = self.asdevice(latent, torch.int64).reshape(N) if latent is not None else None
        student = comps.kind == PMC_KIND_STUDENT_T
        r = self.zeros((N, comps.ld)) if want_r else None
        log_rho = self.zeros((N, comps.ld)) if want_log_rho else None
        expo = self.zeros((N, comps.ld)) if want_exponent else None
        ps = int(self.lib.pmc_stats_stride(D))
        nflat = NSCALARS + K * ps + 2 * K
        flat = out if out is not None else self.zeros(nflat)
        assert flat.numel() == nflat
        vsums = flat[NSCALARS + K * ps:] if student else None
        ws = self._workspace(N, K, D)
        stats_pack = pack
        if shift is not None:
            # only the means of a pack matter to the statistics kernel: no matrices, no factorisations
            stats_pack = self._means_pack(shift, K, D)
        if r is None and log_rho is None and expo is None:
            # the E-step proper: one call (pmc_estep_about: the moments about `shift` when given); for small D one
            # kernel, the N x K matrix stays on chip
            fused = bool(self.lib.pmc_estep_is_fused(K, D, comps.kind, int(mode)))
            u = None if fused else self._tilebuf("u", N, K)
            scratch = self._tilebuf("scratch", N, K) if student else None
            _lib.check(self._timed(
                "pmc_estep[fused]" if fused else "pmc_estep", self.lib.pmc_estep_about,
                self._p(x), N, D, self._p(pack), K, comps.kind, int(mode), int(bool(max_init_zero)),
                self._p(sw), self._p(lat), self._p(u), self._p(scratch), self._p(vsums),
                self._p(flat[NSCALARS:]), self._p(flat), self._p(ws), self._p(stats_pack if shift is not None else None),
                self._stream()), "pmc_estep")
            return dict(stats=flat, r=None, log_rho=None, exponent=None)
        u = self._tilebuf("u", N, K)
        scratch = self._tilebuf("scratch", N, K) if student else None
        _lib.check(self._timed(
            "pmc_responsibilities", self.lib.pmc_responsibilities,
            self._p(x), N, D, self._p(pack), K, comps.kind, int(mode), int(bool(max_init_zero)),
            self._p(sw), self._p(lat), self._p(u), self._p(scratch), self._p(vsums), self._p(r),
            self._p(log_rho), self._p(expo), comps.ld, self._p(flat), self._p(ws), self._stream()),
            "pmc_responsibilities")
        _lib.check(self._timed(
            "pmc_sufficient_stats", self.lib.pmc_sufficient_stats,
            self._p(x), N, D, self._p(stats_pack), K, self._p(u), self._p(flat[NSCALARS:]), self._p(ws),
            self._stream()), "pmc_sufficient_stats")
        return dict(stats=flat, r=r, log_rho=log_rho, exponent=expo)

    def estep_from_tiles(self, x, comps, tiles, max_init_zero=False, sample_w=None, out=None):
        """pmc_estep_from_tiles: the Rao-Blackwellised PMC E-step (Gauss or Student-t) of the samples ``tiles``
        were made on, a_nk / rho / gamma from the kept Mahalanobis forms instead of new quadratic forms.
        ``comps`` may be a subset of the mixture behind ``tiles`` (its columns name the positions).  Same
        return value as ``estep``."""
        x = self.asdevice(x)
        N, D = x.shape
        assert comps.kind in (PMC_KIND_GAUSS, PMC_KIND_STUDENT_T) and D == comps.D
        assert tiles.N == N and comps.ld == tiles.K, "kept Mahalanobis forms belong to another sample set / mixture"
        K = comps.K
        pack = self.pack(comps)
        sw = self.asdevice(sample_w).reshape(N) if sample_w is not None else None
        ps = int(self.lib.pmc_stats_stride(D))
        nflat = NSCALARS + K * ps + 2 * K
        flat = out if out is not None else self.zeros(nflat)
        assert flat.numel() == nflat
        vsums = flat[NSCALARS + K * ps:] if comps.kind == PMC_KIND_STUDENT_T else None
        _lib.check(self._timed(
            "pmc_estep_from_tiles", self.lib.pmc_estep_from_tiles,
            self._p(x), N, D, self._p(pack), K, comps.kind, int(bool(max_init_zero)), self._p(sw),
            self._p(tiles.data), tiles.K, self._p(self._tilebuf("u", N, K)), self._p(vsums),
            self._p(flat[NSCALARS:]), self._p(flat), self._p(self._workspace(N, K, D)), self._stream()),
            "pmc_estep_from_tiles")
        return dict(stats=flat, r=None, log_rho=None, exponent=None)

    def weighted_moments(self, x, w):
        """sum w | sum w (x - x_0) | sum w (x - x_0)(x - x_0)^T of weighted samples through the
        statistics kernel (one "component" whose shift is the first sample).  Returns
        (S0, M1 (D), M2 (D x D), shift (D), sum w^2) on the host."""
        x = self.asdevice(x)
        N, D = x.shape
        w = self.asdevice(w).reshape(N)
        shift = self.tohost(x[:1]).reshape(1, D) if N else np.zeros((1, D))
        pack = self._means_pack(shift, 1, D)
        ntile = (N + self.tile - 1) // self.tile
        u = self.zeros(max(ntile, 1) * self.tile)          # tile-major with K = 1: the weight vector
        u[:N] = w
        ps = int(self.lib.pmc_stats_stride(D))
        stats = self.zeros(ps)
        ws = self._workspace(max(N, 1), 1, D)
        _lib.check(self._timed("pmc_sufficient_stats", self.lib.pmc_sufficient_stats,
                               self._p(x), N, D, self._p(pack), 1, self._p(u), self._p(stats), self._p(ws),
                               self._stream()), "pmc_sufficient_stats")
        sums = self.tohost(self.weight_sums(w))
        h = self.tohost(stats)
        il, jl = np.tril_indices(D)
        M2 = np.zeros((D, D))
        M2[il, jl] = h[1 + D:]
        M2[jl, il] = h[1 + D:]
        return float(h[0]), h[1:1 + D].copy(), M2, shift[0], float(sums[2])

    # ------------------------------------------------------------------ the E-step as ONE call of the handle layer
    def ctx(self):
        """this backend's handle-layer context (pmc_init on its device): host arrays in, the reference's conventions out,
        everything K-sized in between on the device"""
        if self._ctx is None:
            h = C.c_void_p()
            _lib.check(self.lib.pmc_init(self.device.index or 0, C.byref(h)), "pmc_init")
            self._ctx = h
        return self._ctx

    def wrap_samples(self, x, sample_w=None):
        """``x`` (N x D device tensor, kept alive by the returned object) as a sample handle of ``ctx()`` -- borrowed, not
        copied; optional sample weights (N, device tensor) likewise.  For ``vb_estep``."""
        x = self.asdevice(x)
        w = self.asdevice(sample_w).reshape(x.shape[0]) if sample_w is not None else None
        self.torch.cuda.current_stream(self.device).synchronize()      # (the context launches on a stream of its own)
        return _WrappedSamples(self, x, w)

    def vb_estep(self, samples, m, W, nu, beta, ln_pi, ln_lambda, shift=None):
        """GaussianInference.E_step (variational.pyx:116-127) as ONE call of the library (pmc_vb_estep): the parameters
        go up in one copy, the pack, the shift pack and the conversion to N_comp / x_mean_comp / S are kernels, one copy
        brings the results back; the far-shift second pass happens inside.  dict(N_comp, x_mean_comp, S, log_q_Z)."""
        m = np.ascontiguousarray(m, dtype=np.float64)
        K, D = m.shape
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        Nk, xbar, S, elq = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(1)
        sh = c(shift).reshape(K, D) if shift is not None else None
        _lib.check(self._timed("pmc_vb_estep", self.lib.pmc_vb_estep, self.ctx(), samples._h, None, K, _dptr(m), _dptr(c(W)), _dptr(c(nu)),
                               _dptr(c(beta)), _dptr(c(ln_pi)), _dptr(c(ln_lambda)), _dptr(sh) if sh is not None else None, _dptr(Nk),
                               _dptr(xbar), _dptr(S), _dptr(elq), None, None), "pmc_vb_estep")
        return dict(N_comp=Nk, x_mean_comp=xbar, S=S, log_q_Z=float(elq[0]))

    def vb_state_supported(self, D):
        """True where the device-resident VB update exists (pmc_vb_state: D <= 64, compiled dimensions)"""
        return 1 <= D <= min(int(self.lib.pmc_vb_max_dim()), int(self.lib.pmc_max_compiled_dim()))

    def vb_state(self, K, D):
        """the K-sized state of a GaussianInference on this backend's device (pmc_vb_state, include/pmc_ctx.h)"""
        return VBState(self, K, D)

    def stats_len(self, K, D):
        """length of the flat statistics buffer of estep()"""
        return NSCALARS + K * int(self.lib.pmc_stats_stride(D)) + 2 * K

    def stats_stride(self, D):
        return int(self.lib.pmc_stats_stride(D))


class VBState(object):
    """Prior, posterior and latest sums of a variational-Bayes fit on the device (``pmc_vb_state``): ``put`` / ``get`` move
    one field, ``step`` runs M-step / E-step / bound there and returns the small block a host needs per iteration."""

    def __init__(self, be, K, D):
        self.be, self.K, self.D, self._h = be, int(K), int(D), None
        h = C.c_void_p()
        _lib.check(be.lib.pmc_vb_state_create(be.ctx(), self.K, self.D, C.byref(h)), "pmc_vb_state_create")
        self._h = h
        self._result = np.empty(int(be.lib.pmc_vb_state_result_len(self.K)))

    def shape(self, name):
        K, D = self.K, self.D
        if name in ("m0", "m", "x_mean_comp", "_shift_prev", "E_m"):
            return (K, D)
        if name in ("inv_W0", "W", "S", "E_W"):
            return (K, D, D)
        return (K,)

    def put(self, name, value):
        a = np.ascontiguousarray(value, dtype=np.float64)
        if a.shape != self.shape(name):
            raise ValueError("field %s: shape %s, expected %s" % (name, a.shape, self.shape(name)))
        _lib.check(self.be.lib.pmc_vb_state_put(self._h, _lib.VB_FIELD_ID[name], _dptr(a)), "pmc_vb_state_put")

    def get(self, name):
        out = np.empty(self.shape(name))
        _lib.check(self.be.lib.pmc_vb_state_get(self._h, _lib.VB_FIELD_ID[name], _dptr(out)), "pmc_vb_state_get")
        return out

    def step(self, samples, mstep=False, estep=False, bound=False, about_prev=False, psi_parts=None):
        """dict(N_comp, far, mean_finite, S_any_finite, log_q_Z, bound (8: L(Q) and its terms)) -- None for an M-step alone
        (queued; a matrix that does not factorise is reported by the next step that returns a block).  ``psi_parts``
        (2 K): the caller's own [E[ln pi] | sum psi + D ln 2] for the E-step (pmc_ctx.h), None = the device's psi."""
        flags = (_lib.VB_DO_MSTEP if mstep else 0) | (_lib.VB_DO_ESTEP if estep else 0) | (_lib.VB_DO_BOUND if bound else 0) | \
            (_lib.VB_ABOUT_PREV if about_prev else 0)
        want = estep or bound
        if psi_parts is not None:
            psi_parts = np.ascontiguousarray(psi_parts, dtype=np.float64)
            assert psi_parts.shape == (2 * self.K,)
        _lib.check(self.be._timed("pmc_vb_state_step", self.be.lib.pmc_vb_state_step, self._h, samples._h if samples is not None else None,
                                  flags, _dptr(psi_parts) if psi_parts is not None else None, _dptr(self._result) if want else None),
                   "pmc_vb_state_step")
        if not want:
            return None
        K, r = self.K, self._result
        return dict(N_comp=r[:K].copy(), far=r[K:2 * K], mean_finite=r[2 * K:3 * K], S_any_finite=r[3 * K:4 * K],
                    log_q_Z=float(r[4 * K]), bound=r[8 * K + 8:8 * K + 16].copy())

    def run(self, samples, max_iterations, old_bound, prune_threshold, rel_tol, abs_tol, about_prev, n_comp, psi_parts=None):
        """``pmc_vb_state_run``: updates until the reference's convergence rules hold, a component falls below
        ``prune_threshold``, the block needs a look, or ``max_iterations`` are done.  ``psi_parts(N_comp) -> 2 K numbers``
        is called once per update (None: the device's psi).  Returns dict(done, reason, decreased, about_prev, bound,
        old_bound) + the last block's entries as ``step`` returns them."""
        K = self.K
        n_comp = np.ascontiguousarray(n_comp, dtype=np.float64)
        assert n_comp.shape == (K,)
        error = []

        def callback(user, k, n_ptr, out_ptr):
            try:
                out = np.ctypeslib.as_array(out_ptr, shape=(2 * k,))
                out[:] = psi_parts(np.ctypeslib.as_array(n_ptr, shape=(k,)))
            except BaseException as exc:                  # (an exception must not unwind through the library's frames)
                error.append(exc)
                np.ctypeslib.as_array(out_ptr, shape=(2 * k,))[:] = np.nan
        cb = _lib.VB_PSI_FN(callback) if psi_parts is not None else None
        info, bounds = (C.c_int * 4)(), np.zeros(2)
        rc = self.be._timed("pmc_vb_state_run", self.be.lib.pmc_vb_state_run, self._h, samples._h, int(max_iterations), float(old_bound),
                            float(prune_threshold), float(rel_tol), float(abs_tol), 1 if about_prev else 0, _dptr(n_comp),
                            C.cast(cb, C.c_void_p) if cb is not None else None, None, _dptr(self._result), info, _dptr(bounds))
        if error:
            raise error[0]
        done = int(info[0])
        out = dict(done=done, reason=int(info[1]), decreased=int(info[2]), about_prev=bool(info[3]), bound=float(bounds[0]),
                   old_bound=float(bounds[1]), rc=rc)
        if done:
            r = self._result
            out.update(N_comp=r[:K].copy(), far=r[K:2 * K], mean_finite=r[2 * K:3 * K], S_any_finite=r[3 * K:4 * K],
                       log_q_Z=float(r[4 * K]), bound_terms=r[8 * K + 8:8 * K + 16].copy())
        return out

    def close(self):
        if self._h is not None:
            self.be.lib.pmc_vb_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass


class _WrappedSamples(object):
    """a device tensor (and optional sample weights) lent to the backend's handle-layer context"""

    def __init__(self, be, x, w):
        self.be, self.x, self.w, self._h = be, x, w, None
        h = C.c_void_p()
        _lib.check(be.lib.pmc_samples_wrap(be.ctx(), C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], C.byref(h)),
                   "pmc_samples_wrap")
        self._h = h
        if w is not None:
            _lib.check(be.lib.pmc_samples_wrap_sample_weights(h, C.c_void_p(w.data_ptr())), "pmc_samples_wrap_sample_weights")

    def __del__(self):
        try:
            if self._h is not None:
                self.be.lib.pmc_samples_free(self._h)
                self._h = None
        except Exception:  # pragma: no cover
            pass

    def __deepcopy__(self, memo):
        return _WrappedSamples(self.be, self.x, self.w)


_default = None


def set_default_backend(backend):
    """Install the process-wide default backend (tests inject their checker here)."""
    global _default
    _default = backend


def get_backend(backend=None):
    """The backend to use: an explicit one, the installed default, or a new HipBackend.
    Raises HipLibraryError when the HIP path is unavailable -- there is no CPU fallback."""
    global _default
    if backend is not None:
        return backend
    if _default is None:
        _default = HipBackend()
    return _default
