"""ctypes binding of libpmc_hip.so (C ABI: include/pmc_hip.h).

The library is the product path.  There is no CPU fallback: if it cannot be loaded every
operation raises ``HipLibraryError`` with the reason.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PMC_HIP_LIBRARY: development aid -- an A/B variant of the library built by `PMC_VARIANT=... python -m pypmc_amd.build`)
LIB_PATH = os.environ.get("PMC_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libpmc_hip.so")

PMC_KIND_GAUSS, PMC_KIND_STUDENT_T, PMC_KIND_VB = 0, 1, 2
PMC_RESP_VB, PMC_RESP_PMC_RB, PMC_RESP_PMC_LATENT = 0, 1, 2
PMC_OK, PMC_EINVAL, PMC_ENOTPOSDEF, PMC_EHIP, PMC_ENODEVICE = 0, -1, -2, -3, -4
NSCALARS = 8
MAX_DIM = 1024    # largest sample dimension (== pmc_max_dim(), tested): per-dimension kernels up to 64, the
                  # run-time-dimension unit (csrc/pmc_big.hip) beyond


def check_dim(dim):
    """Sample dimensions beyond the kernels' limit are refused where a density is built, not at its
    first evaluation (the reference's loops take any length, pypmc/tools/_linalg.pyx:32-37)."""
    if dim > MAX_DIM:
        raise ValueError("pypmc_amd's gfx950 kernels take sample dimensions up to %d "
                         "(got %d); there is no CPU fallback" % (MAX_DIM, dim))


class HipLibraryError(RuntimeError):
    """libpmc_hip.so is missing / unloadable, or a call into it failed."""


class NotPositiveDefinite(HipLibraryError, np.linalg.LinAlgError):
    """A precision matrix handed to pmc_pack_components is not positive definite.  Also a
    ``numpy.linalg.LinAlgError``: callers written against the reference catch that
    (pypmc/mix_adapt/pmc.pyx:227-244, pypmc/density/gauss.pyx:40-48)."""


_vp, _i64, _i32, _int = C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_int
_dp = C.POINTER(C.c_double)

# name -> (restype, argtypes); must list every symbol include/pmc_hip.h declares
SIGNATURES = {
    "pmc_abi_version": (_int, []),
    "pmc_last_error": (C.c_char_p, []),
    "pmc_device_count": (_int, []),
    "pmc_device_arch": (_int, [_int, C.c_char_p, C.c_size_t]),
    "pmc_max_dim": (_int, []),
    "pmc_max_compiled_dim": (_int, []),
    "pmc_padded_dim": (_int, [_int]),
    "pmc_pack_stride": (_i64, [_int]),
    "pmc_tile": (_int, []),
    "pmc_pack_components": (_int, [_int, _int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _i32, _dp]),
    "pmc_pack_means": (_int, [_int, _int, _dp, _dp]),
    "pmc_pack_components_device": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmc_pack_status": (_int, [_int, _dp]),
    "pmc_pack_means_device": (_int, [_int, _int, _vp, _vp, _vp]),
    "pmc_convert_stats_len": (_i64, [_int, _int]),
    "pmc_convert_stats_device": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmc_stream_release": (_int, [_vp]),
    "pmc_workspace_bytes": (_i64, [_i64, _int, _int]),
    "pmc_tile_buffer_len": (_i64, [_i64, _int]),
    "pmc_stats_stride": (_i64, [_int]),
    "pmc_mixture_logpdf": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _vp, _vp, _i64, _vp, _vp,
                                  _vp, _vp, _vp, _vp]),
    "pmc_importance_weights": (_int, [_vp, _i64, _int, _vp, _int, _int, _vp, _int, _int, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp]),
    "pmc_weight_sums": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "pmc_propose": (_int, [_vp, _vp, _vp, _vp, _int, _int, _i64, _i64, C.c_uint64, _vp, _vp, _vp]),
    "pmc_logsumexp2d": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "pmc_combine_weights": (_int, [_vp, _i64, _int, _vp, _int, _vp, C.c_double, _int, _vp, _vp, _vp]),
    "pmc_responsibilities": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "pmc_sufficient_stats": (_int, [_vp, _i64, _int, _vp, _int, _vp, _vp, _vp, _vp]),
    "pmc_comm_unique_id": (_int, [_vp]),
    "pmc_comm_init": (_int, [_int, _int, _vp, _int, C.POINTER(_vp)]),
    "pmc_comm_rank": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pmc_comm_allreduce_sum": (_int, [_vp, _vp, _i64, _vp]),
    "pmc_comm_destroy": (_int, [_vp]),
    "pmc_p2p_create": (_int, [_int, _int, _i64, _int, C.POINTER(_vp)]),
    "pmc_p2p_handle": (_int, [_vp, _vp]),
    "pmc_p2p_connect": (_int, [_vp, _vp]),
    "pmc_p2p_allreduce_sum": (_int, [_vp, _vp, _i64, _vp]),
    "pmc_p2p_status": (_int, [_vp, _vp]),
    "pmc_p2p_info": (_int, [_vp, C.c_char_p, C.c_size_t]),
    "pmc_p2p_destroy": (_int, [_vp]),
    "pmc_timing_enable": (_int, [_int]),
    "pmc_get_timings": (_int, [_vp, _int, C.POINTER(C.c_int)]),
    "pmc_configure": (_int, [C.c_char_p, C.c_double]),
    "pmc_option_get": (_int, [C.c_char_p, C.POINTER(C.c_double)]),
    "pmc_option_default": (_int, [C.c_char_p, C.POINTER(C.c_double)]),
    "pmc_estep_is_fused": (_int, [_int, _int, _int, _int]),
    "pmc_estep": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmc_estep_about": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp]),
    "pmc_maha_tiles_size": (_i64, [_i64, _int]),
    "pmc_mixture_logpdf_keep": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _vp, _vp, _i64, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp]),
    "pmc_importance_weights_keep": (_int, [_vp, _i64, _int, _vp, _int, _int, _vp, _int, _int, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp]),
    "pmc_importance_weights_emit": (_int, [_vp, _i64, _int, _vp, _int, _int, _vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _vp]),
    "pmc_estep_from_u": (_int, [_vp, _i64, _int, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "pmc_gscale_len": (_i64, [_i64, _int]),
    "pmc_maha_gemm_tiles": (_int, [_i64, _int, _int]),
    "pmc_maha_gemm_report": (_int, [_vp, _i64, _int, _int, _vp, _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pmc_importance_weights_emit_grouped": (_int, [_vp, _i64, _int, _vp, _int, _int, _vp, _int, _int, _vp, _vp, _vp, _vp,
                                                   _vp, _vp, _vp, _vp, _vp]),
    "pmc_importance_weights_emit_live": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _vp, _int, _int, _vp, _vp, _vp, _vp,
                                                _vp, _vp, _vp, _vp, _vp]),
    "pmc_estep_from_u_grouped": (_int, [_vp, _i64, _int, _vp, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "pmc_estep_from_tiles": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp,
                                    _vp]),
    # the K-sized half of a VB iteration on the device (struct pmc_vb_fields * travels as a plain pointer)
    "pmc_spd_inverse_len": (_i64, [_int, _int]),
    "pmc_spd_inverse_device": (_int, [_int, _int, _vp, _vp, _vp]),
    "pmc_vb_max_dim": (_int, []),
    "pmc_vb_mstep_device": (_int, [_int, _int, _vp, _vp, _vp]),
    "pmc_vb_mstep_status": (_int, [_int, _dp]),
    "pmc_vb_expectations_device": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp]),
    "pmc_vb_pack_device": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmc_vb_convert_after_device": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmc_vb_small_len": (_i64, [_int]),
    "pmc_vb_after_device": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmc_vb_newshift_device": (_int, [_int, _int, _vp, _vp, _vp, _vp]),
    "pmc_vb_bound_scratch_len": (_i64, [_int]),
    "pmc_vb_bound_device": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp]),
    "pmc_host_digamma": (C.c_double, [C.c_double]),
    "pmc_host_lgamma": (C.c_double, [C.c_double]),
}



# the handle layer (include/pmc_ctx.h): host pointers in and out; handles are opaque pointers
_pp = C.POINTER(C.c_void_p)
_ip = C.POINTER(C.c_int64)
CTX_SIGNATURES = {
    "pmc_init": (_int, [_int, _pp]),
    "pmc_init_devices": (_int, [_int, C.POINTER(C.c_int), _pp]),
    "pmc_ctx_device_count": (_int, [_vp]),
    "pmc_ctx_devices": (_int, [_vp, C.POINTER(C.c_int), _int]),
    "pmc_samples_shard": (_int, [_vp, _int, _ip, _ip]),
    "pmc_ctx_join": (_int, [_vp, _int, _int, _vp]),
    "pmc_ctx_p2p_open": (_int, [_vp, _int, _int, _i64, _vp]),
    "pmc_ctx_p2p_connect": (_int, [_vp, _vp]),
    "pmc_shutdown": (_int, [_vp]),
    "pmc_mixture_create": (_int, [_vp, _int, _int, _int, _dp, _dp, _dp, _dp, _dp, _pp]),
    "pmc_mixture_update": (_int, [_vp, _dp, _dp, _dp, _dp, _dp]),
    "pmc_mixture_destroy": (_int, [_vp]),
    "pmc_samples_upload": (_int, [_vp, _dp, _i64, _int, _pp]),
    "pmc_samples_generate": (_int, [_vp, _vp, _dp, _ip, C.c_uint64, _i64, _pp]),
    "pmc_samples_wrap": (_int, [_vp, _vp, _i64, _int, _pp]),
    "pmc_samples_set_sample_weights": (_int, [_vp, _dp]),
    "pmc_samples_wrap_sample_weights": (_int, [_vp, _vp]),
    "pmc_samples_count": (_i64, [_vp]),
    "pmc_samples_download": (_int, [_vp, _dp]),
    "pmc_samples_origin": (_int, [_vp, _ip]),
    "pmc_samples_free": (_int, [_vp]),
    "pmc_mix_logpdf": (_int, [_vp, _vp, _dp, _dp]),
    "pmc_mix_logpdf_components": (_int, [_vp, _vp, _i32, _int, _dp]),
    "pmc_ctx_configure": (_int, [_vp, C.c_char_p, C.c_double]),
    "pmc_ctx_timing_enable": (_int, [_vp, _int]),
    "pmc_ctx_get_timings": (_int, [_vp, _vp, _int, C.POINTER(C.c_int)]),
    "pmc_is_weights": (_int, [_vp, _vp, _dp, _vp, _dp, _dp, _dp]),
    "pmc_vb_estep": (_int, [_vp, _vp, _dp, _int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "pmc_pmc_update_stats": (_int, [_vp, _vp, _vp, _dp, _int, _ip, _int, _dp, _dp, _dp, _dp, _dp, _dp]),
    "pmc_weighted_moments": (_int, [_vp, _vp, _dp, _int, _dp, _dp]),
    "pmc_host_convert_stats": (_int, [_int, _int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    "pmc_host_chol_inv_det_batch": (_int, [_int, _int, _dp, _vp, _vp, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    "pmc_vb_state_create": (_int, [_vp, _int, _int, _pp]),
    "pmc_vb_state_destroy": (_int, [_vp]),
    "pmc_vb_state_put": (_int, [_vp, _int, _dp]),
    "pmc_vb_state_get": (_int, [_vp, _int, _dp]),
    "pmc_vb_state_result_len": (_i64, [_int]),
    "pmc_vb_state_step": (_int, [_vp, _vp, _int, _dp, _dp]),
    "pmc_vb_state_run": (_int, [_vp, _vp, _int, C.c_double, C.c_double, C.c_double, C.c_double, _int, _dp, _vp, _vp, _dp,
                                C.POINTER(C.c_int), _dp]),
}
VB_PSI_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))
VB_RUN_CAP, VB_RUN_CONVERGED, VB_RUN_PRUNE, VB_RUN_LOOK = 0, 1, 2, 3

# enum pmc_vb_field / the step flags (include/pmc_ctx.h)
VB_FIELDS = ("alpha0", "beta0", "nu0", "m0", "inv_W0", "log_det_W0", "alpha", "beta", "nu", "m", "W", "log_det_W",
             "expectation_det_ln_lambda", "expectation_ln_pi", "N_comp", "x_mean_comp", "S", "_shift_prev",
             "E_m", "E_W", "E_beta", "E_nu", "E_ln_pi", "E_ln_lambda")
VB_FIELD_ID = dict((n, i) for i, n in enumerate(VB_FIELDS))
VB_DO_MSTEP, VB_DO_ESTEP, VB_DO_BOUND, VB_ABOUT_PREV = 1, 2, 4, 8


class Timing(C.Structure):
    """struct pmc_timing (include/pmc_hip.h)"""
    _fields_ = [("name", C.c_char * 48), ("calls", C.c_int), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


_lib = None
_load_error = None


def load():
    """Return the loaded library or raise HipLibraryError (never falls back)."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise HipLibraryError(_load_error)
    if not os.path.exists(LIB_PATH):
        _load_error = ("%s not found: build it with `python -m pypmc_amd.build` "
                       "(there is no CPU fallback)" % LIB_PATH)
        raise HipLibraryError(_load_error)
    try:
        # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  Import torch first so
        # that libpmc_hip.so binds to that same runtime instance: two HIP runtimes in one process
        # do not see each other's devices, streams or allocations.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(CTX_SIGNATURES.items()):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    except (OSError, AttributeError, ImportError) as exc:
        _load_error = "cannot load %s: %s" % (LIB_PATH, exc)
        raise HipLibraryError(_load_error)
    _lib = lib
    return _lib


def last_error():
    msg = load().pmc_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status, what=""):
    """Turn a negative status into an exception carrying pmc_last_error()."""
    if status is not None and status < 0:
        msg = "%s failed (%d): %s" % (what or "libpmc_hip call", status, last_error())
        if status == PMC_ENOTPOSDEF:
            raise NotPositiveDefinite(msg)
        raise HipLibraryError(msg)
    return status
