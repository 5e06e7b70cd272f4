"""Round 5: the K-sized host steps around an E-step as kernels -- pmc_pack_components_device, pmc_pack_means_device,
pmc_convert_stats_device (include/pmc_hip.h) -- held to the host functions they stand in for BIT FOR BIT
(pmc_pack_components: the Cholesky factorisations of pypmc's W_k / inv_sigma_k, variational.pyx:116-136;
pmc_host_convert_stats: N_comp / x_mean_comp / S, variational.pyx:699-932, pmc.pyx:188-222)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    return HipBackend()


def spd(rs, K, D, cond=1.0):
    A = rs.normal(size=(K, D, D))
    P = np.einsum('kij,klj->kil', A, A) / D + 0.5 * np.eye(D)
    if cond > 1:
        s = np.logspace(0, np.log10(cond) / 2, D)
        P = P * s[None, :, None] * s[None, None, :]
    return np.ascontiguousarray(0.5 * (P + P.transpose(0, 2, 1)))


@pytest.mark.parametrize("K,D", [(1, 1), (3, 2), (5, 7), (4, 9), (64, 20), (32, 24), (7, 31), (128, 40), (9, 48), (6, 57), (11, 64),
                                 (1000, 3)])
def test_pack_components_on_the_device_is_the_host_pack(be, K, D):
    import torch
    lib = be.lib
    rs = np.random.RandomState(K + 100 * D)
    mu, prec = rs.normal(size=(K, D)), spd(rs, K, D, cond=1e6 if D % 2 else 1.0)
    c = rs.normal(size=(4, K))
    w = rs.uniform(size=K)
    col = rs.permutation(K).astype(np.int32)
    shift = rs.normal(size=(K, D))
    stride = lib.pmc_pack_stride(D)
    for with_consts in (True, False):
        host, hmeans = np.empty(K * stride), np.empty(K * stride)
        args = [dp(np.ascontiguousarray(v)) for v in c] + [dp(w), col.ctypes.data_as(C.POINTER(C.c_int32))] if with_consts \
            else [None] * 6
        assert lib.pmc_pack_components(K, D, dp(mu), dp(prec), *args, dp(host)) == 0
        assert lib.pmc_pack_means(K, D, dp(shift), dp(hmeans)) == 0
        t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(be.device) if a is not None else None
        d_mu, d_prec, d_shift = t(mu), t(prec), t(shift)
        dc = [t(v) for v in c] + [t(w), t(col)] if with_consts else [None] * 6
        pack = torch.full((K * stride,), 7.0, dtype=torch.float64, device=be.device)         # (poisoned: every slot is written)
        mpack = torch.full((K * stride,), 7.0, dtype=torch.float64, device=be.device)
        status = torch.full((2 * K,), 7.0, dtype=torch.float64, device=be.device)
        p = be._p
        assert lib.pmc_pack_components_device(K, D, p(d_mu), p(d_prec), *[p(v) for v in dc], p(pack), p(status), p(d_shift),
                                              p(mpack), be._stream()) == 0, lib.pmc_last_error()
        hs = be.tohost(status)
        assert lib.pmc_pack_status(K, dp(hs)) == 0 and not hs[:K].any()
        got, gotm = be.tohost(pack), be.tohost(mpack)
        assert got.tobytes() == host.tobytes(), "device pack differs from pmc_pack_components"
        assert gotm.tobytes() == hmeans.tobytes(), "device shift pack differs from pmc_pack_means"
        mp2 = torch.full((K * stride,), 7.0, dtype=torch.float64, device=be.device)
        assert lib.pmc_pack_means_device(K, D, p(d_shift), p(mp2), be._stream()) == 0
        assert be.tohost(mp2).tobytes() == hmeans.tobytes()


def test_a_matrix_that_does_not_factorise_is_named(be):
    import torch
    from pypmc_amd import _lib
    lib = be.lib
    K, D = 6, 20
    rs = np.random.RandomState(1)
    mu, prec = rs.normal(size=(K, D)), spd(rs, K, D)
    prec[4] = -np.eye(D)
    prec[2, 7, 7] = np.nan
    stride = lib.pmc_pack_stride(D)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(be.device)
    pack, status = torch.zeros(K * stride, dtype=torch.float64, device=be.device), torch.zeros(2 * K, dtype=torch.float64, device=be.device)
    p = be._p
    assert lib.pmc_pack_components_device(K, D, p(t(mu)), p(t(prec)), None, None, None, None, None, None, p(pack), p(status), None,
                                          None, be._stream()) == 0
    hs = be.tohost(status)
    assert lib.pmc_pack_status(K, dp(hs)) == _lib.PMC_ENOTPOSDEF
    msg = _lib.last_error()
    host = np.empty(K * stride)
    assert lib.pmc_pack_components(K, D, dp(mu), dp(prec), None, None, None, None, None, None, dp(host)) == _lib.PMC_ENOTPOSDEF
    assert msg.split("(pivot")[0] == _lib.last_error().split("(pivot")[0] and "component 2" in msg     # the lowest failing one
    assert hs[4] == 1.0 and hs[2] > 0 and not hs[[0, 1, 3, 5]].any()
    # dimensions beyond the compiled ones stay with the host builder
    assert lib.pmc_pack_components_device(2, 70, p(pack), p(pack), None, None, None, None, None, None, p(pack), p(status), None, None,
                                          be._stream()) == _lib.PMC_EINVAL
    # ... and through the handle layer: PMC_ENOTPOSDEF, the component named
    ctx, s = C.c_void_p(), C.c_void_p()
    assert lib.pmc_init(0, C.byref(ctx)) == 0
    x = np.ascontiguousarray(rs.normal(size=(500, D)))
    assert lib.pmc_samples_upload(ctx, dp(x), 500, D, C.byref(s)) == 0
    o = [np.empty(K), np.empty((K, D)), np.empty((K, D, D))]
    rc = lib.pmc_vb_estep(ctx, s, None, K, dp(mu), dp(prec), dp(np.full(K, D + 1.)), dp(np.ones(K)), dp(np.zeros(K)), dp(np.zeros(K)),
                          None, dp(o[0]), dp(o[1]), dp(o[2]), None, None, None)
    assert rc == _lib.PMC_ENOTPOSDEF and "component 2" in _lib.last_error()
    lib.pmc_samples_free(s)
    assert lib.pmc_shutdown(ctx) == 0


@pytest.mark.parametrize("K,D,student", [(1, 1, False), (5, 3, True), (64, 20, False), (128, 40, True), (1030, 2, False), (7, 64, True)])
def test_convert_stats_on_the_device_is_the_host_conversion(be, K, D, student):
    import torch
    from pypmc_amd.mix_adapt._stats import convert_stats
    lib = be.lib
    rs = np.random.RandomState(K + D)
    ps = 1 + D + D * (D + 1) // 2
    for trial in range(4):
        flat = rs.normal(size=8 + K * ps + 2 * K)
        body = flat[8:8 + K * ps].reshape(K, ps)
        body[:, 0] = np.abs(body[:, 0]) * rs.choice([1, 1, 1e-3, 1e3, 0], size=K)
        body[:, 1:1 + D] *= rs.choice([1., 1e-2, 1e2])
        if trial == 1:
            flat[8 + rs.randint(K * ps)] = np.nan
        if trial == 2:
            flat[8 + rs.randint(K * ps)] = np.inf
        if trial == 3:                                       # a component far from its shift
            body[0, 1:1 + D] = 1e3 * abs(body[0, 0])
        vs = flat[8 + K * ps:].reshape(K, 2)
        vs[:, 0] = np.abs(vs[:, 0]) + 0.1
        shift = rs.normal(size=(K, D))
        with np.errstate(all='ignore'):
            sc, S0, M1, mean, cov, far, V1, _ = convert_stats(flat, K, D, shift, n_cov='vsum0' if student else None)
        n = int(lib.pmc_convert_stats_len(K, D))
        assert n == K * (2 + 2 * D + D * D) + 8
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(be.device)
        d_flat, d_shift, d_ncov = t(flat), t(shift), (t(vs[:, 0]) if student else None)
        out = torch.full((n,), 7.0, dtype=torch.float64, device=be.device)
        p = be._p
        assert lib.pmc_convert_stats_device(K, D, p(d_flat[8:]), p(d_shift), p(d_ncov), p(d_flat), p(out), be._stream()) == 0
        h = be.tohost(out)
        o = 0
        for ref, size in ((S0, K), (M1, K * D), (mean, K * D), (cov, K * D * D)):
            assert np.array_equal(h[o:o + size], np.asarray(ref).reshape(-1), equal_nan=True), (trial, o)
            o += size
        assert bool(h[o:o + K].any()) == bool(far), trial
        assert set(np.unique(h[o:o + K])) <= {0.0, 1.0}
        assert np.array_equal(h[o + K:], flat[:8], equal_nan=True)


# ---- round 6: the PMC update's K factorisations and the pack of a big mixture on the device -------------------------------
@pytest.mark.parametrize("K,D,cond", [(1, 1, 1), (3, 2, 1), (17, 5, 1e4), (128, 40, 1), (128, 40, 1e6), (40, 64, 1e3), (300, 20, 1e2)])
def test_chol_inv_det_batch_on_the_device_agrees_with_lapack(be, K, D, cond):
    """backend.chol_inv_det_batch (pmc_spd_inverse_device: potrf / potri's algorithm, one wavefront per matrix) against
    tools._linalg.chol_inv_det_batch (LAPACK; pypmc/tools/_linalg.pyx:41-95): to rounding, scaled by the condition number"""
    from pypmc_amd.tools._linalg import chol_inv_det_batch
    rs = np.random.RandomState(7 * K + D)
    sig = spd(rs, K, D, cond=cond)
    L, I, ld = be.chol_inv_det_batch(sig)
    L0, I0, ld0 = chol_inv_det_batch(sig, check_symmetric=False)
    scale = lambda a: np.abs(a).max(axis=(1, 2), keepdims=True)
    assert (np.abs(L - L0) <= 1e-14 * np.sqrt(cond) * scale(L0)).all()
    assert (np.abs(I - I0) <= 4e-15 * cond * scale(I0)).all()
    np.testing.assert_allclose(ld, ld0, rtol=1e-13, atol=1e-12)
    np.testing.assert_array_equal(I, I.transpose(0, 2, 1))
    assert (np.triu(L, 1) == 0).all()
    # and the inverse IS an inverse
    assert np.abs(np.einsum('kij,kjl->kil', sig, I) - np.eye(D)).max() < 1e-12 * cond


def test_chol_inv_det_batch_on_the_device_reports_failures(be):
    rs = np.random.RandomState(3)
    sig = spd(rs, 6, 5)
    sig[4] = -sig[4]
    with pytest.raises(np.linalg.LinAlgError):
        be.chol_inv_det_batch(sig)
    sig = spd(rs, 6, 5)
    sig[2, 1, 1] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        be.chol_inv_det_batch(sig)
    assert be.chol_inv_det_batch(spd(rs, 2, 65)) is None           # beyond one wavefront per matrix: the caller's LAPACK


@pytest.mark.parametrize("K,D,kind", [(128, 40, "gauss"), (64, 32, "student"), (200, 20, "gauss"), (5, 3, "gauss")])
def test_the_pack_of_a_big_mixture_built_on_the_device_is_the_host_pack(be, K, D, kind):
    """backend._build_pack takes the device's builder from DEVICE_LINALG_FROM matrix elements on: the same bits"""
    from pypmc_amd.backend import ComponentSet
    from pypmc_amd._lib import PMC_KIND_GAUSS, PMC_KIND_STUDENT_T
    rs = np.random.RandomState(K + D)
    cs = ComponentSet(PMC_KIND_STUDENT_T if kind == "student" else PMC_KIND_GAUSS, rs.normal(size=(K, D)), spd(rs, K, D),
                      c0=rs.normal(size=K), c1=rs.normal(size=K), c2=rs.uniform(size=K), c3=rs.uniform(1, 9, size=K),
                      weight=rs.uniform(size=K), column=rs.permutation(K))
    dev = be._build_pack_device(cs)
    assert dev is not None
    saved, type(be).DEVICE_LINALG_FROM = type(be).DEVICE_LINALG_FROM, 1 << 62
    try:
        host = be._build_pack(cs)
    finally:
        type(be).DEVICE_LINALG_FROM = saved
    np.testing.assert_array_equal(dev.cpu().numpy(), host.cpu().numpy())
    bad = ComponentSet(PMC_KIND_GAUSS, rs.normal(size=(K, D)), -spd(rs, K, D))
    with pytest.raises(np.linalg.LinAlgError):
        be._build_pack_device(bad)
