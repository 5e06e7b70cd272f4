"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/pmc_hip.h
declares, its host-only entry points work, and without a GPU the product path fails loudly
instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="pmc_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pypmc_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), "libpmc_hip.so does not export " + name
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"
    assert lib.pmc_abi_version() == 2
    # the handle layer on top of it (include/pmc_ctx.h)
    ctx_names = declared_symbols("pmc_ctx.h")
    assert len(ctx_names) >= 16 and not set(ctx_names) & set(names)
    for name in ctx_names:
        assert hasattr(lib, name), "libpmc_hip.so does not export " + name
    assert sorted(_lib.CTX_SIGNATURES) == ctx_names, "ctypes binding and pmc_ctx.h disagree"


def test_handle_layer_fails_loudly_without_a_gpu():
    """pmc_init on a box without a device: a status and a message, no crash, no fallback"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pypmc_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.pmc_init(0, C.byref(ctx))
    assert rc < 0 and not ctx.value and _lib.last_error()
    assert lib.pmc_shutdown(None) == 0 and lib.pmc_mixture_destroy(None) == 0 and lib.pmc_samples_free(None) == 0


def test_host_side_entry_points():
    from pypmc_amd import _lib
    lib = _lib.load()
    assert lib.pmc_max_dim() == 1024 and lib.pmc_max_compiled_dim() == 64 and lib.pmc_tile() == 64
    assert lib.pmc_padded_dim(20) == 20 and lib.pmc_padded_dim(9) == 10 and lib.pmc_padded_dim(33) == 40
    assert lib.pmc_padded_dim(65) == 65 and lib.pmc_padded_dim(1024) == 1024          # the run-time-dimension unit
    assert lib.pmc_padded_dim(1025) < 0 and "not supported" in _lib.last_error()
    assert lib.pmc_padded_dim(0) < 0
    assert lib.pmc_stats_stride(20) == 1 + 20 + 210
    assert lib.pmc_tile_buffer_len(65, 3) == 2 * 3 * 64
    assert lib.pmc_workspace_bytes(10 ** 6, 32, 20) > 0
    # pmc_pack_components: precision = R^T R, padding, constants, columns
    rs = np.random.RandomState(0)
    K, D = 3, 9
    Dp = lib.pmc_padded_dim(D)
    stride = lib.pmc_pack_stride(D)
    assert stride % 8 == 0 and stride >= Dp + Dp * (Dp + 1) // 2 + 6
    A = rs.normal(size=(K, D, D))
    prec = np.einsum('kij,klj->kil', A, A) + 0.1 * np.eye(D)
    mu = rs.normal(size=(K, D))
    c = rs.normal(size=(4, K))
    w = rs.uniform(size=K)
    col = np.array([4, 0, 2], dtype=np.int32)
    pack = np.empty(K * stride)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib.pmc_pack_components(K, D, dp(mu), dp(prec), dp(c[0].copy()), dp(c[1].copy()), dp(c[2].copy()),
                                 dp(c[3].copy()), dp(w), col.ctypes.data_as(C.POINTER(C.c_int32)), dp(pack))
    assert rc == 0
    T = Dp * (Dp + 1) // 2
    iu = np.triu_indices(Dp)
    for k in range(K):
        pk = pack[k * stride:(k + 1) * stride]
        np.testing.assert_array_equal(pk[:D], mu[k])
        assert (pk[D:Dp] == 0).all()
        R = np.zeros((Dp, Dp))
        R[iu] = pk[Dp:Dp + T]
        np.testing.assert_allclose(R[:D, :D].T.dot(R[:D, :D]), prec[k], rtol=1e-12, atol=1e-13)
        assert (R[D:, :] == 0).all() and (R[:, D:] == 0).all()
        np.testing.assert_array_equal(pk[Dp + T:Dp + T + 4], c[:, k])
        assert pk[Dp + T + 4] == w[k]
        assert np.frombuffer(pk[Dp + T + 5:Dp + T + 6].tobytes(), dtype=np.int64)[0] == col[k]
    bad = prec.copy()
    bad[1] = -np.eye(D)
    assert lib.pmc_pack_components(K, D, dp(mu), dp(bad), None, None, None, None, None, None, dp(pack)) == -2
    assert "component 1" in _lib.last_error()


def test_workspace_bytes_is_monotone_in_the_component_count():
    """One workspace serves every launch of a call that involves up to K components (a proposal and a larger target,
    pmc_importance_weights): its size must not shrink when K grows (advice r4: the chunk counts of the statistics kernels
    halve where ceil(K / 32) steps up, and the regions of a K = 32 launch did not fit the workspace of K = 33)."""
    from pypmc_amd import _lib
    lib = _lib.load()
    for D in (2, 5, 20, 24, 32, 40, 48, 64, 70):
        for N in (1, 1000, 65536, 200000, 10 ** 6):
            sizes = [lib.pmc_workspace_bytes(N, K, D) for K in range(1, 140)]
            assert all(s > 0 for s in sizes)
            assert all(b >= a for a, b in zip(sizes, sizes[1:])), (D, N)


def test_evaluate_once_entry_points_check_their_arguments():
    """pmc_maha_tiles_size and the argument checks of pmc_estep_from_tiles (they return before any launch)."""
    from pypmc_amd import _lib
    lib = _lib.load()
    assert lib.pmc_maha_tiles_size(1, 3) == 3 * 64 and lib.pmc_maha_tiles_size(65, 2) == 2 * 2 * 64
    assert lib.pmc_maha_tiles_size(0, 5) == 0 and lib.pmc_maha_tiles_size(-1, 5) == 0 and lib.pmc_maha_tiles_size(10, 0) == 0
    buf = np.zeros(64)
    p = buf.ctypes.data_as(C.c_void_p)          # never dereferenced: every call below fails its checks first
    args = lambda **kw: [kw.get("x", p), kw.get("N", 10), kw.get("D", 2), kw.get("pack", p), kw.get("K", 2),
                         kw.get("kind", 0), 0, None, kw.get("tiles", p), kw.get("K_tiles", 2), kw.get("u", p),
                         kw.get("vsums", None), kw.get("stats", p), kw.get("scalars", p), kw.get("ws", p), None]
    for bad, needle in ((dict(K=0), "bad N/K"), (dict(K_tiles=0), "bad N/K"), (dict(pack=None), "bad N/K"),
                        (dict(stats=None), "bad N/K"), (dict(ws=None), "bad N/K"), (dict(N=-1), "bad N/K"),
                        (dict(kind=2), "kind must be GAUSS or STUDENT_T"), (dict(kind=1), "Student-t needs d_vsums"),
                        (dict(D=1025), "not supported"), (dict(x=None), "is NULL"), (dict(tiles=None), "is NULL"),
                        (dict(u=None), "is NULL")):
        assert lib.pmc_estep_from_tiles(*args(**bad)) == -1, bad
        assert needle in _lib.last_error(), (bad, _lib.last_error())


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pypmc_amd.backend import HipBackend, HipLibraryError
    with pytest.raises(HipLibraryError, match="no CPU fallback"):
        HipBackend()
    from pypmc_amd.density.gauss import Gauss
    with pytest.raises(HipLibraryError):
        Gauss([0.], [[1.]]).evaluate(np.array([0.]))


def test_product_package_never_touches_the_oracle():
    """pypmc_amd/ must not import, load or shell out to anything under oracle/."""
    pkg = os.path.join(ROOT, "pypmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("oraclebackend", ""), os.path.join(dirpath, f)


def test_ctx_demo_builds_with_plain_gcc(tmp_path):
    """the handle layer needs nothing but a C compiler on the caller's side: examples/ctx_demo.c with gcc -std=c99,
    pmc_ctx.h / pmc_hip.h as its only non-standard headers; run without a GPU it reports the missing device and exits"""
    import subprocess
    from pypmc_amd import _lib
    exe = str(tmp_path / "ctx_demo")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "ctx_demo.c"), "-L", libdir, "-lpmc_hip", "-Wl,-rpath," + libdir, "-lm",
                    "-o", exe], check=True)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "10"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 3 and "pmc_init" in r.stderr


def test_host_convert_stats_is_bit_identical_to_numpy():
    """pmc_host_convert_stats (the library's host code) against split_stats + shift_is_far + centred_moments of
    pypmc_amd/mix_adapt/_stats.py: same operations in the same order, same bits -- zeros, NaN and inf included"""
    from pypmc_amd.mix_adapt._stats import split_stats, centred_moments, shift_is_far, convert_stats
    rs = np.random.RandomState(0)
    for trial in range(200):
        K, D = int(rs.randint(1, 40)), int(rs.randint(1, 25))
        ps = 1 + D + D * (D + 1) // 2
        flat = rs.normal(size=8 + K * ps + 2 * K)
        body = flat[8:8 + K * ps].reshape(K, ps)
        body[:, 0] = np.abs(body[:, 0]) * rs.choice([1, 1, 1e-3, 1e3, 0], size=K)
        body[:, 1:1 + D] *= rs.choice([1., 1e-2, 1e2])
        if trial % 7 == 0:
            flat[8 + rs.randint(K * ps)] = np.nan
        if trial % 11 == 0:
            flat[8 + rs.randint(K * ps)] = np.inf
        vs = flat[8 + K * ps:].reshape(K, 2)
        vs[:, 0] = np.abs(vs[:, 0]) + 0.1
        shift = rs.normal(size=(K, D))
        student = trial % 2 == 0
        sc, S0, M1, M2, V1, V2 = split_stats(flat, K, D)
        with np.errstate(all='ignore'):
            far = shift_is_far(S0, M1, M2)
            mean, cov = centred_moments(S0, M1, M2, shift, S0_cov=V1 if student else None)
            got = convert_stats(flat, K, D, shift, n_cov='vsum0' if student else None)
        for a, b in ((sc, got[0]), (S0, got[1]), (M1, got[2]), (mean, got[3]), (cov, got[4]), (V1, got[6]), (V2, got[7])):
            assert np.array_equal(a, b, equal_nan=True), trial
        assert far == got[5], trial


def test_split_plan_invariants():
    """round 6: the components of a sample block in pieces (include/pmc_hip.h, "split_components").  What the plan decides
    for a shape is host arithmetic: for a sweep of shapes -- every compiled dimension class, K = 1 ... 1000, a target mixture
    larger than the proposal, launches from one block to beyond split_max_rounds -- the pieces cover every component exactly
    once, the grid is whole blocks + blocks x pieces, the blocks in pieces fit the library's ticket counters, and what the
    launch writes into the pieces' region fits the bytes pmc_workspace_bytes reserved for it behind everything else."""
    from pypmc_amd import _lib
    lib = _lib.load()
    fn = lib.pmc_internal_split_plan
    fn.restype = C.c_int
    fn.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    out = (C.c_int64 * 10)()
    rs = np.random.RandomState(0)
    seen_on = seen_tail = 0
    shapes = [(N, K, K2, D) for D in (2, 8, 16, 20, 24, 30, 32, 40, 64)
              for N in (1, 255, 256, 257, 4096, 65536, 262144, 300000, 1_000_000, 1_250_000, 5_000_000, 10_000_000, 40_000_000)
              for K, K2 in ((1, 0), (3, 0), (4, 4), (16, 0), (32, 4), (33, 40), (64, 0), (128, 4), (1000, 7), (5, 300))]
    for N, K, K2, D in shapes:
        nblocks = -(-(-(-N // 64)) // 4)
        ws = int(lib.pmc_workspace_bytes(N, max(K, K2), D))
        for resp in (0, 1):
            assert fn(N, K, K2 if not resp else 0, D, resp, out) == 0, lib.pmc_last_error()
            on, b1, s1, c1, s2, c2, grid, used, reserved, offset = [int(v) for v in out]
            if not on:
                continue
            seen_on += 1
            bt = nblocks - b1
            seen_tail += b1 > 0
            assert 0 <= b1 < nblocks and 1 <= bt <= 4096, (N, K, D, resp, b1)
            units1 = -(-K // 16) if resp else K
            assert s1 >= 1 and c1 >= 1 and (s1 - 1) * c1 < units1 <= s1 * c1, "the pieces cover the components exactly"
            if not resp and K2:
                assert s2 >= 1 and (s2 - 1) * c2 < K2 <= s2 * c2
            else:
                assert s2 == 0
            assert s1 + s2 >= 2 and grid == b1 + bt * (s1 + s2) and grid < 2 ** 31
            assert used <= reserved, (N, K, K2, D, resp, used, reserved)
            assert offset % 256 == 0 and offset + used <= ws, "the pieces' region lies inside the workspace"
    assert seen_on > 200 and seen_tail > 20
    # switched off: no plan
    assert lib.pmc_configure(b"split_components", 0.0) == 0
    try:
        assert fn(4096, 128, 0, 40, 0, out) == 0 and out[0] == 0
    finally:
        d = C.c_double()
        assert lib.pmc_option_default(b"split_components", C.byref(d)) == 0
        assert lib.pmc_configure(b"split_components", d.value) == 0


def test_psi_and_ln_gamma_of_the_device_update_agree_with_scipy():
    """pmc_host_digamma / pmc_host_lgamma are the functions the device-resident VB update uses (pmc_vbstate.hip:
    recurrence + asymptotic series); scipy's are what the reference's M-step and bound call
    (variational.pyx:759-772, :1220-1275)."""
    from scipy.special import digamma, gammaln
    from pypmc_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(3)
    xs = np.concatenate([10 ** rng.uniform(-8, 8, 4000), np.linspace(0.01, 30, 2000),
                         [1.0, 2.0, 1.4616321449683623, 0.5, 1e-5, 1e7, 9.999999, 10.0]])
    d = np.array([lib.pmc_host_digamma(float(x)) for x in xs])
    g = np.array([lib.pmc_host_lgamma(float(x)) for x in xs])
    assert np.max(np.abs(d - digamma(xs)) / (1 + np.abs(digamma(xs)))) < 2e-15
    assert np.max(np.abs(g - gammaln(xs)) / (1 + np.abs(gammaln(xs)))) < 1e-14
    for bad in (0.0, -1.5, float("nan")):
        assert np.isnan(lib.pmc_host_digamma(bad)) and np.isnan(lib.pmc_host_lgamma(bad))
