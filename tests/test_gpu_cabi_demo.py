"""The C ABI from plain C++/HIP (examples/cabi_demo.cpp): no Python, no PyTorch on the caller's side.
Compiled with hipcc on the GPU box, its printed numbers are compared with the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_demo_matches_oracle(tmp_path):
    from oracle import oracle as orc
    import pypmc_amd.build as build
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = build.build()
    exe = str(tmp_path / "cabi_demo")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "cabi_demo.cpp"), "-L", os.path.dirname(lib), "-lpmc_hip",
                    "-Wl,-rpath," + os.path.dirname(lib), "-o", exe], check=True)
    N = 1237
    res = subprocess.run([exe, str(N)], check=True, stdout=subprocess.PIPE, text=True).stdout.splitlines()
    assert res[0].startswith("abi 2 arch gfx950")
    D, K = 4, 3
    mu = np.array([[0, 0, 0, 0], [2, -1, 0.5, 1], [-3, 2, 1, -1]], dtype=float)
    var = np.array([[1, 2, 0.5, 1], [0.3, 0.7, 1.1, 2.0], [1.5, 0.4, 0.9, 1.2]])
    w = np.array([0.5, 0.3, 0.2])
    n = np.arange(N)[:, None]
    x = np.sin(0.37 * n + 1.3 * np.arange(D)[None, :]) * 3.0
    inv = np.array([np.diag(1. / v) for v in var])
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.log(var).sum(axis=1)
    lq = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)[0]
    lt = orc.mixture_multi_evaluate(0, x, w, mu, np.tile(np.eye(D), (K, 1, 1)), np.full(K, -0.5 * D * np.log(2 * np.pi)))[0]
    wts = orc.is_weights(lt, lq)
    got_lq = np.array([float(line.split()[2]) for line in res if line.startswith("logq")])
    np.testing.assert_allclose(got_lq, lq[:5], rtol=1e-12)
    sums = np.array([float(v) for v in [line for line in res if line.startswith("sums")][0].split()[1:]])
    np.testing.assert_allclose(sums, [wts.sum(), (wts * np.log(wts)).sum(), (wts ** 2).sum()], rtol=1e-11)
    rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
    got_nk = np.array([float(line.split()[2]) for line in res if line.startswith("N_k")])
    np.testing.assert_allclose(got_nk, (wts[:, None] * rho).sum(axis=0), rtol=1e-11)
    # the E-step went through the fused kernel, and the library's own timing reported it
    assert "fused 1" in res
    timing = [line for line in res if line.startswith("timing ")]
    assert any(line.startswith("timing k_logpdf: 2 launches") for line in timing)       # plain + keeping
    assert any(line.startswith("timing k_estep_fused: 1 launches") for line in timing)
    # the evaluate-once iteration (pmc_importance_weights_keep + pmc_estep_from_tiles) gave the same statistics
    worst = float([line for line in res if line.startswith("from_tiles")][0].split()[-1])
    assert worst < 1e-11
    # the library's own RCCL communicator (one rank): the sum over one rank leaves the buffer as it was
    assert "comm rank 0 of 1 allreduce identical" in res


def test_ctx_demo_plain_c_matches_oracle(tmp_path):
    """examples/ctx_demo.c: the handle layer (include/pmc_ctx.h) from C99 compiled with gcc -- no HIP headers, no device
    pointers on the caller's side -- against the oracle's loops"""
    from oracle import oracle as orc
    import pypmc_amd.build as build
    lib = build.build()
    exe = str(tmp_path / "ctx_demo")
    subprocess.run(["gcc", "-O2", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "ctx_demo.c"),
                    "-L", os.path.dirname(lib), "-lpmc_hip", "-Wl,-rpath," + os.path.dirname(lib), "-lm", "-o", exe], check=True)
    N = 2311
    res = subprocess.run([exe, str(N)], check=True, stdout=subprocess.PIPE, text=True).stdout.splitlines()
    assert res[-1] == "done"
    D, K = 4, 3
    mu = np.array([[0, 0, 0, 0], [2, -1, 0.5, 1], [-3, 2, 1, -1]], dtype=float)
    var = np.array([[1, 2, 0.5, 1], [0.3, 0.7, 1.1, 2.0], [1.5, 0.4, 0.9, 1.2]])
    w = np.array([0.5, 0.3, 0.2])
    n = np.arange(N)[:, None]
    x = np.sin(0.37 * n + 1.3 * np.arange(D)[None, :]) * 3.0
    inv = np.array([np.diag(1. / v) for v in var])
    ln = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.log(var).sum(axis=1)
    lq = orc.mixture_multi_evaluate(0, x, w, mu, inv, ln)[0]
    lt = orc.mixture_multi_evaluate(0, x, w, mu, np.tile(np.eye(D), (K, 1, 1)), np.full(K, -0.5 * D * np.log(2 * np.pi)))[0]
    wts = orc.is_weights(lt, lq)
    col = lambda tag, i=2: np.array([float(line.split()[i]) for line in res if line.startswith(tag + " ")])
    np.testing.assert_allclose(col("logq"), lq[:5], rtol=1e-12)
    sums = np.array([float(v) for v in [line for line in res if line.startswith("sums")][0].split()[1:]])
    np.testing.assert_allclose(sums, [wts.sum(), (wts * np.log(wts)).sum(), (wts ** 2).sum()], rtol=1e-11)
    np.testing.assert_allclose(col("perplexity", 1), orc.perp(wts), rtol=1e-11)
    # gaussian_pmc's sums (pmc.pyx:188-222)
    rho = orc.rho_rb(0, x, w, mu, inv, ln, None, None, list(range(K)))
    alpha, nmu, ncov = orc.pmc_reductions(x, rho, None, wts, list(range(K)))
    np.testing.assert_allclose(col("alpha"), alpha / wts.sum(), rtol=1e-10)      # the oracle returns sum w rho
    np.testing.assert_allclose(col("mu0"), nmu[:, 0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(col("sigma00"), ncov[:, 0, 0], rtol=1e-10, atol=1e-12)
    ll = [line for line in res if line.startswith("loglik")][0].split()
    np.testing.assert_allclose([float(ll[1]), float(ll[3])], [(wts * lq).sum(), wts.sum()], rtol=1e-11)
    # GaussianInference.E_step (variational.pyx:699-932, :1003-1013)
    nu, beta = D + 2.0 + np.arange(K), 1.0 + np.arange(K)
    o = orc.vb_estep(x, None, mu, inv / nu[:, None, None], beta, nu, np.log(w), 0.25 * np.arange(K))
    np.testing.assert_allclose(col("N_comp"), o["N_comp"], rtol=1e-11)
    np.testing.assert_allclose(col("S00"), o["S"][:, 0, 0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(col("elogqz", 1), o["expectation_log_q_Z"], rtol=1e-10)
