"""The reference-side binding for real: examples/cython_binding/pmc_hip_binding.pyx -- Cython
over include/pmc_ctx.h, typed memoryviews as in the reference's own .pyx files -- compiled with cythonize + gcc and run
against the golden vectors generated from the reference."""
import importlib.util
import os
import subprocess
import sys
import sysconfig

import numpy as np
import pytest

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_binding(tmp):
    """cython -> C -> extension module, the commands a setup.py of the reference would issue"""
    from pypmc_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    src = os.path.join(ROOT, "examples", "cython_binding", "pmc_hip_binding.pyx")
    c_file = os.path.join(tmp, "pmc_hip_binding.c")
    subprocess.run([sys.executable, "-m", "cython", "-3", src, "-o", c_file], check=True)
    so = os.path.join(tmp, "pmc_hip_binding" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-I", sysconfig.get_paths()["include"], "-I", np.get_include(),
                    "-I", os.path.join(ROOT, "include"), c_file, "-L", libdir, "-lpmc_hip", "-Wl,-rpath," + libdir,
                    "-o", so], check=True)
    return so


def test_binding_compiles_without_hip_headers(tmp_path):
    so = build_binding(str(tmp_path))
    assert os.path.getsize(so) > 10000


@pytest.mark.gpu
def test_binding_reproduces_the_reference(tmp_path):
    so = build_binding(str(tmp_path))
    spec = importlib.util.spec_from_file_location("pmc_hip_binding", so)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    c64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    try:
        # GaussianInference.E_step on the reference's own data and parameters, both stages
        g = load_golden("vb_d20k8")
        data = b.DeviceSamples(c64(g["data"]))
        for stage in ("e0_", "u1_"):
            Nk, xbar, S, elq = b.E_step(data, c64(g[stage + "m"]), c64(g[stage + "W"]), c64(g[stage + "nu"]), c64(g[stage + "beta"]),
                                        c64(g[stage + "expectation_ln_pi"]), c64(g[stage + "expectation_det_ln_lambda"]))
            np.testing.assert_allclose(Nk, g[stage + "N_comp"], rtol=1e-10)
            np.testing.assert_allclose(xbar, g[stage + "x_mean_comp"], rtol=1e-10, atol=1e-13)
            np.testing.assert_allclose(S, g[stage + "S"], rtol=1e-10, atol=1e-12)
        # MixtureDensity.multi_evaluate
        g = load_golden("logpdf_gauss_d20k16")
        x = b.DeviceSamples(c64(g["x"]))
        out, ind = b.multi_evaluate(x, c64(g["weights"]), c64(g["mu"]), c64(g["inv_sigma"]), c64(g["log_norm"]))
        assert np.max(np.abs(out - g["out"]) / np.abs(g["out"])) < 1e-10
        assert np.max(np.abs(ind - g["individual"]) / np.abs(g["individual"])) < 1e-10
        # ... and its subset mode: the listed columns are written, the others keep the caller's values
        sub = np.full_like(g["individual"], -7.25)
        which = np.array([1, 5, 6, 15], dtype=np.int32)
        assert b.multi_evaluate_components(x, c64(g["weights"]), c64(g["mu"]), c64(g["inv_sigma"]), c64(g["log_norm"]),
                                           sub, which) is None
        assert np.max(np.abs(sub[:, which] - g["individual"][:, which]) / np.abs(g["individual"][:, which])) < 1e-10
        rest = np.setdiff1d(np.arange(sub.shape[1]), which)
        assert (sub[:, rest] == -7.25).all()
        # gaussian_pmc
        g = load_golden("pmc_gauss_d5k4")
        x = b.DeviceSamples(c64(g["samples"]))
        alpha, mu, sigma = b.gaussian_pmc_sums(x, c64(g["in_weights"]), c64(g["in_mu"]), c64(g["in_inv_sigma"]),
                                               c64(g["in_log_norm"]), c64(g["weights"]))
        np.testing.assert_allclose(alpha / alpha.sum(), g["rb_w_weights"], rtol=1e-10)
        np.testing.assert_allclose(mu, g["rb_w_mu"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(sigma, g["rb_w_sigma"], rtol=1e-10, atol=1e-12)
        del data, x
    finally:
        b.shutdown()
