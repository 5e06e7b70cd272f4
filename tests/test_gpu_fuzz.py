"""The randomised all-dimensions sweep (tests/fuzz_gpu.py) with fixed seeds: every sample dimension
1..64 -- exact and zero-padded kernel units -- random K <= 40, ragged N, every entry point of the path
against the oracle at the contract tolerance (1e-10 relative on log-pdf, importance weights, r, rho)."""
import pytest

import fuzz_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_sweep_all_dimensions(seed):
    from pypmc_amd.backend import HipBackend
    worst = fuzz_gpu.sweep(seed=seed, rounds=1, be=HipBackend(), verbose=False)
    for name in ("logpdf", "individual", "weights", "student logpdf", "vb r", "pmc rho"):
        assert worst[name] < 1e-10, (name, worst[name])


# sample dimensions beyond the compiled units: the run-time-dimension unit (csrc/pmc_big.hip) -- dimensions that are
# and are not multiples of 4 / 16, on both sides of its LDS staging limits (128, 256, 512)
BIG_DIMS = [65, 66, 79, 96, 100, 127, 128, 130, 200, 257]


@pytest.mark.parametrize("seed", [0, 1])
def test_sweep_big_dimensions(seed):
    from pypmc_amd.backend import HipBackend
    worst = fuzz_gpu.sweep(seed=seed, rounds=1, be=HipBackend(), verbose=False, dims=BIG_DIMS, kmax=9, nmax=700)
    for name in ("logpdf", "individual", "weights", "student logpdf", "vb r", "pmc rho"):
        assert worst[name] < 1e-10, (name, worst[name])


# the large-N forms of pmc_estep forced on at 16 384 ... 20 300 samples: exact and padded units from D = 8 (where the
# common-shift statistics start), K over one, two and three groups of 32 with ragged last groups
FAST_DIMS = [8, 9, 12, 16, 20, 23, 24, 30, 32, 37, 40, 48, 64]


@pytest.mark.parametrize("seed", [0, 1])
def test_sweep_fast_paths(seed):
    from pypmc_amd.backend import HipBackend
    worst = fuzz_gpu.sweep(seed=10 + seed, rounds=1, be=HipBackend(), verbose=False, dims=FAST_DIMS, kmax=70, fast_paths=True)
    for name in ("logpdf", "weights", "vb r", "pmc rho", "vb one-kernel stats", "pmc one-kernel stats"):
        assert worst[name] < 1e-9, (name, worst[name])


def test_very_big_dimension():
    from pypmc_amd.backend import HipBackend
    fuzz_gpu.sweep(seed=3, rounds=1, be=HipBackend(), verbose=False, dims=[520, 1024], kmax=3, nmax=200)


def test_big_dimension_scratch_is_bounded_and_chunked():
    """D > 64: the Mahalanobis forms the caller does not keep live in a scratch of bounded size, the samples go in
    chunks of a multiple of 256 -- same numbers, bit for bit, as one chunk (advice r2: the scratch was 8 N (K + K_t) bytes)"""
    import numpy as np
    from pypmc_amd.backend import HipBackend
    from test_gpu_kernels import mk, draw, gauss_set
    be = HipBackend()
    D, K, N = 72, 5, 5003
    mu, cov, w = mk(K, D, 4)
    x, _ = draw(mu, cov, w, N, 5)
    prop = gauss_set(mu, cov, w)[0]
    target = gauss_set(*mk(2, D, 6))[0]
    sw = np.random.RandomState(1).uniform(0.5, 1.5, N)

    def run():
        a = be.importance_weights(x, prop, target, sample_w=sw, want_out=True, want_log_target=True)
        b = be.importance_weights(x, prop, target, keep=True)
        c = be.logpdf(x, prop, want_individual=True, want_scalars=True, log_target=np.zeros(N))
        return [be.tohost(t).copy() for t in (a["weights"], a["out"], a["log_target"], a["scalars"], b["weights"],
                                              b["tiles"].data, c["out"], c["individual"], c["scalars"])]
    whole = run()
    be.configure("big_dim_scratch_bytes", 64 * 1024)            # 56 bytes per sample -> chunks of 1024 samples
    try:
        chunked = run()
    finally:
        be.configure("big_dim_scratch_bytes", 256 * 1024 * 1024)
    for got, ref in zip(chunked, whole):
        np.testing.assert_array_equal(got, ref)
