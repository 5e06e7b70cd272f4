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


def test_very_big_dimension():
    from pypmc_amd.backend import HipBackend
    fuzz_gpu.sweep(seed=3, rounds=1, be=HipBackend(), verbose=False, dims=[520, 1024], kmax=3, nmax=200)
