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
