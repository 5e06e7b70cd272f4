"""The same front-end cases on the product path: HipBackend -> ctypes -> libpmc_hip.so -> gfx950
kernels.  Parity targets: golden vectors generated from the reference (1e-10 relative on
log-weights / responsibilities, bit-exact component indices and counts)."""
import pytest

import frontend_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from pypmc_amd.backend import HipBackend
    return HipBackend()


@pytest.mark.parametrize("case", frontend_cases.ALL_CASES, ids=lambda c: c.__name__)
def test_case(case, be):
    case(be)


def test_default_backend_is_hip():
    from pypmc_amd.backend import get_backend, HipBackend
    assert isinstance(get_backend(), HipBackend)
