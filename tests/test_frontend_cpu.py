"""Host-side logic of the front-end on CPU: every case of tests/frontend_cases.py runs with the
oracle-backed checker injected as backend (no GPU, no HIP library calls)."""
import pytest

import frontend_cases
from oracle_backend import OracleBackend


@pytest.fixture(scope="module")
def be():
    return OracleBackend()


@pytest.mark.parametrize("case", frontend_cases.ALL_CASES, ids=lambda c: c.__name__)
def test_case(case, be):
    case(be)
