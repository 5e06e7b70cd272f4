"""Same inputs, same bits: every reduction of the hot path runs in a fixed order (per-tile partials summed in
index order by the finishing kernels), so repeated launches must agree bitwise -- a missing barrier or an
atomics-ordered sum shows up as a differing launch (scripts/determinism_stress.py is the long form)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.mark.gpu
def test_repeated_launches_agree_bitwise():
    import determinism_stress
    assert determinism_stress.sweep(25, verbose=False) == 0
