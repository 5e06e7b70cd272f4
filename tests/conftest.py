import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libpmc_hip.so is a build artefact (git-ignored): build it if this checkout has none yet
    (hipcc cross-compiles for gfx950 without a GPU; objects are cached under pypmc_amd/csrc/build)."""
    from pypmc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from pypmc_amd import build
        build.build()
    yield
