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


@pytest.fixture(autouse=True)
def _poisoned_allocator(request):
    """GPU tests start with torch's caching allocator holding blocks full of NaN bit patterns, small pool and
    large pool alike: a kernel that reads workspace, partials or an output it was meant to write first then
    fails every time instead of once in twenty full-suite runs (fresh memory on a new box is all zeros)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        nan = float("nan")
        big = torch.full((1 << 26,), nan, dtype=torch.float64, device="cuda:0")             # 512 MiB, large pool
        mid = [torch.full((1 << 18,), nan, dtype=torch.float64, device="cuda:0") for _ in range(16)]   # 2 MiB each
        small = [torch.full((1 << s,), nan, dtype=torch.float64, device="cuda:0") for s in range(6, 17) for _ in range(8)]
        torch.cuda.synchronize()
        del big, mid, small
        torch.manual_seed(20260929)      # tests that draw on the device without a generator stay reproducible
    yield
