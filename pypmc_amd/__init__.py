"""pypmc_amd -- MI355X-native adaptive-importance-sampling core behind pypmc's public API.

The accelerated path: mixture log-pdf -> importance weights / perplexity ->
VB E-step / PMC responsibilities + sufficient statistics, as hand-written gfx950 kernels in
``pypmc_amd/lib/libpmc_hip.so`` (C ABI: include/pmc_hip.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import backend, parallel          # noqa: F401
from . import tools, density, sampler, mix_adapt   # noqa: F401

tools.util.log_to_stdout()                 # pypmc/__init__.py:13
