"""Development aid: the constructor E-step of a GaussianInference from the device-resident state against the host path, field by field
(bitwise), and the E-step repeated on the same object."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_kernels import mk
from pypmc_amd.density.mixture import create_gaussian_mixture
from pypmc_amd.mix_adapt.variational import GaussianInference
D, K, N = 20, 32, 600_000
mixture = create_gaussian_mixture(*mk(K, D, 71))
np.random.seed(72)
x = mixture.propose(N)
out = {}
for dev in (True, False):
    GaussianInference.device_update = dev
    vb = GaussianInference(x, initial_guess=mixture)
    out[dev] = vb
a, b = out[True], out[False]
for n in ("expectation_det_ln_lambda", "expectation_ln_pi", "N_comp", "x_mean_comp", "S", "nu", "beta", "m", "W", "log_det_W"):
    u, v = np.asarray(a._peek(n)), np.asarray(b._peek(n))
    print(n, np.array_equal(u, v), np.abs(u - v).max())
print(a._expectation_log_q_Z, b._expectation_log_q_Z)
print(20 * np.log(2 * np.pi), 20 * 1.83787706640934548356, np.log(2*np.pi) == 1.83787706640934548356)
# run to run: the same object, E-step repeated
for rep in range(3):
    a.E_step(); b.E_step()
    print(rep, a._expectation_log_q_Z, b._expectation_log_q_Z, np.array_equal(a._peek("N_comp"), b._peek("N_comp")))
