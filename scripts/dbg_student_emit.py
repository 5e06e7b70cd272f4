import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import mk
from pypmc_amd.backend import get_backend
from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture, component_set
be = get_backend()
D5, K5, KT5 = 40, 128, 4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rs = np.random.RandomState(5)
tmu, tcov, tw = mk(KT5, D5, 11)
tmu /= 3.0
target = create_gaussian_mixture(tmu, tcov, tw)
which = np.arange(K5) % KT5
means, covs = tmu[which] + rs.normal(0, 0.15, (K5, D5)), 1.5 * tcov[which]
for name, prop in (("student", create_t_mixture(means, covs, np.full(K5, 8.))), ("gauss", create_gaussian_mixture(means, covs))):
    np.random.seed(1)
    x = prop.propose(N, device=True)
    ps, ts = component_set(prop.components, prop.weights), component_set(target.components, target.weights)
    for emit in (False, True):
        r = be.importance_weights(x, ps, ts, emit=emit)
        torch.cuda.synchronize()
        rep = be.maha_gemm_report(N, K5, D5)
        t0 = time.perf_counter()
        for _ in range(5):
            be.importance_weights(x, ps, ts, emit=emit)
        torch.cuda.synchronize()
        print(name, "emit" if emit else "plain", "N", N, "report", None if rep is None else (rep["refused"], rep["workgroups"], rep["norms"]),
              "ms per call %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
