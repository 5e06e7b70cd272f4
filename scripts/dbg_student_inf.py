import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from test_gpu_kernels import mk, draw, student_set
from pypmc_amd.backend import HipBackend
from oracle import oracle as orc
be = HipBackend()
for D, K in [(1, 2), (2, 2), (4, 3), (20, 4), (40, 5)]:
    rs = np.random.RandomState(50 + D)
    mu, cov, w = mk(K, D, 60 + D)
    dof = rs.uniform(2., 9., K)
    cs, inv, ln, pf, idf = student_set(mu, cov, w, dof)
    x, _ = draw(mu, cov, w, 20, 3)
    far = np.arange(0, 20, 7)
    x[far, rs.randint(0, D, len(far))] = 1e160 * rs.choice([-1., 1.], len(far))
    with np.errstate(all="ignore"):
        ref, ref_ind = orc.mixture_multi_evaluate(1, x, w, mu, inv, ln, pf, idf)
    res = be.logpdf(x, cs, want_individual=True)
    ind, out = be.tohost(res["individual"]), be.tohost(res["out"])
    print(D, K, "far rows ind:", ind[far].tolist(), "out:", out[far].tolist(), "ref:", ref[far].tolist(), ref_ind[far].tolist())
