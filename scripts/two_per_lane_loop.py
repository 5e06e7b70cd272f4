"""the headline's importance-weight pass (D = 20, K = 32 + 4, N = 1e7) in a loop: kernel time from the library's events.
PMC_AB_TWO_PER_LANE=1 with a -DPMC_TWO_PER_LANE build takes the two-samples-per-lane kernel (scripts/two_per_lane_ab.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import mk, gauss_params
from pypmc_amd.backend import HipBackend, ComponentSet
be = HipBackend()
for K, KT, D, N in ((32, 4, 20, 10_000_000), (64, 4, 20, 5_000_000), (32, 4, 16, 10_000_000), (32, 4, 24, 8_000_000)):
    mu, cov, w = mk(K, D, 1); tmu, tcov, tw = mk(KT, D, 11)
    inv, ln = gauss_params(mu, cov); tinv, tln = gauss_params(tmu, tcov)
    prop, tgt = ComponentSet(0, mu, inv, c0=ln, weight=w), ComponentSet(0, tmu, tinv, c0=tln, weight=tw)
    rs = np.random.RandomState(2); comp = rs.choice(K, N, p=w)
    x = be.asdevice((mu[comp] + np.einsum('nij,nj->ni', np.linalg.cholesky(cov)[comp], rs.normal(size=(N, D)))))
    ref = None
    for _ in range(30): r = be.importance_weights(x, prop, tgt)
    torch.cuda.synchronize(); be.kernel_timing(True); be.kernel_timings()
    for _ in range(30): r = be.importance_weights(x, prop, tgt)
    t = be.kernel_timings(); be.kernel_timing(False)
    print("D=%d K=%d+%d N=%d  k_logpdf %.4f ms   sum w = %.17g" % (D, K, KT, N, t["k_logpdf"]["ms"] / t["k_logpdf"]["calls"],
                                                                 float(r["scalars"][0])), flush=True)
    del x
