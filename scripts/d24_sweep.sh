#!/bin/bash
# verdict r4 #7: the D = 24 cliff (k_stats_gemm<24> 0.66, k_resp<24> 0.67 of the fp64 peak at K = 64, N = 4e6).
# (1) tile shapes of k_stats_gemm<24>: 325 monomials = 21 column tiles = 3 x 7; the shape of rounds 3-4 (C6 x CGW4 = 24 tiles)
#     leaves 3 of 24 column tiles idle; (2) grouped responsibilities at D = 24 (off below K = 128 since round 3).  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export KB_ARGS="--K 64"
bash scripts/tune_unit.sh stats 24 estep:k_stats \
  "-DPMC_GEMM_C=6 -DPMC_GEMM_CGW=4 -DPMC_GEMM_SL=2 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=7 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=4 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=7 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=2 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=7 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=2 -DPMC_GEMM_NS=1 -DPMC_GEMM_WGS=2" \
  "-DPMC_GEMM_C=7 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=4 -DPMC_GEMM_NS=1" \
  "-DPMC_GEMM_C=3 -DPMC_GEMM_CGW=7 -DPMC_GEMM_SL=2 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=3 -DPMC_GEMM_CGW=7 -DPMC_GEMM_SL=1 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=4 -DPMC_GEMM_CGW=6 -DPMC_GEMM_SL=2 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=6 -DPMC_GEMM_CGW=4 -DPMC_GEMM_SL=4 -DPMC_GEMM_NS=2"
echo "--- responsibilities at D = 24: k_resp (option 0) against k_resp_groups (option 2), K = 32 / 64 / 128"
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import mk, vb_params
from pypmc_amd.backend import HipBackend, ComponentSet
be = HipBackend(0)
D, N = 24, 4_000_000
for K in (32, 64, 128):
    mu, cov, w = mk(K, D, 3)
    W, beta, nu, ln_pi, ln_lambda = vb_params(mu, cov, w, N)
    post = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, D, dtype=torch.float64, device='cuda', generator=g) * 1.2
    x += torch.tensor(mu, device='cuda')[torch.randint(0, K, (N,), device='cuda', generator=g)]
    stats = be.zeros(be.stats_len(K, D))
    pack = be.pack(post)
    for opt in (0, 2, 0, 2):
        be.configure("estep_grouped_responsibilities", opt)
        for _ in range(3):
            be.estep(x, post, 0, pack=pack, out=stats)
        torch.cuda.synchronize()
        be.kernel_timings(); be.kernel_timing(True)
        for _ in range(10):
            be.estep(x, post, 0, pack=pack, out=stats)
        torch.cuda.synchronize()
        be.kernel_timing(False)
        t = be.kernel_timings()
        print("D=24 K=%d grouped=%d: " % (K, opt) + "  ".join("%s %.4f" % (k, v["ms"] / v["calls"]) for k, v in t.items()), flush=True)
    be.configure("estep_grouped_responsibilities", 1)
    del x
PY
