#!/bin/bash
# Kernel times over the compiled sample dimensions (K = 32, 4e6 samples; run on the GPU box):
#   bash scripts/dims_bench.sh > gpurun_out/dims.txt
cd "$(dirname "$0")/.."
for d in 2 5 8 12 16 20 24 30 32 40 48 64; do
  python scripts/kbench.py --N 4000000 --D $d --K 32 --reps 3 2>/dev/null | python -c "
import json, sys
r = json.load(sys.stdin)
D, K, N = $d, 32, 4000000
f_lp = N * (K * (D * D + 4 * D) + K * 40); f_st = N * K * (1 + 2 * D + D * (D + 1))
print('D=%2d  logpdf %7.3f ms (%5.1f TF)  resp %7.3f ms (%5.1f TF)  stats %7.3f ms (%5.1f TF)' % (
    D, r['logpdf']['ms'], f_lp / r['logpdf']['ms'] * 1e-9, r['vb_resp_only']['ms'], f_lp / r['vb_resp_only']['ms'] * 1e-9,
    r['vb_stats_only']['ms'], f_st / r['vb_stats_only']['ms'] * 1e-9))"
done
