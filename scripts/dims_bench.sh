#!/bin/bash
# Kernel times over the compiled sample dimensions (4e6 samples; run on the GPU box):
#   bash scripts/dims_bench.sh [K] [dims...] > gpurun_out/dims.txt
# logpdf / resp / stats: the three kernels alone; estep: pmc_estep (ONE fused kernel where pmc_estep_is_fused)
cd "$(dirname "$0")/.."
K=${1:-32}
shift
DIMS=${@:-2 5 8 12 16 20 24 30 32 40 48 64}
for d in $DIMS; do
  python scripts/kbench.py --N 4000000 --D $d --K $K --reps 3 2>/dev/null | python -c "
import json, sys
r = json.load(sys.stdin)
D, K, N = $d, $K, 4000000
f_lp = N * (K * (D * D + 4 * D) + K * 40); f_st = N * K * (1 + 2 * D + D * (D + 1))
print('D=%2d K=%2d  logpdf %7.3f ms (%5.1f TF)  resp %7.3f ms (%5.1f TF)  stats %7.3f ms (%5.1f TF)  estep %7.3f ms (%5.1f TF)%s' % (
    D, K, r['logpdf']['ms'], f_lp / r['logpdf']['ms'] * 1e-9, r['vb_resp_only']['ms'], f_lp / r['vb_resp_only']['ms'] * 1e-9,
    r['vb_stats_only']['ms'], f_st / r['vb_stats_only']['ms'] * 1e-9, r['vb_estep']['ms'],
    (f_lp + f_st) / r['vb_estep']['ms'] * 1e-9, '  [fused]' if r['vb_estep'].get('fused') else ''))"
done
