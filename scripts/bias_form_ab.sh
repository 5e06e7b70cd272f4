#!/bin/bash
# what could the "bias" form of the Mahalanobis products -- |R x' - b|^2 with b = R (mu - c) in the pack and
# x' = x - c formed once per sample: no subtraction per pair, no d[] registers -- gain at D = 20?  The product against a
# TIMING-ONLY variant with that instruction stream (wrong numbers; -DPMC_AB_BIAS, built by
#   PMC_VARIANT=bias PMC_VARIANT_UNITS=pmc_persample_d20_p0 PMC_EXTRA_FLAGS=-DPMC_AB_BIAS python -m pypmc_amd.build ).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/pypmc_amd/lib/libpmc_hip_bias.so
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import json,sys
a=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms_per_step %.3f' % a['ms_per_step'], ' '.join('%s %.4f' % kv for kv in a['kernel_ms'].items()))"; }
for round in 1 2 3; do
  python $R/bench.py --no-cpu-baseline --no-configs --no-traffic 2>/dev/null | line product
  PMC_HIP_LIBRARY=$V python $R/bench.py --no-cpu-baseline --no-configs --no-traffic 2>/dev/null | line "bias   "
done
echo "--- SQ counters: product, then variant"
python $R/scripts/kprof.py --filter k_ -- python $R/bench.py --no-cpu-baseline --no-configs --no-traffic --steps 3 --warmup 1 --prewarm 0 2>&1 | grep -A1 "k_logpdf\|k_resp_groups" | grep -v "^--"
PMC_HIP_LIBRARY=$V python $R/scripts/kprof.py --filter k_ -- python $R/bench.py --no-cpu-baseline --no-configs --no-traffic --steps 3 --warmup 1 --prewarm 0 2>&1 | grep -A1 "k_logpdf\|k_resp_groups" | grep -v "^--"
