#!/bin/bash
# usage (GPU box): scripts/ab_build_flags.sh "D [--student];D ..." "FLAGS_A" "FLAGS_B" ...
# Rebuilds the WHOLE library with each flag set (pack layout and kernels must agree: e.g. -DPMC_DPP_FROM=24), then once
# more without extra flags (the baseline, which is also what is left behind), and prints kbench's log-pdf /
# responsibility / E-step times for every "D [--student]" case.
cd $(dirname $0)/..
CASES=$1; shift
for cfg in "$@" ""; do
  PMC_EXTRA_FLAGS="$cfg" python -m pypmc_amd.build -j 48 --force > /dev/null 2>&1 || { echo "build failed: $cfg"; continue; }
  IFS=';' read -ra CS <<< "$CASES"
  for c in "${CS[@]}"; do
    python scripts/kbench.py --N ${KB_N:-4000000} --D $c 2>/dev/null | python -c "
import json, sys
r = json.load(sys.stdin)
print('D=%d K=%d st=%d  logpdf %7.4f (%7.4f)  logpdf+is %7.4f  resp %7.4f (%7.4f)  estep %7.4f   [%s]' % (r['D'], r['K'], r['student'], r['logpdf']['ms'], r['logpdf']['ms_median'], r['logpdf+is']['ms'], r['vb_resp_only']['ms'], r['vb_resp_only']['ms_median'], r['vb_estep']['ms'], '''$cfg'''))"
  done
done
