#!/bin/bash
# usage: scripts/tune_unit.sh UNIT D KEY "FLAGS1" "FLAGS2" ...   (run on the GPU box through gpurun)
# Rebuilds one kernel unit (persample | stats | fused) of one dimension with extra -D flags and prints
# the kbench entry KEY (logpdf | vb_resp_only | vb_stats_only | vb_estep).  KB_ARGS adds kbench flags.
UNIT=$1; D=$2; KEY=$3; shift 3
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
cd $(dirname $0)/..
for cfg in "$@"; do
  hipcc $FLAGS -DPMC_D=$D -DPMC_PADDED=0 $cfg -c pypmc_amd/csrc/pmc_${UNIT}.hip \
     -o pypmc_amd/csrc/build/pmc_${UNIT}_d${D}_p0.o 2>&1 | grep -E "error" 
  hipcc --offload-arch=gfx950 -shared -fPIC -o pypmc_amd/lib/libpmc_hip.so pypmc_amd/csrc/build/*.o
  python scripts/kbench.py --N 4000000 --D $D ${KB_ARGS} 2>/dev/null | python -c "
import json, sys
r = json.load(sys.stdin)
print('D=%d K=%d %-14s %8.4f ms (median %8.4f)   [%s]' % (r['D'], r['K'], '$KEY', r['$KEY']['ms'], r['$KEY']['ms_median'], '''$cfg'''))"
done
