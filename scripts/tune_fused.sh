#!/bin/bash
# usage: scripts/tune_fused.sh "D list" "K list" "FLAGS1" "FLAGS2" ...   (run on the GPU box through gpurun)
# Rebuilds the dispatcher and the fused E-step unit of each dimension with extra -D flags (the launch geometry
# lives in the dispatcher, so both see the same macros) and prints kbench's vb_estep time.
DS=$1; KS=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
cd $(dirname $0)/..
for cfg in "$@"; do
  hipcc $FLAGS $cfg -c pypmc_amd/csrc/pmc_api.hip -o pypmc_amd/csrc/build/pmc_api.o 2>&1 | grep -E "error"
  for D in $DS; do
    hipcc $FLAGS -DPMC_D=$D -DPMC_PADDED=0 $cfg -c pypmc_amd/csrc/pmc_fused.hip -o pypmc_amd/csrc/build/pmc_fused_d${D}_p0.o 2>&1 | grep -E "error"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o pypmc_amd/lib/libpmc_hip.so pypmc_amd/csrc/build/*.o
  for D in $DS; do for K in $KS; do
    python scripts/kbench.py --N 4000000 --D $D --K $K --reps 7 2>/dev/null | python -c "
import json, sys
r = json.load(sys.stdin)
print('D=%d K=%-3d vb_estep %8.4f ms (median %8.4f)   [%s]' % (r['D'], r['K'], r['vb_estep']['ms'], r['vb_estep']['ms_median'], '''$cfg'''))"
  done; done
done
