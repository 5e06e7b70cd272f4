#!/bin/bash
# usage: scripts/tune_stats.sh D "FLAGS1" "FLAGS2" ...   (run on the GPU box through gpurun)
# Rebuilds the statistics unit of one dimension with extra -D flags and times it (kbench).
D=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
cd $(dirname $0)/..
for cfg in "$@"; do
  hipcc $FLAGS -DPMC_D=$D -DPMC_PADDED=0 $cfg -c pypmc_amd/csrc/pmc_stats.hip \
     -o pypmc_amd/csrc/build/pmc_stats_d${D}_p0.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|Spill: [1-9]" | sed 's/.*remark: *//' | tr '\n' ' '
  hipcc --offload-arch=gfx950 -shared -fPIC -o pypmc_amd/lib/libpmc_hip.so pypmc_amd/csrc/build/*.o
  echo "== $cfg"
  python scripts/kbench.py --N 4000000 --D $D ${KB_ARGS} 2>&1 | grep -A1 "vb_stats_only" | tail -1
done
