#!/bin/bash
# k_propose<40>: wavefronts per SIMD the compiler is bound to (registers: 512 / n), on the GPU box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
for w in 2 1 3; do
  hipcc $FLAGS -DPMC_D=40 -DPMC_PADDED=0 -DPMC_PROPOSE_WAVES=$w -c pypmc_amd/csrc/pmc_propose.hip -o pypmc_amd/csrc/build/pmc_propose_d40_p0.o 2>&1 | grep error
  hipcc --offload-arch=gfx950 -shared -fPIC -o pypmc_amd/lib/libpmc_hip.so pypmc_amd/csrc/build/*.o -ldl
  echo "== waves per SIMD >= $w: $(python scripts/kres.py pypmc_amd/csrc/build/pmc_propose_d40_p0.o | cut -c60-140)"
  python scripts/propose_time.py 2>&1 | grep "D=40 K=128"
done
