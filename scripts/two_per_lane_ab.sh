#!/bin/bash
# verdict r5 #3: two samples per lane in the D = 20 vector kernels -- every scalar coefficient feeds two multiply-adds (half the
# s_load / s_waitcnt per pair), ~190 registers, two wavefronts per SIMD.  A REAL variant (k_logpdf2: same bits), built by
#   PMC_VARIANT=two PMC_VARIANT_UNITS=pmc_persample_d16_p0,pmc_persample_d20_p0,pmc_persample_d24_p0 PMC_EXTRA_FLAGS=-DPMC_TWO_PER_LANE python -m pypmc_amd.build
# and switched on by PMC_AB_TWO_PER_LANE=1.  Kernel times alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/pypmc_amd/lib/libpmc_hip_two.so
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  echo "--- product"; timeout 300 python $R/scripts/two_per_lane_loop.py 2>&1 | grep -v amdgpu
  echo "--- two samples per lane"; PMC_HIP_LIBRARY=$V PMC_AB_TWO_PER_LANE=1 timeout 300 python $R/scripts/two_per_lane_loop.py 2>&1 | grep -v amdgpu
done
