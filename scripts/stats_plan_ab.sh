#!/bin/bash
# What k_stats_gemm's plan (the common shift and the a-priori test, made by every workgroup before its first tile) costs per
# launch: the product against a TIMING-ONLY variant without it, at the full size and at one GPU's share of eight.
#   HERE:  PMC_VARIANT=noplan PMC_VARIANT_UNITS=pmc_stats_d20_p0 PMC_EXTRA_FLAGS=-DPMC_AB_NOPLAN python -m pypmc_amd.build
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/pypmc_amd/lib/libpmc_hip_noplan.so
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for N in 10000000 1250000 156250; do
    timeout 200 python $R/scripts/estep_loop.py --K 32,64 --N $N 2>&1 | grep -v amdgpu
    PMC_HIP_LIBRARY=$V timeout 200 python $R/scripts/estep_loop.py --K 32,64 --N $N 2>&1 | grep -v amdgpu
  done
done
