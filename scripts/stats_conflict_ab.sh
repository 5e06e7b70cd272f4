#!/bin/bash
# verdict r5 #3: k_stats_gemm's LDS bank-conflict fraction (0.21 at D = 20 and D = 40) -- on the critical path or not?  The product against
# a TIMING-ONLY variant whose B-operand reads cannot conflict (-DPMC_AB_NOCONFLICT, wrong numbers), built by
#   PMC_VARIANT=nocf PMC_VARIANT_UNITS=pmc_stats_d20_p0,pmc_stats_d40_p0 PMC_EXTRA_FLAGS=-DPMC_AB_NOCONFLICT python -m pypmc_amd.build
# Kernel times alternating on one box; the conflict counters of both.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/pypmc_amd/lib/libpmc_hip_nocf.so
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  timeout 200 python $R/scripts/estep_loop.py --K 32,64 2>&1 | grep -v amdgpu
  PMC_HIP_LIBRARY=$V timeout 200 python $R/scripts/estep_loop.py --K 32,64 2>&1 | grep -v amdgpu
  timeout 200 python $R/scripts/estep_loop.py --K 128 --D 40 --N 2000000 2>&1 | grep -v amdgpu
  PMC_HIP_LIBRARY=$V timeout 200 python $R/scripts/estep_loop.py --K 128 --D 40 --N 2000000 2>&1 | grep -v amdgpu
done
echo "--- SQ_LDS_BANK_CONFLICT / SQ_LDS_ACTIVE... (K = 32, D = 20): product, then variant"
timeout 200 python $R/scripts/kprof.py --filter k_stats -- python $R/scripts/estep_loop.py --reps 3 --K 32 2>&1 | grep -i -A2 "k_stats_gemm" | head -8
PMC_HIP_LIBRARY=$V timeout 200 python $R/scripts/kprof.py --filter k_stats -- python $R/scripts/estep_loop.py --reps 3 --K 32 2>&1 | grep -i -A2 "k_stats_gemm" | head -8
