#!/bin/bash
# verdict r3 "next" 2: what do the 7 GB of u that the E-step writes and reads back at D = 20 cost?  The product library
# against a variant whose k_resp_groups does not store u and whose k_stats_gemm reads u (and the factors) of ONE tile, i.e.
# out of L2 -- wrong numbers, timing only (built here: PMC_VARIANT=nou ... python -m pypmc_amd.build).  Kernel times
# alternating, SQ clocks (scripts/kprof.py), socket power (scripts/power_probe.py).  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/pypmc_amd/lib/libpmc_hip_nou.so
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  python $R/scripts/estep_loop.py
  PMC_HIP_LIBRARY=$V python $R/scripts/estep_loop.py
done
echo "--- SQ counters (K = 32): product, then variant"
python $R/scripts/kprof.py --filter k_ -- python $R/scripts/estep_loop.py --reps 3 --K 32 2>&1 | grep -A1 "k_resp_groups\|k_stats_gemm" | grep -v "^--"
PMC_HIP_LIBRARY=$V python $R/scripts/kprof.py --filter k_ -- python $R/scripts/estep_loop.py --reps 3 --K 32 2>&1 | grep -A1 "k_resp_groups\|k_stats_gemm" | grep -v "^--"
echo "--- power (E-step loop): product, then variant"
python $R/scripts/power_probe.py 4 2>&1 | grep -i "estep\|cap" | head -4
PMC_HIP_LIBRARY=$V python $R/scripts/power_probe.py 4 2>&1 | grep -i "estep\|cap" | head -4
