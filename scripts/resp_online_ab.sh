#!/bin/bash
# verdict r4 #6: what could an "online" k_resp_groups -- u' written against a reference known before the group, no parking in
# LDS, no second pass over the group -- gain at best?  The product against a TIMING-ONLY variant with exactly that
# instruction stream (wrong numbers; -DPMC_AB_ONLINE, built by
#   PMC_VARIANT=online PMC_VARIANT_UNITS=pmc_persample_d20_p0 PMC_EXTRA_FLAGS=-DPMC_AB_ONLINE python -m pypmc_amd.build ).
# Kernel times alternating on one box; SQ counters of both.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/pypmc_amd/lib/libpmc_hip_online.so
cd /tmp && export TMPDIR=/tmp
for round in 1 2 3; do
  python $R/scripts/estep_loop.py
  PMC_HIP_LIBRARY=$V python $R/scripts/estep_loop.py
done
echo "--- SQ counters (K = 32): product, then variant"
python $R/scripts/kprof.py --filter k_resp -- python $R/scripts/estep_loop.py --reps 3 --K 32 2>&1 | grep -A1 "k_resp_groups" | grep -v "^--"
PMC_HIP_LIBRARY=$V python $R/scripts/kprof.py --filter k_resp -- python $R/scripts/estep_loop.py --reps 3 --K 32 2>&1 | grep -A1 "k_resp_groups" | grep -v "^--"
