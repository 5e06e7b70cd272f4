#!/bin/bash
# Timing-only A/B of k_vb_mstep's three stages (Cholesky, inverse of the factor, Gram matrix): which one is the 38 us?
# run HERE:  bash scripts/vb_mstep_ab.sh build     on the GPU box: bash scripts/vb_mstep_ab.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for v in NO_CHOL NO_INV NO_GRAM; do
    PMC_VARIANT=vb_$v PMC_VARIANT_UNITS=pmc_vbstate PMC_EXTRA_FLAGS="-DVB_AB_$v" python -m pypmc_amd.build > /dev/null
  done
  exit 0
fi
cd /tmp; export TMPDIR=/tmp
for v in "" vb_NO_CHOL vb_NO_INV vb_NO_GRAM; do
  lib=/root/repo/pypmc_amd/lib/libpmc_hip${v:+_$v}.so
  rm -rf /tmp/ab_$v; PMC_HIP_LIBRARY=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python /root/repo/scripts/vb_mstep_loop.py > /dev/null 2>&1 || true
  python3 - "$v" <<PY
import csv, glob, sys
for f in glob.glob("/tmp/ab_%s/*/*kernel_stats.csv" % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if "k_vb_mstep" in r["Name"]:
            print("variant %-12s %s calls, avg %.1f us" % (sys.argv[1] or "product", r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
