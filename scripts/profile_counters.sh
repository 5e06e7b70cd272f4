#!/bin/bash
# SQ counters of the bench kernels (run on the GPU box; two passes of <= 6 counters, kernel-trace only):
#   gpurun -- 'bash scripts/profile_counters.sh'   then   python scripts/collect_counters.py r01
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/sq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -o t -- \
        python "$R/bench.py" --no-cpu-baseline --no-configs --no-traffic --steps 3 --warmup 1 --prewarm 0 > /dev/null 2> "$OUT/pass$i.log"
    echo "pass $i rc=$?"
done
