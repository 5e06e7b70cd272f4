#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/profile_configs.sh'): BASELINE configurations 3 and 4 (and 5) as
# bench.py runs them, each under rocprofv3 --kernel-trace --stats and under the SQ-counter passes of
# scripts/kprof.py.  Everything lands in gpurun_out/cfgprof/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/cfgprof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in ${1:-cfg3 cfg4 cfg5}; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_stats" -o t -- \
        python "$R/bench.py" --configs-only $cfg > "$OUT/${cfg}_stdout.json" 2> "$OUT/${cfg}_stderr.log"
    cp $(find "$OUT/${cfg}_stats" -name "*kernel_stats.csv" | head -1) "$OUT/${cfg}_kernel_stats.csv"
    timeout 900 python "$R/scripts/kprof.py" --filter k_ --dir /tmp/kprof_$cfg --out "$OUT/${cfg}_sq_counters.json" -- \
        python "$R/bench.py" --configs-only $cfg > "$OUT/${cfg}_sq_counters.txt" 2>&1
    rm -rf "$OUT/${cfg}_stats"
done
ls -la "$OUT"
