#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/profile_bench.sh'): the driver's bench.py command under
# rocprofv3 (kernel trace + stats), the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate
# runs, kernel-trace only -- see MI355X_MICROARCH.md), and one unprofiled run with the CPU baseline.
# Everything lands in gpurun_out/prof/; scripts/collect_profiles.py turns it into profiles/rNN_*.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$R/bench.py" > "$OUT/bench_stdout.json" 2> "$OUT/bench_stderr.log"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
    python "$R/bench.py" --no-cpu-baseline --no-traffic --no-configs > "$OUT/bench_profiled_stdout.json" 2> "$OUT/bench_profiled_stderr.log"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o t -- \
        python "$R/bench.py" --no-cpu-baseline --no-configs --no-traffic --steps 3 --warmup 1 --prewarm 0 > /dev/null 2> "$OUT/pmc_$c.log"
done
find "$OUT" -name "*.csv" | head -20
