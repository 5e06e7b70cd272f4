#!/bin/bash
# rocprofv3 kernel summaries beyond the bench command (run on the GPU box): BASELINE config 5 as the evaluate-once
# loop and as the reference's evaluate-twice loop, and the small-dimension E-step.  -> gpurun_out/extra/*.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/extra
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, command...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o p -- "$@" > "$OUT/$name.log" 2>&1
  f=$(find "$OUT/$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
}
run cfg5_evaluate_once python "$R/examples/pmc_device_loop.py" 12500000 4
run cfg5_evaluate_twice python "$R/examples/pmc_device_loop.py" 12500000 4 evaluate-twice
run estep_d2_k32 python "$R/scripts/kbench.py" --N 4000000 --K 32 --D 2 --reps 5
run estep_d5_k32 python "$R/scripts/kbench.py" --N 4000000 --K 32 --D 5 --reps 5
run big_d128 python "$R/scripts/kbench.py" --N 1000000 --K 32 --D 128 --reps 5
ls "$OUT"/*.csv
