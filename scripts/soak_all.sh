#!/bin/bash
# Long soak on the GPU box: the full GPU suite repeatedly, the fuzz with 40 more seeds, the determinism stress.
cd "$(dirname "$0")/.."
fails=0
for i in $(seq 1 ${1:-8}); do
  r=$(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "suite run $i: $r"
  case "$r" in *failed*|*error*|*Error*) fails=$((fails+1));; esac
done
timeout 1500 python scripts/soak_fuzz.py 2>&1 | tail -18
timeout 900 python scripts/determinism_stress.py --reps 300 2>&1 | tail -3
echo "suite failures: $fails"
