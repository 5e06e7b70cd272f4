cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2
python bench.py --no-cpu-baseline --no-traffic --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"
