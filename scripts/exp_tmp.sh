cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
