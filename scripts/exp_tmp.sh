cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_streams.py -x -q 2>&1 | tail -12; done
