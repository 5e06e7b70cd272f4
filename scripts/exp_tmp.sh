cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nccl.py tests/test_gpu_distributed.py tests/test_gpu_ctx.py -q -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5
