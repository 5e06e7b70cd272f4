cd $GRAFT_REPO_ROOT
timeout 600 python scripts/ctx_transfer_bench.py 10000000 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_ctx.py -x -q 2>&1 | tail -3
