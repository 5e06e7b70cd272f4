cd $GRAFT_REPO_ROOT
PMC_CTX_PIN_BYTES=0 timeout 600 python scripts/ctx_transfer_bench.py 10000000 child 2>&1 | grep -v amdgpu.ids
