cd $GRAFT_REPO_ROOT
timeout 600 python scripts/step_overheads.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout 600 python scripts/step_overheads.py 1250000 2>&1 | grep -v amdgpu.ids | tail -12
