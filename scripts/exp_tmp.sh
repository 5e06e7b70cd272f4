cd $GRAFT_REPO_ROOT
O=gpurun_out/exp3; mkdir -p $O
timeout 300 python scripts/power_probe.py 8 > $O/power.txt 2>&1; cat $O/power.txt
