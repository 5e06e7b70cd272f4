cd $GRAFT_REPO_ROOT
O=gpurun_out/exp2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_gpu_distributed.py -x -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 120 python examples/pmc_torchrun.py 2000 > $O/example.txt 2>&1; tail -8 $O/example.txt
timeout 1200 python scripts/cliff_sweep.py > $O/cliff.txt 2>&1; cat $O/cliff.txt
