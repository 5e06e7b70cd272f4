cd $GRAFT_REPO_ROOT
bash scripts/profile_bench.sh > gpurun_out/profile_bench.log 2>&1; tail -3 gpurun_out/profile_bench.log
bash scripts/profile_counters.sh > gpurun_out/profile_counters.log 2>&1; tail -2 gpurun_out/profile_counters.log
bash scripts/profile_configs.sh "cfg3 cfg4 cfg5" > gpurun_out/profile_configs.log 2>&1; tail -3 gpurun_out/profile_configs.log
# keep what goes back under 64 MiB: the csv summaries, not the traces
find gpurun_out/prof gpurun_out/sq gpurun_out/cfgprof -name "*.csv" -size +20M -delete
find gpurun_out/prof gpurun_out/sq -name "*kernel_trace.csv" -delete
du -sh gpurun_out
