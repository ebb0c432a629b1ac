# rocprofv3 kernel trace of the bf16 C2 train step -> gpurun_out/prof_step/ (CSV stats); summarise with tools/prof_summary.py
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_step -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/prof_step.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_step -name "*kernel_trace.csv" -delete
find gpurun_out/prof_step -name "*kernel_stats.csv" | head -1 | xargs -I{} python tools/prof_summary.py {} 7
