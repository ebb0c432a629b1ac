# kernel trace (start/end per launch) of a few steps -> gpurun_out/trace_step.csv  (tools/trace_analyse.py reads it)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_step_d
env "$@" rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_step_d -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/trace_step.log 2>&1
cp $(find $R/gpurun_out/trace_step_d -name "*kernel_trace.csv" | head -1) $R/gpurun_out/trace_step.csv
rm -rf $R/gpurun_out/trace_step_d
tail -2 $R/gpurun_out/trace_step.log
