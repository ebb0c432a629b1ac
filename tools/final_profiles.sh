# Round-end evidence: full bench line, rocprofv3 kernel stats (as timed = two streams, and serial), HBM traffic passes.
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.log
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf $R/gpurun_out/prof_s$mode
  A3T_SIDE_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s$mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile > $R/gpurun_out/prof_s$mode.log 2>&1
  find $R/gpurun_out/prof_s$mode -name "*kernel_trace.csv" -delete
done
cd $R
bash tools/pmc_step.sh > gpurun_out/pmc_step.log 2>&1
tail -1 gpurun_out/bench_n1.json | cut -c1-400
