# Round-2 evidence set -> gpurun_out/r02_* (copied into profiles/ afterwards):
#   bench line (all legs), rocprofv3 kernel stats of the step (two streams = as timed, one stream), HBM traffic per kernel
#   (separate FETCH_SIZE / WRITE_SIZE passes), SQ counters of the FFN GEMM classes (MFMA busy), vocoder / collate stats.
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.log
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf $R/gpurun_out/r02_prof_s$mode
  A3T_SIDE_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_s$mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile > $R/gpurun_out/r02_prof_s$mode.log 2>&1
  find $R/gpurun_out/r02_prof_s$mode -name "*kernel_trace.csv" -delete
  cp $(find $R/gpurun_out/r02_prof_s$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_step_bf16_kernel_stats_$([ $mode = 1 ] && echo two_streams || echo one_stream).csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r02_pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r02_pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile > $R/gpurun_out/r02_pmc_$c.log 2>&1
done
cd $R
F=$(find gpurun_out/r02_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/r02_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $F $W gpurun_out/r02_hbm_traffic_per_kernel.json > gpurun_out/r02_traffic.log
find gpurun_out/r02_pmc_FETCH_SIZE gpurun_out/r02_pmc_WRITE_SIZE -name "*.csv" -size +2M -delete
bash tools/r02_evidence.sh r02 > gpurun_out/r02_evidence.log 2>&1
tail -1 gpurun_out/r02_bench_n1.json | cut -c1-600
tail -12 gpurun_out/r02_evidence.log
