#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$PWD
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_parity_r2.py tests/test_gpu_r3.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5 ) > gpurun_out/r05_c9_pytest.txt
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf $R/gpurun_out/r05_prof_s$mode
  A3T_SIDE_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_s$mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/r05_prof_s$mode.log 2>&1
  find $R/gpurun_out/r05_prof_s$mode -name "*kernel_trace.csv" -delete
  cp $(find $R/gpurun_out/r05_prof_s$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_c9_kernel_stats_$([ $mode = 1 ] && echo two_streams || echo one_stream).csv
  rm -rf $R/gpurun_out/r05_prof_s$mode
done
