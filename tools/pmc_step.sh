# HBM traffic per kernel of the bf16 C2 train step: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
# summarised by tools/traffic_summary.py into gpurun_out/hbm_traffic_per_kernel.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
F=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $F $W gpurun_out/hbm_traffic_per_kernel.json
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +2M -delete
