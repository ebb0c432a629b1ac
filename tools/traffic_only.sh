R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/t_pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/t_pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/t_pmc_$c.log 2>&1
done
cd $R
F=$(find gpurun_out/t_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/t_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $F $W gpurun_out/t_hbm_traffic.json > gpurun_out/t_traffic.log
find gpurun_out/t_pmc_FETCH_SIZE gpurun_out/t_pmc_WRITE_SIZE -name "*.csv" -size +1M -delete
python - <<'PY'
import json
t=json.load(open('gpurun_out/t_hbm_traffic.json'))
for k,v in t.items():
    if 'gemm' in k: print(k[:60], round(v['fetch_bytes_per_launch']/1e6,1), round(v['write_bytes_per_launch']/1e6,1))
PY
