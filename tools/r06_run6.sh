R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
L=$R/a3t_amd/lib/liba3t_hip_dslinear.so
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for v in xcd linear; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${v}_$c
    if [ $v = linear ]; then export A3T_LIB_PATH=$L; else unset A3T_LIB_PATH; fi
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${v}_$c -- python $R/tools/attn_pmc_workload.py > /tmp/pmc_${v}_$c.log 2>&1
  done
  F=$(find /tmp/pmc_${v}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  W=$(find /tmp/pmc_${v}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  echo "== $v" >> $R/gpurun_out/r06_ds_map_traffic.txt
  python $R/tools/traffic_summary.py $F $W /tmp/t_$v.json | grep attn >> $R/gpurun_out/r06_ds_map_traffic.txt
done
cat $R/gpurun_out/r06_ds_map_traffic.txt
