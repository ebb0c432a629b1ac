# L2 / L1 hit rates and memory latency counters of the FFN GEMM kernels (tools/pmc_gemm.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "Name:\s*\(TCC\|TCP\|TA_\|TD_\)[A-Za-z0-9_]*" | sort -u > $R/gpurun_out/avail_cache_counters.txt
wc -l $R/gpurun_out/avail_cache_counters.txt
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RD_LAT_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TA_BUSY_sum TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_$i -- python $R/tools/pmc_gemm.py > $R/gpurun_out/pmcc_$i.log 2>&1
  tail -2 $R/gpurun_out/pmcc_$i.log
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/pmcc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm' not in k: continue
        a = agg[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in agg.items():
    print(k[:60])
    for c, (n, v) in sorted(d.items()):
        print(f"   {c:34s} {v/n:16.0f}")
PY
find gpurun_out/pmcc_* -name "*.csv" -size +1M -delete
