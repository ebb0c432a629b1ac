#!/bin/bash
# counters of the tn3 weight-gradient kernel with operands in the Infinity Cache (warm) and in HBM (cold)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCC|TCP|SQ|TA|TD|GRBM)_[A-Z0-9_]+\b" | sort -u > $R/gpurun_out/r05_pmc_avail.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum TCC_EA_RD_UNCACHED_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA_RDREQ_LEVEL_sum TCC_EA_RD_MAM_DRAM_sum TCC_EA_RDREQ_DRAM_sum TCC_BUBBLE_sum"; do
  i=$((i+1))
  for cold in 0 1; do
    rm -rf $R/gpurun_out/pmc_tn3_${i}_$cold
    COLD=$cold timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tn3_${i}_$cold -- python $R/tools/tn3_cold_probe.py > $R/gpurun_out/pmc_tn3_${i}_$cold.log 2>&1
  done
done
cd $R
python - > gpurun_out/r05_tn3_pmc.txt 2>&1 <<'PY'
import collections, csv, glob
for cold in (0, 1):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/pmc_tn3_*_{cold}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "tn3_kernel" not in r["Kernel_Name"]:
                continue
            a = agg[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    print("cold" if cold else "warm", {k: round(v / n, 1) for k, (n, v) in sorted(agg.items())})
PY
rm -rf gpurun_out/pmc_tn3_*
