# SQ counters of attn_bwd_ds_kernel (tools/attn_ds_time.py as the workload) -> gpurun_out/attn_ds_pmc.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/dspmc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/dspmc_$i -- python $R/tools/attn_ds_time.py > /tmp/dspmc_$i.log 2>&1
done
python - <<'PY' > $R/gpurun_out/attn_ds_pmc.txt
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/dspmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_bwd_ds" not in k and "softmax_bwd" not in k:
            continue
        a = agg[k[:60]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print(f"   {c:28s} {v / n:16.1f}")
PY
cat $R/gpurun_out/attn_ds_pmc.txt
