# SQ-level counters of the FFN GEMM kernels (tools/pmc_gemm.py): where do the waves spend their cycles?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcg_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcg_$i -- python $R/tools/pmc_gemm.py > $R/gpurun_out/pmcg_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/pmcg_*/*/*counter_collection.csv'):
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
find gpurun_out/pmcg_* -name "*.csv" -size +1M -delete
