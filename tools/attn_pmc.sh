# SQ counters of the fused attention kernels at the benchmark shape (tools/attn_bench.py) -> gpurun_out/<tag>_attn_pmc.json
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/${TAG}_apmc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_apmc_$i -- python $R/tools/attn_bench.py > $R/gpurun_out/${TAG}_apmc_$i.log 2>&1
done
cd $R
python - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f'gpurun_out/{tag}_apmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'attn_' not in k: continue
        a = agg[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
out = {}
for k, d in agg.items():
    row = {c: v / n for c, (n, v) in d.items()}
    if 'GRBM_GUI_ACTIVE' in row and 'SQ_VALU_MFMA_BUSY_CYCLES' in row:
        row['mfma_util'] = row['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * row['GRBM_GUI_ACTIVE'] / 8)
    if row.get('SQ_WAVE_CYCLES'):
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS'):
            if c in row: row[c + '_frac'] = row[c] / row['SQ_WAVE_CYCLES']
    if row.get('SQ_LDS_IDX_ACTIVE'):
        row['lds_conflict_frac'] = row.get('SQ_LDS_BANK_CONFLICT', 0) / row['SQ_LDS_IDX_ACTIVE']
    out[k] = row
    print(k[:50], {c: round(v, 3) for c, v in row.items() if c.endswith('_frac') or c in ('mfma_util',)})
json.dump(out, open(f'gpurun_out/{tag}_attn_pmc.json', 'w'), indent=1)
PY
find gpurun_out/${TAG}_apmc_* -name "*.csv" -size +1M -delete
