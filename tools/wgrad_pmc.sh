R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  rm -rf /tmp/wp; A3T_GEMM_8P_TN=$v rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/wp -- python $R/tools/wgrad_pmc.py > /dev/null 2>&1
  python - $v <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('/tmp/wp/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            a = agg[r['Kernel_Name']][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in agg.items():
    print('8P_TN=' + sys.argv[1], k[:44], {c: round(v / n) for c, (n, v) in d.items()})
PY
done; done
