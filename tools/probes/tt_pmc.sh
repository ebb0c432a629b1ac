# HBM bytes and LDS bank conflicts of the streaming attention-backward GEMM against the 128-row kernel (tools/tt_gemm_check.py runs both)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ttpmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ttpmc_$c -- python $R/tools/tt_gemm_check.py > /tmp/ttpmc_$c.log 2>&1
done
rm -rf /tmp/ttpmc_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ttpmc_lds -- python $R/tools/tt_gemm_check.py > /tmp/ttpmc_lds.log 2>&1
cd $R
python tools/traffic_summary.py $(ls /tmp/ttpmc_FETCH_SIZE/*/*counter_collection.csv) $(ls /tmp/ttpmc_WRITE_SIZE/*/*counter_collection.csv) /tmp/tt_traffic.json | grep gemm
python - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/ttpmc_lds/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    row = {c: v / n for c, (n, v) in d.items()}
    print(k[:50], {c: round(v / 1e6, 2) for c, v in row.items()}, "conflict/active %.3f" % (row.get("SQ_LDS_BANK_CONFLICT", 0) / max(row.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
          "mfma_util %.3f" % (row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * row.get("GRBM_GUI_ACTIVE", 1) / 8.0)))
PY
