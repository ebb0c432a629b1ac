# SQ counters of the FFN GEMM classes (workload: tools/gemm_pmc_workload.py = configs[1] on the 128x128 and panel kernels, configs[3]
# on the 8-phase kernel) -> gpurun_out/${A3T_ROUND:-r05}_gemm_mfma_busy.json; part of the per-round evidence set (tools/r06_profiles.sh calls it)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rm -rf /tmp/pmcg_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcg_$i -- python $R/tools/gemm_pmc_workload.py > /tmp/pmcg_$i.log 2>&1
done
cd $R
python - <<'PY'
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pmcg_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k:
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {}
for k, d in agg.items():
    row = {c: v / n for c, (n, v) in d.items()}
    if row.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in row:       # busy cycles summed over 1024 SIMDs / (cycles x SIMDs)
        row["mfma_util"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0)
    if row.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in row:
                row[c + "_frac"] = row[c] / row["SQ_WAVE_CYCLES"]
    out[k] = row
    print(k[:70], {c: round(v, 3) for c, v in row.items() if c.endswith("_frac") or c == "mfma_util"})
json.dump(out, open("gpurun_out/" + __import__("os").environ.get("A3T_ROUND", "r05") + "_gemm_mfma_busy.json", "w"), indent=1)
PY
