"""Summarise tools/r02_evidence.sh: per-kernel SQ counters of the FFN GEMM classes -> MFMA-pipe utilisation,
and the kernel-stats tables of the vocoder / collate legs.  Writes gpurun_out/<tag>_mfma_busy.json and
<tag>_{voc,col}_kernel_stats.csv (copied to profiles/ by hand)."""
import collections
import csv
import glob
import json
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"gpurun_out/{tag}_pmcg_*/*/*counter_collection.csv"):
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
    row["launches_per_pass"] = max(n for n, _ in d.values())
    # SQ_VALU_MFMA_BUSY_CYCLES = MFMA-pipe busy cycles summed over the 1024 SIMDs (= 32 per v_mfma_f32_32x32x16_bf16:
    # 123 863 040 = 32 x 3 870 720 MFMAs for 35840x1536x1152, checked); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so
    # utilisation of the matrix pipes = busy / (1024 SIMDs x kernel cycles) with kernel cycles = GRBM_GUI_ACTIVE / 8
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row and row.get("GRBM_GUI_ACTIVE"):
        row["mfma_util"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0)
        row["kernel_cycles"] = row["GRBM_GUI_ACTIVE"] / 8.0
    if "SQ_WAVE_CYCLES" in row and row["SQ_WAVE_CYCLES"]:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in row:
                row[c + "_frac"] = row[c] / row["SQ_WAVE_CYCLES"]
    out[k] = row
json.dump(out, open(f"gpurun_out/{tag}_mfma_busy.json", "w"), indent=1)
for k, row in out.items():
    print(k[:70], {c: round(v, 3) for c, v in row.items() if c.endswith("_frac") or c.startswith("mfma")})
for leg in ("voc", "col"):
    fs = glob.glob(f"gpurun_out/{tag}_prof_{leg}/*/*kernel_stats.csv")
    if fs:
        shutil.copy(fs[0], f"gpurun_out/{tag}_{leg}_kernel_stats.csv")
        rows = list(csv.DictReader(open(fs[0])))
        for r in rows[:8]:
            print(leg, r["Name"][:70], r["Calls"], f"{float(r['AverageNs'])/1e3:.1f}us", r["Percentage"])
