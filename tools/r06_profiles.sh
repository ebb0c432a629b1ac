#!/bin/bash
# Round-6 evidence set -> gpurun_out/r06_* (copied into profiles/ afterwards), ONE run of the final code on ONE box:
#   bench line (all legs), rocprofv3 kernel stats of the step (two streams = as timed, one stream), kernel stats of the configs[3]
#   step, HBM traffic per kernel (separate FETCH_SIZE / WRITE_SIZE passes; bench.py reads r06_hbm_traffic_per_kernel.json), SQ
#   counters of the GEMM classes and of the attention kernels, weight-gradient kernel timings (warm / cold operands), attention
#   micro-benchmarks, collate, step A/B logs for both configurations.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export A3T_ROUND=r06 PYTHONPATH=$R
cd $R; mkdir -p gpurun_out
python bench.py > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.log
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf $R/gpurun_out/r06_prof_s$mode
  A3T_SIDE_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_s$mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/r06_prof_s$mode.log 2>&1
  find $R/gpurun_out/r06_prof_s$mode -name "*kernel_trace.csv" -delete
  cp $(find $R/gpurun_out/r06_prof_s$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06_step_bf16_kernel_stats_$([ $mode = 1 ] && echo two_streams || echo one_stream).csv
  rm -rf $R/gpurun_out/r06_prof_s$mode
done
rm -rf $R/gpurun_out/r06_prof_c4
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_c4 -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench, json
print(json.dumps(bench.c4_leg(torch.device('cuda',0), 'bf16', steps=5, warmup=2)))
" > $R/gpurun_out/r06_prof_c4.log 2>&1
find $R/gpurun_out/r06_prof_c4 -name "*kernel_trace.csv" -delete
cp $(find $R/gpurun_out/r06_prof_c4 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06_c4_step_bf16_kernel_stats.csv
rm -rf $R/gpurun_out/r06_prof_c4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r06_pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r06_pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/r06_pmc_$c.log 2>&1
done
cd $R
F=$(find gpurun_out/r06_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/r06_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $F $W gpurun_out/r06_hbm_traffic_per_kernel.json > gpurun_out/r06_traffic.log
rm -rf gpurun_out/r06_pmc_FETCH_SIZE gpurun_out/r06_pmc_WRITE_SIZE
# SQ counters of the attention kernels (fused forward, inference and training variants; score-gradient kernel)
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/r06_pmca_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r06_pmca_$i -- python $R/tools/attn_pmc_workload.py > $R/gpurun_out/r06_pmca_$i.log 2>&1
done
cd $R
python - <<'PY' > gpurun_out/r06_attn_pmc.log 2>&1
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/r06_pmca_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k:
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {}
for k, d in agg.items():
    row = {c: v / n for c, (n, v) in d.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row and row.get("GRBM_GUI_ACTIVE"):
        row["mfma_util"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0)
    if row.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if c in row:
                row[c + "_frac"] = row[c] / row["SQ_WAVE_CYCLES"]
    out[k] = row
    print(k[:80], {c: round(v, 4) for c, v in row.items() if c.endswith("_frac") or c.startswith("mfma") or "CONFLICT" in c})
json.dump(out, open("gpurun_out/r06_attn_pmc.json", "w"), indent=1)
PY
rm -rf gpurun_out/r06_pmca_*
bash tools/gemm_pmc.sh > gpurun_out/r06_gemm_pmc.log 2>&1
cd $R
A3T_GEMM_8P_TN3=1 python tools/g8_tn_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_tn3_check.txt
python tools/attn_bench.py > gpurun_out/r06_attn_bench.txt 2>&1
TRAIN=1 python tools/attn_fwd_time.py >> gpurun_out/r06_attn_bench.txt 2>&1
TRAIN=2 python tools/attn_fwd_time.py >> gpurun_out/r06_attn_bench.txt 2>&1
python tools/ffn_dgrad2_time.py >> gpurun_out/r06_attn_bench.txt 2>&1
python tools/attn_ds_time.py >> gpurun_out/r06_attn_bench.txt 2>&1
python tools/collate_time.py > gpurun_out/r06_collate_time.txt 2>&1
bash tools/step_ab.sh "default:A3T_X=0" "weight_gradients_on_the_128x128_kernel:A3T_GEMM_8P_TN3=0" "no_grouped_linear_weight_gradients:A3T_WGRAD_GROUP=0" "attention_products_on_the_128_row_kernel:A3T_GEMM_TT=0" "two_saved_probability_tensors:A3T_ATTN_SIGNED=0" "stored_dbd_matrix:A3T_ATTN_DBD_VIEW=0" "dk_on_the_second_side_queue:A3T_ATTN_DK_MAIN=0" "dq_as_two_launches:A3T_ATTN_DQ_DUAL=0" "ffn_mask_from_the_saved_activation:A3T_FFN_KEEP4=0" "both:A3T_ATTN_SIGNED=0 A3T_FFN_KEEP4=0" "materialised_score_gradients:A3T_ATTN_BWD_DS=0" "materialised_attention_forward:A3T_FUSED_ATTN_TRAIN=0" "one_stream:A3T_SIDE_STREAM=0" "default_again:A3T_X=0" > gpurun_out/r06_step_ab.txt 2>&1
bash tools/c4_ab.sh "default:A3T_X=0" "stored_dbd_matrix:A3T_ATTN_DBD_VIEW=0" "two_saved_probability_tensors:A3T_ATTN_SIGNED=0" "materialised_score_gradients:A3T_ATTN_BWD_DS=0" "attention_products_on_the_128_row_kernel:A3T_GEMM_TT=0" "128x384_tiles_everywhere:A3T_GEMM_8P_TN3=1" "default_again:A3T_X=0" > gpurun_out/r06_c4_ab_final.txt 2>&1
bash tools/trace_step.sh > gpurun_out/r06_trace.log 2>&1
python tools/trace_analyse.py gpurun_out/trace_step.csv > gpurun_out/r06_trace_analysis.txt 2>&1
rm -f gpurun_out/trace_step.csv
python tools/gemm_shapes.py > gpurun_out/r06_gemm_shapes.txt 2>&1
python tools/hbm_bound_table.py r06 gpurun_out > gpurun_out/r06_hbm_bound_kernels.txt 2>&1
tail -c 400 gpurun_out/r06_bench_n1.json; cat gpurun_out/r06_step_ab.txt
