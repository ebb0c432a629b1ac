# Round-4 evidence set -> gpurun_out/r04_* (copied into profiles/ afterwards), ONE run of the final code:
#   bench line (all legs), rocprofv3 kernel stats of the step (two streams = as timed, one stream), kernel stats of the
#   configs[3] step, HBM traffic per kernel (separate FETCH_SIZE / WRITE_SIZE passes), SQ counters of the fused attention
#   forward (MFMA busy, waits, LDS conflicts), attention micro-benchmarks, per-stage cycle stamps, step A/B logs, collate.
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.log
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf $R/gpurun_out/r04_prof_s$mode
  A3T_SIDE_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof_s$mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/r04_prof_s$mode.log 2>&1
  find $R/gpurun_out/r04_prof_s$mode -name "*kernel_trace.csv" -delete
  cp $(find $R/gpurun_out/r04_prof_s$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04_step_bf16_kernel_stats_$([ $mode = 1 ] && echo two_streams || echo one_stream).csv
  rm -rf $R/gpurun_out/r04_prof_s$mode
done
rm -rf $R/gpurun_out/r04_prof_c4
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof_c4 -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench, json
print(json.dumps(bench.c4_leg(torch.device('cuda',0), 'bf16', steps=5, warmup=2)))
" > $R/gpurun_out/r04_prof_c4.log 2>&1
find $R/gpurun_out/r04_prof_c4 -name "*kernel_trace.csv" -delete
cp $(find $R/gpurun_out/r04_prof_c4 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04_c4_step_bf16_kernel_stats.csv
rm -rf $R/gpurun_out/r04_prof_c4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r04_pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r04_pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/r04_pmc_$c.log 2>&1
done
cd $R
F=$(find gpurun_out/r04_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/r04_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $F $W gpurun_out/r04_hbm_traffic_per_kernel.json > gpurun_out/r04_traffic.log
rm -rf gpurun_out/r04_pmc_FETCH_SIZE gpurun_out/r04_pmc_WRITE_SIZE
# SQ counters of the fused attention forward (inference and training variants)
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/r04_pmca_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r04_pmca_$i -- python $R/tools/attn_pmc_r04.py > $R/gpurun_out/r04_pmca_$i.log 2>&1
done
cd $R
python - <<'PY' > gpurun_out/r04_attn_pmc.log 2>&1
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/r04_pmca_*/*/*counter_collection.csv"):
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
json.dump(out, open("gpurun_out/r04_attn_pmc.json", "w"), indent=1)
PY
rm -rf gpurun_out/r04_pmca_*
bash tools/r04_gemm_pmc.sh > gpurun_out/r04_gemm_pmc.log 2>&1
cd $R
python tools/attn_bench.py > gpurun_out/r04_attn_bench.txt 2>&1
for v in 32 16; do A3T_ATTN_FWD=$v python tools/attn_fwd_time.py >> gpurun_out/r04_attn_bench.txt 2>&1; done
TRAIN=1 python tools/attn_fwd_time.py >> gpurun_out/r04_attn_bench.txt 2>&1
python tools/attn_fwd_time.py 16 4 1800 128 >> gpurun_out/r04_attn_bench.txt 2>&1
TRAIN=1 python tools/attn_fwd_time.py 16 4 1800 128 >> gpurun_out/r04_attn_bench.txt 2>&1
A3T_ATTN_FWD=16 python tools/attn_fwd_time.py 16 4 1800 128 >> gpurun_out/r04_attn_bench.txt 2>&1
python tools/collate_time.py > gpurun_out/r04_collate_time.txt 2>&1
bash tools/step_ab.sh "default:A3T_X=0" "materialised_attention_forward:A3T_FUSED_ATTN_TRAIN=0" "one_stream:A3T_SIDE_STREAM=0" "without_ffn_weight_gradients(bound,wrong_gradients):A3T_EXPERIMENT_SKIP_FFN_WGRAD=1" "without_dprobs_softmaxbwd_dqu(bound,wrong_gradients):A3T_EXPERIMENT_SKIP_ATTN_BWD=1" "without_dprobs_softmaxbwd(bound,wrong_gradients):A3T_EXPERIMENT_SKIP_ATTN_BWD=2" "default_again:A3T_X=0" > gpurun_out/r04_step_ab.txt 2>&1
bash tools/c4_ab.sh "default:A3T_X=0" "materialised_attention_forward:A3T_FUSED_ATTN_TRAIN=0" > gpurun_out/r04_c4_ab.txt 2>&1
tail -c 600 gpurun_out/r04_bench_n1.json; cat gpurun_out/r04_attn_pmc.log | tail -4; cat gpurun_out/r04_step_ab.txt
