# Round-3 evidence set -> gpurun_out/r03_* (copied into profiles/ afterwards), ONE run of the final code:
#   bench line (all legs incl. configs[3]), rocprofv3 kernel stats of the step (two streams = as timed, one stream),
#   HBM traffic per kernel (separate FETCH_SIZE / WRITE_SIZE passes), SQ counters of the FFN GEMM classes and of the
#   8-phase kernel on the configs[3] FFN shapes (MFMA busy), kernel stats of the configs[3] step, GEMM micro-benchmarks.
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.log
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf $R/gpurun_out/r03_prof_s$mode
  A3T_SIDE_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_s$mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/r03_prof_s$mode.log 2>&1
  find $R/gpurun_out/r03_prof_s$mode -name "*kernel_trace.csv" -delete
  cp $(find $R/gpurun_out/r03_prof_s$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03_step_bf16_kernel_stats_$([ $mode = 1 ] && echo two_streams || echo one_stream).csv
done
rm -rf $R/gpurun_out/r03_prof_c4
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_c4 -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench, json
print(json.dumps(bench.c4_leg(torch.device('cuda',0), 'bf16', steps=5, warmup=2)))
" > $R/gpurun_out/r03_prof_c4.log 2>&1
find $R/gpurun_out/r03_prof_c4 -name "*kernel_trace.csv" -delete
cp $(find $R/gpurun_out/r03_prof_c4 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03_c4_step_bf16_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r03_pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/r03_pmc_$c.log 2>&1
done
cd $R
F=$(find gpurun_out/r03_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/r03_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $F $W gpurun_out/r03_hbm_traffic_per_kernel.json > gpurun_out/r03_traffic.log
find gpurun_out/r03_pmc_FETCH_SIZE gpurun_out/r03_pmc_WRITE_SIZE -name "*.csv" -size +2M -delete
# SQ counters: FFN GEMM classes of configs[1] (128x128 and panel kernels) and configs[3] (8-phase kernel)
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/r03_pmcg_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmcg_$i -- python $R/tools/pmc_gemm_r03.py > $R/gpurun_out/r03_pmcg_$i.log 2>&1
done
cd $R
python tools/r02_evidence_summary.py r03 > gpurun_out/r03_evidence.log 2>&1
find gpurun_out/r03_p* -name "*.csv" -size +1M -delete
PYTHONPATH=. python tools/g8_check.py > gpurun_out/r03_g8_check.txt 2>&1
PYTHONPATH=. python tools/g8_c4_shapes.py > gpurun_out/r03_g8_c4_shapes.txt 2>&1
PYTHONPATH=. python tools/g8_tn_check.py > gpurun_out/r03_g8_tn_check.txt 2>&1
PYTHONPATH=. python tools/mel_floor.py > gpurun_out/r03_mel_floor.txt 2>&1
tools/probes/gemm8p > gpurun_out/r03_gemm8p_probe.txt 2>&1
PYTHONPATH=. python tools/pn_check.py > gpurun_out/r03_pn_check.txt 2>&1
tools/probes/gemm_pn > gpurun_out/r03_gemm_pn_probe.txt 2>&1
tail -1 gpurun_out/r03_bench_n1.json | cut -c1-500
tail -12 gpurun_out/r03_evidence.log
