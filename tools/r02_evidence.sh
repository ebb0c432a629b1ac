# Round-2 evidence pass: per-GEMM-class SQ counters (MFMA busy), vocoder + collate kernel stats, baseline bench line.
# Usage (through gpurun): bash tools/r02_evidence.sh [tag]
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/${TAG}_pmcg_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmcg_$i -- python $R/tools/pmc_gemm.py > $R/gpurun_out/${TAG}_pmcg_$i.log 2>&1
done
rm -rf $R/gpurun_out/${TAG}_prof_voc $R/gpurun_out/${TAG}_prof_col
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_voc -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench
print(bench.vocoder_rtf(torch.device('cuda',0)))
" > $R/gpurun_out/${TAG}_prof_voc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_col -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench
print(bench.collate_leg(torch.device('cuda',0)))
" > $R/gpurun_out/${TAG}_prof_col.log 2>&1
cd $R
python tools/r02_evidence_summary.py $TAG
find gpurun_out/${TAG}_p* -name "*.csv" -size +1M -delete
