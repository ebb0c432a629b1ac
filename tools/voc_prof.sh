cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_voc
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_voc -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench
print(bench.vocoder_rtf(torch.device('cuda',0)))
" > $R/gpurun_out/prof_voc.log 2>&1
cd $R
find gpurun_out/prof_voc -name "*kernel_trace.csv" -delete
python tools/prof_summary.py $(dirname $(find gpurun_out/prof_voc -name "*kernel_stats.csv" | head -1)) 12
