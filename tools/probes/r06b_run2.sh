set -x
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_attn_fused.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06b_t2.log
cat gpurun_out/r06b_t2.log
{
for sh in "32 2 1120 192" "16 4 1800 128"; do
  for tr in 0 1 2; do TRAIN=$tr python tools/attn_fwd_time.py $sh; done
  python tools/attn_ds_time.py $sh
done
} 2>&1 | grep -v '^+' | tee gpurun_out/r06b_attn_alone.txt
