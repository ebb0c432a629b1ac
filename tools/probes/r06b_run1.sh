set -x
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attn_fused.py -x -q -m gpu -k "single_tensor or one_saved or engine_training_step" 2>&1 | tail -15 > gpurun_out/r06b_t1.log
cat gpurun_out/r06b_t1.log
timeout 600 bash tools/step_ab.sh "two:A3T_ATTN_SIGNED=0" "one:A3T_ATTN_SIGNED=1" "two:A3T_ATTN_SIGNED=0" "one:A3T_ATTN_SIGNED=1" 2>&1 | tee gpurun_out/r06b_signed_step_ab.txt
timeout 600 bash tools/c4_ab.sh "two:A3T_ATTN_SIGNED=0" "one:A3T_ATTN_SIGNED=1" "two:A3T_ATTN_SIGNED=0" "one:A3T_ATTN_SIGNED=1" 2>&1 | tee gpurun_out/r06b_signed_c4_ab.txt
