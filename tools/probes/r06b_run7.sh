cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_attn_fused.py -x -q -m gpu -k "two_products or streaming or engine_training_step or one_saved or single_tensor" 2>&1 | tail -6
bash tools/step_ab.sh "two_launches:A3T_ATTN_DQ_DUAL=0" "one_launch:A3T_ATTN_DQ_DUAL=1" "two_launches:A3T_ATTN_DQ_DUAL=0" "one_launch:A3T_ATTN_DQ_DUAL=1" "two_launches:A3T_ATTN_DQ_DUAL=0" "one_launch:A3T_ATTN_DQ_DUAL=1" 2>&1 | tee gpurun_out/r06b_dq_dual_step_ab.txt
