mkdir -p gpurun_out
python -m pytest tests/test_gpu_attn_fused.py tests/test_gpu_e2e.py tests/test_gpu_parity_r2.py tests/test_gpu_kernels.py -x -q > gpurun_out/r06_tests_3.log 2>&1; tail -15 gpurun_out/r06_tests_3.log
bash tools/step_ab.sh "new:A3T_X=0" "bias_kernel_off:A3T_POS_BIAS_IN_KERNEL=0" "dq_acc_off:A3T_DQ_ACC=0" "both_off:A3T_POS_BIAS_IN_KERNEL=0 A3T_DQ_ACC=0" "new_again:A3T_X=0" "both_off_again:A3T_POS_BIAS_IN_KERNEL=0 A3T_DQ_ACC=0" > gpurun_out/r06_step_ab_3.txt 2>&1
cat gpurun_out/r06_step_ab_3.txt
python tools/attn_bench.py > gpurun_out/r06_attn_bench.txt 2>&1; tail -3 gpurun_out/r06_attn_bench.txt
python tools/attn_ds_time.py > gpurun_out/r06_attn_ds_time.txt 2>&1; tail -8 gpurun_out/r06_attn_ds_time.txt
