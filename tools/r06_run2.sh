mkdir -p gpurun_out
python -m pytest tests/test_gpu_attn_fused.py -x -q > gpurun_out/r06_tests_2.log 2>&1; tail -2 gpurun_out/r06_tests_2.log
P=$PWD/a3t_amd/lib/liba3t_hip_prev.so
bash tools/step_ab.sh "prev_lib:A3T_LIB_PATH=$P" "new_lib_xcd_ds:A3T_X=0" "split2:A3T_TN3_SPLIT_MULT=2" "split3:A3T_TN3_SPLIT_MULT=3" "split4:A3T_TN3_SPLIT_MULT=4" \
  "defer_cnv:A3T_FFN_WGRAD_AT=cnv" "defer_cnv_mha:A3T_FFN_WGRAD_AT=cnv+mha" "defer_cnv_split2:A3T_FFN_WGRAD_AT=cnv A3T_TN3_SPLIT_MULT=2" \
  "prev_lib_again:A3T_LIB_PATH=$P" "new_again:A3T_X=0" > gpurun_out/r06_step_ab_2.txt 2>&1
cat gpurun_out/r06_step_ab_2.txt
