mkdir -p gpurun_out
sed -i 's/--steps 20 --warmup 8/--steps 60 --warmup 10/' tools/step_ab.sh
bash tools/step_ab.sh "new:A3T_X=0" "both_off:A3T_POS_BIAS_IN_KERNEL=0 A3T_DQ_ACC=0" "bias_off:A3T_POS_BIAS_IN_KERNEL=0" "dq_off:A3T_DQ_ACC=0" "new:A3T_X=0" "both_off:A3T_POS_BIAS_IN_KERNEL=0 A3T_DQ_ACC=0" "bias_off:A3T_POS_BIAS_IN_KERNEL=0" "dq_off:A3T_DQ_ACC=0" "new:A3T_X=0" "both_off:A3T_POS_BIAS_IN_KERNEL=0 A3T_DQ_ACC=0" > gpurun_out/r06_step_ab_5.txt 2>&1
cat gpurun_out/r06_step_ab_5.txt
