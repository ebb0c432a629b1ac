mkdir -p gpurun_out
(echo "== A3T_SIDE_STREAM=0 tests/test_gpu_e2e.py"; A3T_SIDE_STREAM=0 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | grep -E "passed|failed") > gpurun_out/r06_final_checks.txt
cat gpurun_out/r06_final_checks.txt
bash tools/step_ab.sh "default:A3T_X=0" "optimizer_on_side_stream_probe:A3T_EXP_OPT_SIDE=1" "default2:A3T_X=0" "optimizer_on_side_stream_probe2:A3T_EXP_OPT_SIDE=1"
