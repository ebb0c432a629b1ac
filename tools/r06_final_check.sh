# final checks of round 6 beyond the suite: the one-stream e2e tests, configs[3]'s full-size oracle parity at its own batch (B = 16), slowest tests
mkdir -p gpurun_out
(echo "== A3T_SIDE_STREAM=0 tests/test_gpu_e2e.py"; A3T_SIDE_STREAM=0 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -1) > gpurun_out/r06_final_checks.txt
(echo "== A3T_C4_B=16 tests/test_gpu_fullsize_oracle.py -k c4"; A3T_C4_B=16 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -s -k c4 2>&1 | grep -E "^\[c4\]|passed|failed|Error|error" ) > gpurun_out/r06_fullsize_c4_B16.txt 2>&1
cat gpurun_out/r06_final_checks.txt gpurun_out/r06_fullsize_c4_B16.txt
python -m pytest tests -m gpu -q --durations=12 2>&1 | grep -E "s call|passed" | head -14
