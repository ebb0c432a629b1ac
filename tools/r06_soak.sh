# round-6 soak of the final code -> gpurun_out/r06_long_fuzz_and_train.txt, gpurun_out/r06_knob_matrix.txt
mkdir -p gpurun_out
(bash tools/long_fuzz.sh; echo "== 400 optimizer steps of configs[1]"; PYTHONPATH=. python tools/long_train.py 400 2>&1 | grep -v amdgpu | tail -20) > gpurun_out/r06_long_fuzz_and_train.txt 2>&1
bash tools/knob_matrix.sh > gpurun_out/r06_knob_matrix.txt 2>&1
tail -25 gpurun_out/r06_long_fuzz_and_train.txt; cat gpurun_out/r06_knob_matrix.txt | tail -45
