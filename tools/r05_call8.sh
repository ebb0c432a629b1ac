#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_parity_r2.py -x -q 2>&1 | tail -5 ) > gpurun_out/r05_c8_pytest.txt
tools/step_ab.sh "default:A3T_X=0" "no_group:A3T_WGRAD_GROUP=0" "default_again:A3T_X=0" "no_group_again:A3T_WGRAD_GROUP=0" > gpurun_out/r05_c8_step_ab.txt 2>&1
