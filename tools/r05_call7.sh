#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD
A3T_GEMM_8P_TN3=1 timeout 600 python tools/g8_tn_check.py > gpurun_out/r05_c7_tn3_check.txt 2>&1
tools/step_ab.sh "default:A3T_X=0" "tn3_off:A3T_GEMM_8P_TN3=0" "default_again:A3T_X=0" > gpurun_out/r05_c7_step_ab.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "tn3 or 8phase_tn" 2>&1 | tail -3 ) > gpurun_out/r05_c7_pytest.txt
