#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_parity_r2.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 ) > gpurun_out/r05_c12_pytest.txt
bash tools/c4_ab.sh "default:A3T_X=0" "atomic_epilogue:A3T_GEMM_8P_TN_SLAB=0" "ds_on:A3T_ATTN_BWD_DS=1" "tn3_all:A3T_GEMM_8P_TN3=1" "default_again:A3T_X=0" > gpurun_out/r05_c12_c4_ab.txt 2>&1
tools/step_ab.sh "default:A3T_X=0" "default_again:A3T_X=0" > gpurun_out/r05_c12_step_ab.txt 2>&1
