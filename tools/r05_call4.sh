#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD
python tools/g8_tn_check.py > gpurun_out/r05_c4_tn_check_slab.txt 2>&1
A3T_GEMM_8P_TN_SLAB=0 python tools/g8_tn_check.py > gpurun_out/r05_c4_tn_check_atomic.txt 2>&1
tools/step_ab.sh "default:A3T_X=0" "tn8p_slab:A3T_GEMM_8P_TN=1" "tn8p_atomic:A3T_GEMM_8P_TN=1 A3T_GEMM_8P_TN_SLAB=0" "default_again:A3T_X=0" > gpurun_out/r05_c4_step_ab.txt 2>&1
