#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD
A3T_GEMM_8P_TN3=1 timeout 600 python tools/g8_tn_check.py > gpurun_out/r05_c5_tn3_check.txt 2>&1
A3T_GEMM_8P_TN3=0 timeout 600 python tools/g8_tn_check.py > gpurun_out/r05_c5_tn_check.txt 2>&1
tools/step_ab.sh "default:A3T_GEMM_8P_TN3=0" "tn3_heur:A3T_GEMM_8P_TN=1 A3T_GEMM_8P_TN3=2" "tn3_all:A3T_GEMM_8P_TN=1 A3T_GEMM_8P_TN3=1" "tn_slab:A3T_GEMM_8P_TN=1 A3T_GEMM_8P_TN3=0" "default_again:A3T_GEMM_8P_TN3=0" > gpurun_out/r05_c5_step_ab.txt 2>&1
