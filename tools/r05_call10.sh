#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5 ) > gpurun_out/r05_c10_pytest.txt
tools/step_ab.sh "default:A3T_X=0" "attn_bwd_ds:A3T_ATTN_BWD_DS=1" "fuse_ln_fwd:A3T_FUSE_LN_FWD=1" "dbd_head_major:A3T_ATTN_DBD_HM=1" "side2_off:A3T_SIDE2=0" "tn3_all:A3T_GEMM_8P_TN3=1" "default_again:A3T_X=0" "side_late:A3T_SIDE_LATE=1" "one_stream:A3T_SIDE_STREAM=0" > gpurun_out/r05_c10_step_ab.txt 2>&1
A3T_GEMM_8P_TN3=1 timeout 600 python tools/g8_tn_check.py 2>&1 | grep "^wgrad [0-9]" > gpurun_out/r05_c10_tn3_time.txt
