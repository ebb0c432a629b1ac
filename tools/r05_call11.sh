#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 ) > gpurun_out/r05_c11_pytest.txt
tools/step_ab.sh "default:A3T_X=0" "materialised_score_gradients:A3T_ATTN_BWD_DS=0" "default_again:A3T_X=0" "materialised_again:A3T_ATTN_BWD_DS=0" "ds_and_fused_ln_fwd:A3T_FUSE_LN_FWD=1" > gpurun_out/r05_c11_step_ab.txt 2>&1
bash tools/c4_ab.sh "default:A3T_X=0" "materialised_score_gradients:A3T_ATTN_BWD_DS=0" "no_tn3:A3T_GEMM_8P_TN3=0" "no_group:A3T_WGRAD_GROUP=0" "default_again:A3T_X=0" > gpurun_out/r05_c11_c4_ab.txt 2>&1
