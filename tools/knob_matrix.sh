# every documented A/B knob still gives a correct step: the end-to-end GPU tests under each setting
for kv in "A3T_SIDE_STREAM=0" "A3T_GEMM_WN3=0" "A3T_GEMM_WN3=1" "A3T_GEMM_STAGES=1" "A3T_GEMM_STAGES=2" "A3T_GEMM_8P=0" "A3T_GEMM_8P=1" "A3T_FFN_8P=0" "A3T_PWG_FUSED=0" "A3T_FUSED_ATTN=fwd" "A3T_FUSED_ATTN=0" "A3T_FUSED_ATTN_TRAIN=0" "A3T_FUSED_ATTN_TRAIN=2" "A3T_ATTN_BWD_DS=0" "A3T_ATTN_SIGNED=0" "A3T_ATTN_DQ_DUAL=0" "A3T_ATTN_DBD_VIEW=0" "A3T_ATTN_DK_MAIN=0" "A3T_FFN_KEEP4=0" "A3T_GEMM_8P_TN3=0" "A3T_GEMM_TT=0" "A3T_GEMM_TT=1" "A3T_GEMM_8P_TN3=1" "A3T_WGRAD_GROUP=0" "A3T_ATTN_SPLIT=0" "A3T_SIDE_DEPTH=2"; do
  echo "== $kv"; env $kv python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -1
done
