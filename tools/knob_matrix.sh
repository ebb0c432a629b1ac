# every documented A/B knob still gives a correct step: the end-to-end GPU tests under each setting
for kv in "A3T_SIDE_STREAM=0" "A3T_GEMM_WN3=0" "A3T_GEMM_WN3=1" "A3T_COLSUM_SLOTS=1" "A3T_COLSUM_SLOTS=64" "A3T_GEMM_STAGES=1" "A3T_GEMM_STAGES=2" "A3T_GEMM_8P=0" "A3T_GEMM_8P=1" "A3T_FFN_8P=0" "A3T_ATTN_DBD_HM=1" "A3T_ATTN_REGEN_MASK=0" "A3T_LN_BLOCKS=128" "A3T_PWG_FUSED=0" "A3T_FUSED_ATTN=1" "A3T_FUSED_ATTN=fwd" "A3T_FUSED_ATTN=0" "A3T_FUSED_ATTN_TRAIN=0" "A3T_FUSED_ATTN_TRAIN=2" "A3T_ATTN_FWD=16" "A3T_GEMM_COLGROUP=4" "A3T_MAIN_PRIORITY=0"; do
  echo "== $kv"; env $kv python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -1
done
