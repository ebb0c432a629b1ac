cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_attn_fused.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_e2e.py tests/test_gpu_parity_r2.py -x -q -m gpu 2>&1 | tail -4
for kv in "A3T_ATTN_DBD_VIEW=0" "A3T_ATTN_DQ_DUAL=0" "A3T_GEMM_TT=0" "A3T_GEMM_TT=1" "A3T_ATTN_SIGNED=0" "A3T_SIDE_DEPTH=8"; do echo "== $kv"; env $kv python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -1; done
