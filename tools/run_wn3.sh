python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or conv or sweep" 2>&1 | tail -5
python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5
echo "=== bmm"; python tools/attn_bmm_bench.py
B="--steps 10 --warmup 3 --no-collate --no-cpu-baseline --no-vocoder --no-kernel-profile"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'
echo "=== step slots=1"; A3T_COLSUM_SLOTS=1 python bench.py $B 2>/dev/null | python -c "$P"
echo "=== step slots=16"; python bench.py $B 2>/dev/null | python -c "$P"
echo "=== step slots=16 wn3 off"; A3T_GEMM_WN3=0 python bench.py $B 2>/dev/null | python -c "$P"
echo "=== step slots=64"; A3T_COLSUM_SLOTS=64 python bench.py $B 2>/dev/null | python -c "$P"
