cd /root/repo
export TMPDIR=/tmp
for s in 41 42; do echo "== engine seed $s"; timeout 900 python tests/fuzz_engine.py $s 25 2>&1 | grep -v amdgpu | tail -3; done
echo "== 300 optimizer steps of configs[1]"; PYTHONPATH=. python tools/long_train.py 300 2>&1 | grep -v amdgpu | tail -6
for kv in "A3T_ATTN_DBD_VIEW=0" "A3T_SIDE_STREAM=0" "A3T_FUSED_ATTN_TRAIN=2"; do echo "== $kv"; env $kv python -m pytest tests/test_gpu_e2e.py tests/test_gpu_attn_fused.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -1; done
