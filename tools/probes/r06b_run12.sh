cd /root/repo
timeout 900 python -m pytest tests/test_gpu_attn_fused.py -x -q -m gpu -k "single_tensor" 2>&1 | grep -B5 -A25 'Error\|assert' | head -80
