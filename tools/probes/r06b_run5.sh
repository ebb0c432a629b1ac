cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_r3.py -x -q -m gpu -k "keep_image or panel_gemm_kernel" 2>&1 | tail -3
python tools/ffn_dgrad2_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b_ffn_dgrad2_c.txt
