set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_r3.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/wgrad_time.py 2>&1 | tail -4; A3T_LIB_PATH=/root/repo/a3t_amd/lib/liba3t_hip_prev.so timeout 300 python tools/wgrad_time.py 2>&1 | tail -4
P=/root/repo/a3t_amd/lib/liba3t_hip_prev.so
timeout 900 bash tools/step_ab.sh "three_phases:A3T_LIB_PATH=$P" "two_phases:A3T_X=0" "three_phases:A3T_LIB_PATH=$P" "two_phases:A3T_X=0" "three_phases:A3T_LIB_PATH=$P" "two_phases:A3T_X=0"
