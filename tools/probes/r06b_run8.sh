cd /root/repo
export TMPDIR=/tmp
bash tools/step_ab.sh "default:A3T_X=0" "wn3_forced:A3T_GEMM_WN3=1" "default:A3T_X=0" "wn3_forced:A3T_GEMM_WN3=1"
A3T_GEMM_WN3=1 python tools/gemm_shapes.py 2>&1 | grep 'glds_kernel<0, 1\|glds_kernel<0, 2, 2, 0' | head -8
python tools/gemm_shapes.py 2>&1 | grep 'glds_kernel<0, 1\|glds_kernel<0, 2, 2, 0' | head -8
