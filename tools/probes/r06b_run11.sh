cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attn_fused.py -x -q -m gpu 2>&1 | tail -4
for sh in "32 2 1120 192" "16 4 1800 128"; do for tr in 1 2; do TRAIN=$tr python tools/attn_fwd_time.py $sh 2>&1 | grep -v amdgpu; done; done
bash tools/step_ab.sh "committed:A3T_LIB_PATH=$PWD/a3t_amd/lib/liba3t_hip_base.so" "new:A3T_X=1" "committed:A3T_LIB_PATH=$PWD/a3t_amd/lib/liba3t_hip_base.so" "new:A3T_X=1"
