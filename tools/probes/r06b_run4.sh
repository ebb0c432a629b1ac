set -x
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_r3.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06b_t4.log
cat gpurun_out/r06b_t4.log
python tools/ffn_dgrad2_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b_ffn_dgrad2_b.txt
timeout 600 bash tools/step_ab.sh "glds:A3T_PN_KEEP_OUT=0" "pn:A3T_PN_KEEP_OUT=1" "glds:A3T_PN_KEEP_OUT=0" "pn:A3T_PN_KEEP_OUT=1" "S:A3T_FFN_KEEP4=0" 2>&1 | tee gpurun_out/r06b_keep4_fwd_pn_step_ab.txt
