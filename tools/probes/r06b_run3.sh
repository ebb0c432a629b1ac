set -x
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_r3.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06b_t3.log
cat gpurun_out/r06b_t3.log
python tools/ffn_dgrad2_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b_ffn_dgrad2.txt
timeout 600 bash tools/step_ab.sh "S:A3T_FFN_KEEP4=0" "keep4:A3T_FFN_KEEP4=1" "S:A3T_FFN_KEEP4=0" "keep4:A3T_FFN_KEEP4=1" 2>&1 | tee gpurun_out/r06b_keep4_step_ab.txt
