cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attn_fused.py -x -q -m gpu -k "view_of_ds or stored_dbd or two_products or engine_training_step" 2>&1 | tail -12
python tools/attn_ds_time.py 2>&1 | grep -v amdgpu
bash tools/step_ab.sh "stored_dbd:A3T_ATTN_DBD_VIEW=0" "view:A3T_ATTN_DBD_VIEW=1" "stored_dbd:A3T_ATTN_DBD_VIEW=0" "view:A3T_ATTN_DBD_VIEW=1" 2>&1 | tee gpurun_out/r06c_dbd_view_step_ab.txt
