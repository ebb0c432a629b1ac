cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attn_fused.py -x -q -m gpu -k "stored_dbd or engine_training_step or one_saved" 2>&1 | tail -4
bash tools/c4_ab.sh "stored_dbd:A3T_ATTN_DBD_VIEW=0" "view:A3T_ATTN_DBD_VIEW=1" "stored_dbd:A3T_ATTN_DBD_VIEW=0" "view:A3T_ATTN_DBD_VIEW=1" 2>&1 | tee gpurun_out/r06c_dbd_view_c4_ab.txt
python tools/attn_ds_time.py 16 4 1800 128 2>&1 | grep -v amdgpu | grep signed
