mkdir -p gpurun_out
L=$PWD/a3t_amd/lib/liba3t_hip_dslinear.so
(echo "== xcd map"; python tools/attn_ds_time.py 2>/dev/null | grep "^ds\|^new"; echo "== linear map"; A3T_LIB_PATH=$L python tools/attn_ds_time.py 2>/dev/null | grep "^ds\|^new"; echo "== xcd map again"; python tools/attn_ds_time.py 2>/dev/null | grep "^ds\|^new") > gpurun_out/r06_ds_map_ab.txt 2>&1
cat gpurun_out/r06_ds_map_ab.txt
python tools/attn_bench.py 2>/dev/null | tail -1 > gpurun_out/r06_attn_bench.txt; cat gpurun_out/r06_attn_bench.txt
bash tools/step_ab.sh "xcd:A3T_X=0" "linear:A3T_LIB_PATH=$L" "xcd2:A3T_X=0" "linear2:A3T_LIB_PATH=$L" > gpurun_out/r06_step_ab_4.txt 2>&1
cat gpurun_out/r06_step_ab_4.txt
