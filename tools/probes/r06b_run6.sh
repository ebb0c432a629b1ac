cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_r3.py -x -q -m gpu -k "keep_image or panel_gemm_kernel" 2>&1 | tail -3
for rep in 1 2; do
A3T_LIB_PATH=$PWD/a3t_amd/lib/liba3t_hip_base.so python tools/gemm_shapes.py 2>&1 | grep -v amdgpu.ids | grep 'pn_kernel\|total' > gpurun_out/r06b_shapes_base_$rep.txt
python tools/gemm_shapes.py 2>&1 | grep -v amdgpu.ids | grep 'pn_kernel\|total' > gpurun_out/r06b_shapes_new_$rep.txt
done
paste -d'\n' gpurun_out/r06b_shapes_base_1.txt gpurun_out/r06b_shapes_new_1.txt | cut -c1-130
echo; tail -1 gpurun_out/r06b_shapes_base_2.txt gpurun_out/r06b_shapes_new_2.txt
bash tools/step_ab.sh "base:A3T_LIB_PATH=$PWD/a3t_amd/lib/liba3t_hip_base.so" "new:A3T_X=1" "base:A3T_LIB_PATH=$PWD/a3t_amd/lib/liba3t_hip_base.so" "new:A3T_X=1"
