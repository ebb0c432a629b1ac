P=$PWD/a3t_amd/lib/liba3t_hip_prev.so
sed -i 's/--steps 20 --warmup 8/--steps 40 --warmup 8/' tools/step_ab.sh
bash tools/step_ab.sh "four_phases_of_15:A3T_LIB_PATH=$P" "two_phases_of_30:A3T_X=0" "four_phases_of_15:A3T_LIB_PATH=$P" "two_phases_of_30:A3T_X=0" "four_phases_of_15:A3T_LIB_PATH=$P" "two_phases_of_30:A3T_X=0"
bash tools/c4_ab.sh "four_phases_of_15:A3T_LIB_PATH=$P" "two_phases_of_30:A3T_X=0" "four_phases_of_15:A3T_LIB_PATH=$P" "two_phases_of_30:A3T_X=0"
