cd /root/repo
export TMPDIR=/tmp
ALT=$PWD/a3t_amd/lib/liba3t_hip_pnfwd.so
for rep in 1 2; do
echo "== forward conv 1 on the panel kernel"; A3T_LIB_PATH=$ALT python tools/gemm_shapes.py 2>&1 | grep -v amdgpu | grep '1536, 1152\|384, 4608\|total'
echo "== committed (128-row kernel)"; python tools/gemm_shapes.py 2>&1 | grep -v amdgpu | grep '1536, 1152\|384, 4608\|total'
done
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c2
# forward-only wall time of the training forward (need_grad=True) under both libraries is not switchable in-process: time fwd here
PY
bash tools/step_ab.sh "pn_fwd:A3T_LIB_PATH=$ALT" "committed:A3T_X=1" "pn_fwd:A3T_LIB_PATH=$ALT" "committed:A3T_X=1"
