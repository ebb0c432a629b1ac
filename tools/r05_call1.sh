#!/bin/bash
# round 5, first GPU call: suite sanity, CU-masked side-stream sweep (same box), oracle cost at full size
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r05_c1_pytest.txt
tools/step_ab.sh "default:A3T_X=0" "side128:A3T_SIDE_CUS=128" "side160:A3T_SIDE_CUS=160" "side192:A3T_SIDE_CUS=192" \
   "side224:A3T_SIDE_CUS=224" "side192_side2_64:A3T_SIDE_CUS=192 A3T_SIDE2_CUS=64" "side224_side2_128:A3T_SIDE_CUS=224 A3T_SIDE2_CUS=128" \
   "side2_128:A3T_SIDE2_CUS=128" "default_again:A3T_X=0" > gpurun_out/r05_c1_cumask.txt 2>&1
python - > gpurun_out/r05_c1_oracle_time.txt 2>&1 <<'PY'
import time, torch, os
from oracle import a3t_oracle as O
torch.set_num_threads(min(os.cpu_count() or 1, 32))
print("cpus", os.cpu_count())
oc = O.A3TConfig(enc_blocks=6, dec_blocks=6)
p = O.to_torch_state(O.procedural_state(O.param_shapes(oc), 0), requires_grad=False)
for B in (8, 32):
    batch = O.synthetic_batch(oc, B, 1000, 120, seed=1)
    t0 = time.time()
    with torch.no_grad():
        loss, _, _ = O.forward_loss(p, batch, oc, True)
    print("fwd B", B, time.time() - t0, float(loss), flush=True)
PY
