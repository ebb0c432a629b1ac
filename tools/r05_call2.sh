#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_fullsize_oracle.py "tests/test_gpu_parity_r2.py::test_bf16_training_step_through_the_fused_attention_forward_against_reference_goldens" -q -s 2>&1 | grep -v Warning | tail -60 ) > gpurun_out/r05_c2_newtests.txt
python - > gpurun_out/r05_c2_oracle_bwd_time.txt 2>&1 <<'PY'
import time, torch, os, resource
from oracle import a3t_oracle as O
torch.set_num_threads(min(os.cpu_count() or 1, 32))
oc = O.A3TConfig(enc_blocks=6, dec_blocks=6)
p = O.to_torch_state(O.procedural_state(O.param_shapes(oc), 0), requires_grad=True)
for B in (4, 16, 32):
    batch = O.synthetic_batch(oc, B, 1000, 120, seed=1)
    t0 = time.time()
    loss, _, _ = O.forward_loss(p, batch, oc, True)
    loss.backward()
    print("fwd+bwd B", B, time.time() - t0, float(loss), "maxrss GB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, flush=True)
    for t in p.values():
        t.grad = None
import subprocess
print(subprocess.run("free -g | head -2", shell=True, capture_output=True, text=True).stdout)
PY
