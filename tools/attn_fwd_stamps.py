"""Per-segment cycle stamps of attn_fwd32_kernel (library built with A3T_EXTRA_FLAGS=-DA3T_ATTN_TIMING)."""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from a3t_amd import ops, _lib
from test_gpu_attn_fused import _inputs

B, H, T, dk = 32, 2, 1120, 192
d, M = H * dk, B * T
qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=1)
ctx = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda")
nwg = B * H * ((T + 127) // 128)
buf = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.a3t_attn_timing_buf.argtypes = [ctypes.c_void_p]
fn = lambda: ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345))
if os.environ.get("TRAIN") == "1":
    probs = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
    pdrop = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
    rs = torch.zeros(B, H, T, device="cuda")
    fn = lambda: ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345))
for _ in range(5):
    fn()
lib.a3t_attn_timing_buf(buf.data_ptr())
fn()
torch.cuda.synchronize()
r = buf.view(nwg, 4, 32).double().cpu()
steps = r[:, :, 31:32]
per = (r[:, :, :31] / steps).mean(dim=(0, 1))
names = {0: "barrier->top", 1: "(empty)", 3: "last PV stage -> end of stages", 4: "wait + barrier"}
tot = 0.0
for k in range(31):
    if per[k] > 0:
        nm = names.get(k, f"before stage {k - 5} (= stage {k - 6})" if k >= 5 else str(k))
        print(f"{k:3d} {nm:40s} {per[k]:8.1f} cycles/step")
        tot += float(per[k])
print("total per step", tot)
