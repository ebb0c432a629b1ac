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
buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.a3t_attn_timing_buf.argtypes = [ctypes.c_void_p]
fn = lambda: ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345))
for _ in range(5):
    fn()
lib.a3t_attn_timing_buf(buf.data_ptr())
fn()
torch.cuda.synchronize()
r = buf.view(nwg, 4, 8).double().cpu()
steps = r[:, :, 5]
names = ["barrier->top(stamp0)", "dma issue", "S/band stages + softmax", "PV stages + handover", "wait+barrier"]
for k, n in enumerate(names):
    per = r[:, :, k] / steps
    print(f"{n:28s} mean {per.mean():8.1f}  min {per.min():8.1f}  max {per.max():8.1f}  cycles/step (s_memtime @100MHz? see total)")
tot = r[:, :, :5].sum(-1) / steps
print("total per step", tot.mean().item())
