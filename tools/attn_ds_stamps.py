"""Per-phase cycle totals of attn_bwd_ds_kernel (library built with A3T_EXTRA_FLAGS=-DA3T_DS_TIMING):
0 loop top -> 1 inputs waited -> 2 image written -> [3 V tile waited + barrier -> 4 tile computed]* -> 5 next inputs requested
-> 6 dS rows stored -> 7 dBD stored."""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from a3t_amd import _lib, ops
from test_gpu_attn_fused import _inputs

B, H, T, dk = 32, 2, 1120, 192
d, M = H * dk, B * T
scale = 1.0 / math.sqrt(dk)
drop = (0.2, 777)
qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=1)
ctx = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda")
probs = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
pdrop = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
rs = torch.zeros(B, H, T, device="cuda")
ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, scale, drop=drop)
dctx = torch.randn(M, d, device="cuda").bfloat16()
delta = torch.zeros(B, H, T, device="cuda")
ds = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
dbd = torch.zeros(H, B, T, T, device="cuda", dtype=torch.bfloat16)
lib = _lib.load()
lib.a3t_attn_timing_buf.argtypes = [ctypes.c_void_p]
lib.a3t_attn_timing_buf.restype = None
buf = torch.zeros(512 * 4 * 8, dtype=torch.int64, device="cuda")
lib.a3t_attn_timing_buf(buf.data_ptr())
for _ in range(3):
    ops.attn_bwd_ds(dctx, ctx, qkv, probs, rs, ds, dbd, B, H, T, scale, drop=drop, dbd_head_major=True)
torch.cuda.synchronize()
t = buf.view(512 * 4, 8).double().cpu()
tot = t.sum(1)
names = ["loop bookkeeping", "wait inputs (+stores)", "image write", "wait V + barrier", "tile compute", "request next inputs",
         "dS row stores", "dBD stores"]
print("cycles per wave (mean over %d waves): total %.0f" % (t.shape[0], float(tot.mean())))
for k, n in enumerate(names):
    print("  %-24s %9.0f  %5.1f %%" % (n, float(t[:, k].mean()), 100.0 * float(t[:, k].mean()) / float(tot.mean())))
