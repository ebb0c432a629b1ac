"""Fused legacy rel-pos attention vs the materialised path at the benchmark shape (B=32, H=2, T=1120, d_k=192)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16
from test_gpu_attn_fused import _inputs, _materialised

B, H, T, dk = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 2, 1120, 192)))
DROP = (0.2, 12345)
d, M = H * dk, B * T
scale = 1.0 / math.sqrt(dk)
qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=1)
dctx = torch.randn(M, d, device="cuda").bfloat16()
ctx = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda")
delta = torch.zeros(B, H, T, device="cuda")


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


unit = 2.0 * B * H * T * T * dk
res = {}
res["fwd_fused_us"] = timeit(lambda: ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, scale, drop=DROP))
res["fwd_materialised_us"] = timeit(lambda: _materialised(qkv, qu, qv, P, keymask, B, H, T, dk, DROP))
zu, zv = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
res["fwd_fused_bias_in_kernel_us"] = timeit(lambda: ops.attn_fwd(None, None, qkv, P, keymask, ctx, lse, B, H, T, scale, drop=DROP, pos_bias=(zu, zv)))
res["add_pos_bias_us"] = timeit(lambda: ops.add_pos_bias(qkv, zu, zv, qu, qv))
res["fwd_tflops"] = 3 * unit / res["fwd_fused_us"] / 1e6
print({k: round(v, 1) for k, v in res.items()})
