"""Time only the fused attention forward at the benchmark shape (A/B + ablation runs of tools/attn_ablate.sh)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from a3t_amd import ops
from test_gpu_attn_fused import _inputs

B, H, T, dk = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 2, 1120, 192)))
DROP = (0.2, 12345)
d, M = H * dk, B * T
qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=1)
ctx = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda")
fn = lambda: ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk), drop=DROP)
if os.environ.get("TRAIN") in ("1", "2"):       # 1: probs + dropped copy, 2: one sign-tagged tensor (the engine's default)
    probs = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
    pdrop = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16) if os.environ.get("TRAIN") == "1" else None
    rs = torch.zeros(B, H, T, device="cuda")
    fn = lambda: ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, 1.0 / math.sqrt(dk), drop=DROP)
x = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    x @ x          # clocks up
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"TRAIN={os.environ.get('TRAIN', '0')} EXP={os.environ.get('A3T_ATTN_EXP', '0')} fwd {us:.1f} us  {3 * 2.0 * B * H * T * T * dk / us / 1e6:.0f} TFLOP/s")
