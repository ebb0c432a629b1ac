"""The attention kernels alone for rocprofv3 counter passes (tools/r05_profiles.sh): a few launches of the fused forward
(inference and training variants) and of the materialised backward's kernels at the benchmark shape."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from a3t_amd import ops
from test_gpu_attn_fused import _inputs

B, H, T, dk = 32, 2, 1120, 192
d, M = H * dk, B * T
qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=1)
ctx = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda")
probs = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
pdrop = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
rs = torch.zeros(B, H, T, device="cuda")
for _ in range(3):
    ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345))
    ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345))
sprobs = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
for _ in range(3):      # the engine's default: one sign-tagged tensor (kernel name ends in ", true>")
    ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, sprobs, None, rs, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345))
torch.cuda.synchronize()
# the score-gradient kernel of the backward (a3t_attn_bwd_ds: default at d_k >= 160 since round 5) on the same tensors
dctx = (torch.randn(M, d, device="cuda") * 0.1).bfloat16()
delta = torch.zeros(B, H, T, device="cuda")
ds = torch.empty(B, H, T, T, device="cuda", dtype=torch.bfloat16)
dbd = torch.empty(B, H, T, T, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.attn_bwd_ds(dctx, ctx, qkv, probs, rs, ds, dbd, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345))
for _ in range(3):      # mask off the sign bits (attn_bwd_ds_kernel<6, 2, 5, true>)
    ops.attn_bwd_ds(dctx, ctx, qkv, sprobs, rs, ds, dbd, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345), signed_probs=True)
bs = T + T * T
flat = torch.zeros(B * H * bs, device="cuda", dtype=torch.bfloat16)
for _ in range(3):      # the engine's default: dS only, dBD is a view of it (attn_bwd_ds_kernel<6, 2, 5, false>)
    ops.attn_bwd_ds(dctx, ctx, qkv, sprobs, rs, flat[T:], None, B, H, T, 1.0 / math.sqrt(dk), drop=(0.2, 12345), signed_probs=True, ds_bs=bs)
torch.cuda.synchronize()
