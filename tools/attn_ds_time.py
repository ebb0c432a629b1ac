"""Time a3t_attn_bwd_ds against the two kernels it replaces (dprobs GEMM + a3t_relpos_softmax_bwd) at
the benchmark shape.  usage: python tools/attn_ds_time.py [B H T dk]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16
from test_gpu_attn_fused import _inputs

B, H, T, dk = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 2, 1120, 192)))
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
dpr = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
vv = qkv.view(-1)[2 * d:]
zb = (H * T * T, T * T)


def new():
    ops.attn_bwd_ds(dctx, ctx, qkv, probs, rs, ds, dbd, B, H, T, scale, drop=drop, dbd_head_major=True)


def old():
    ops.gemm(dctx, vv, dpr, T, T, dk, d, 1, 3 * d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk), b_bs=(T * 3 * d, dk),
             c_bs=zb, compute=BF16)
    ops.relpos_softmax_bwd(probs, dpr, ds, dbd, B, H, T, scale, probs_drop=None, drop_p=drop[0], dbd_head_major=True,
                           drop_key=drop[1], rowscale=rs)


x = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    x @ x
def ds_only():
    ops.attn_bwd_ds(dctx, ctx, qkv, probs, rs, ds, dbd, B, H, T, scale, drop=drop, dbd_head_major=True)


sprobs = torch.zeros(B, H, T, T, device="cuda", dtype=torch.bfloat16)
ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, sprobs, None, rs, B, H, T, scale, drop=drop)


def ds_signed():      # the engine's default since round 6: the mask is the sign bit of the one saved tensor
    ops.attn_bwd_ds(dctx, ctx, qkv, sprobs, rs, ds, dbd, B, H, T, scale, drop=drop, dbd_head_major=True, signed_probs=True)


bs_ = T + T * T
flat_ = torch.zeros(B * H * bs_, device="cuda", dtype=torch.bfloat16)


def ds_signed_no_dbd():      # dBD not stored: read by its consumers as a view of dS (engine default)
    ops.attn_bwd_ds(dctx, ctx, qkv, sprobs, rs, flat_[T:], None, B, H, T, scale, drop=drop, signed_probs=True, ds_bs=bs_)


def ds_nodrop():
    ops.attn_bwd_ds(dctx, ctx, qkv, probs, rs, ds, dbd, B, H, T, scale, drop=(0.0, 0), dbd_head_major=True)


def gemm_only():
    ops.gemm(dctx, vv, dpr, T, T, dk, d, 1, 3 * d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk), b_bs=(T * 3 * d, dk),
             c_bs=zb, compute=BF16)


for name, fn in (("new", new), ("old", old), ("ds", ds_only), ("ds_signed", ds_signed), ("ds_signed_no_dbd", ds_signed_no_dbd), ("ds_nodrop", ds_nodrop), ("dprobs_gemm", gemm_only)):
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
    print(f"{name}: {us:.1f} us  ({3 * 2.0 * B * H * T * T / us / 1e6:.2f} TB/s of the 3 T x T tensors of the fused kernel)")
