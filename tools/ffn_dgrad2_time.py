"""The data gradient of the second FFN conv (multi_layer_conv.py:52-63 backward; M = 35 840, N = 1536, K = 3 x 384, transposed
weights) alone: ReLU' mask from the saved activation (S) or from the row-major keep image (a3t_gemm_desc::keep_layout = 1), on the
panel kernel and on the 128-row kernel; and the forward conv that writes the image.  usage: python tools/ffn_dgrad2_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import ACT_RELU, BF16

lib = _lib.load()
B, T, cin, ff = 32, 1120, 384, 1536
M = B * T
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
x, W1, b1 = rn(M, cin).bfloat16(), rn(ff, 3, cin, sc=0.05).bfloat16(), rn(ff, sc=0.3)
h = torch.empty(M, ff, device="cuda", dtype=torch.bfloat16)
keep = torch.zeros(M * ff // 4, device="cuda", dtype=torch.uint8)
ga, W2t = rn(M, cin).bfloat16(), rn(ff, 3, cin, sc=0.05).bfloat16()
dh = torch.empty(M, ff, device="cuda", dtype=torch.bfloat16)
cs = torch.zeros(ff, device="cuda")


def timeit(name, fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:48s} {us:7.1f} us  {2.0 * M * ff * 3 * cin / us / 1e6:6.0f} TFLOP/s  [{lib.a3t_gemm_last_kernel().decode()}]")


xx = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    xx @ xx
timeit("forward conv 1", lambda: ops.conv_fwd(x, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.1, 77)))
timeit("forward conv 1 + keep image", lambda: ops.conv_fwd(x, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.1, 77),
                                                            keep_out=keep, keep_layout=1))
for mode in (1, 0):
    old = lib.a3t_gemm_pn_mode(mode)
    timeit(f"dgrad conv 2, S = h        (pn_mode {mode})", lambda: ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.625, compute=BF16, colsum=cs, S=h))
    timeit(f"dgrad conv 2, keep image   (pn_mode {mode})", lambda: ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.625, compute=BF16, colsum=cs,
                                                                                 keep_in=keep, keep_layout=1))
    timeit(f"the same conv, no mask     (pn_mode {mode})", lambda: ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.625, compute=BF16, colsum=cs))
    lib.a3t_gemm_pn_mode(old)
