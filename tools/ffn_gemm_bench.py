"""The six GEMM launches of one conv-FFN (MultiLayeredConv1d k=3, d=384, ff=1536) at the benchmark token count, exactly as
the engine issues them (fused epilogues included).  Run under different A3T_GEMM_* settings to compare kernel variants."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import ACT_RELU, BF16

dev = "cuda"
B, T, d, ff = 32, 1120, 384, 1536
M = B * T
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
y = rn(M, d).bfloat16()
x = rn(M, d)
W1 = rn(ff, 3, d, sc=0.03).bfloat16()
W2 = rn(d, 3, ff, sc=0.02).bfloat16()
b1, b2 = rn(ff), rn(d)
h = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
xo = torch.empty(M, d, device=dev)
ga = rn(M, d).bfloat16()
dh = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
dy = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
gW1, gW2 = torch.zeros(ff, 3, d, device=dev), torch.zeros(d, 3, ff, device=dev)
gb1 = torch.zeros(ff, device=dev)
DR = (0.2, 777)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = [
    ("F1 conv1 fwd  NT 35840x1536x1152 bias+relu+drop", lambda: ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=DR)),
    ("F2 conv2 fwd  NT 35840x384x4608 bias+drop+res  ", lambda: ops.conv_fwd(h, W2, xo, T, 1, bias=b2, R=x, alpha=0.5, compute=BF16, drop=DR)),
    ("B1 conv2 dgrad NN 35840x1536x1152 mask+colsum   ", lambda: ops.conv_bwd_data(ga, W2, dh, T, 1, S=h, alpha=0.625, compute=BF16, colsum=gb1)),
    ("B2 conv2 wgrad TN 384x4608x35840               ", lambda: ops.conv_bwd_weight(ga, h, gW2, T, 1, alpha=0.5, compute=BF16)),
    ("B3 conv1 dgrad NN 35840x384x4608               ", lambda: ops.conv_bwd_data(dh, W1, dy, T, 1, compute=BF16)),
    ("B4 conv1 wgrad TN 1536x1152x35840              ", lambda: ops.conv_bwd_weight(dh, y, gW1, T, 1, compute=BF16)),
]
fl = 2.0 * M * d * ff * 3
tot = 0.0
from a3t_amd import _lib
for name, fn in cases:
    us = timeit(fn)
    tot += us
    print(f"{name}: {us:7.1f} us {fl / us / 1e6:7.1f} TFLOP/s   [{_lib.load().a3t_gemm_last_kernel().decode()}]")
print(f"sum {tot:.1f} us ({6 * fl / tot / 1e6:.1f} TFLOP/s), env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("A3T_")))
