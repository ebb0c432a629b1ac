"""Cost of the fused GEMM epilogue options (ReLU' mask S, column sums, dropout) at the FFN shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16, ACT_RELU
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, T = 32, 1120
M = B * T
x = torch.randn(M, 384, device=dev).bfloat16()
W1 = (torch.randn(1536, 3, 384, device=dev) * 0.03).bfloat16()
W2 = (torch.randn(384, 3, 1536, device=dev) * 0.03).bfloat16()
h = torch.empty(M, 1536, device=dev, dtype=torch.bfloat16)
b1 = torch.randn(1536, device=dev)
print("w1 fwd plain         %7.1f us" % timeit(lambda: ops.conv_fwd(x, W1, h, T, 1, compute=BF16)))
print("w1 fwd bias+relu     %7.1f us" % timeit(lambda: ops.conv_fwd(x, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16)))
print("w1 fwd bias+relu+drop%7.1f us" % timeit(lambda: ops.conv_fwd(x, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 77))))
ga = torch.randn(M, 384, device=dev).bfloat16()
dh = torch.empty(M, 1536, device=dev, dtype=torch.bfloat16)
cs = torch.zeros(1536, device=dev)
print("w2 bwd_data plain    %7.1f us" % timeit(lambda: ops.conv_bwd_data(ga, W2, dh, T, 1, compute=BF16)))
print("w2 bwd_data S        %7.1f us" % timeit(lambda: ops.conv_bwd_data(ga, W2, dh, T, 1, S=h, alpha=0.6, compute=BF16)))
print("w2 bwd_data S+colsum %7.1f us" % timeit(lambda: ops.conv_bwd_data(ga, W2, dh, T, 1, S=h, alpha=0.6, compute=BF16, colsum=cs)))
print("w2 bwd_data colsum   %7.1f us" % timeit(lambda: ops.conv_bwd_data(ga, W2, dh, T, 1, compute=BF16, colsum=cs)))
xo = torch.empty(M, 384, device=dev)
R = torch.randn(M, 384, device=dev); b2 = torch.randn(384, device=dev)
print("w2 fwd plain(f32 out)%7.1f us" % timeit(lambda: ops.conv_fwd(h, W2, xo, T, 1, compute=BF16)))
print("w2 fwd bias+R+drop   %7.1f us" % timeit(lambda: ops.conv_fwd(h, W2, xo, T, 1, bias=b2, R=R, alpha=0.5, compute=BF16, drop=(0.2, 5))))
