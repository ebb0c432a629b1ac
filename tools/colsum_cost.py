"""What the fused column-sum epilogue (bias gradients riding on data-gradient GEMMs) costs per GEMM class."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16
dev = "cuda"
bf = torch.bfloat16
B, T = 32, 1120
M = B * T


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e6


def conv(cin, cout, taps):
    dy = torch.randn(M, cout, device=dev).to(bf)
    Wk = torch.randn(cout, taps, cin, device=dev).to(bf)
    dx = torch.empty(M, cin, device=dev, dtype=bf)
    cs = torch.zeros(cin, device=dev)
    a = timeit(lambda: ops.conv_bwd_data(dy, Wk, dx, T, (taps - 1) // 2, compute=BF16))
    b = timeit(lambda: ops.conv_bwd_data(dy, Wk, dx, T, (taps - 1) // 2, compute=BF16, colsum=cs))
    print(f"conv_bwd_data {cout}->{cin} k{taps}: {a:7.1f} us plain, {b:7.1f} us with colsum", flush=True)


def lin(cin, cout):
    dy = torch.randn(M, cout, device=dev).to(bf)
    W = torch.randn(cout, cin, device=dev).to(bf)
    dx = torch.empty(M, cin, device=dev, dtype=bf)
    cs = torch.zeros(cin, device=dev)
    a = timeit(lambda: ops.linear_bwd_data(dy, W, dx, compute=BF16))
    b = timeit(lambda: ops.linear_bwd_data(dy, W, dx, compute=BF16, colsum=cs))
    print(f"linear_bwd_data {cout}->{cin}: {a:7.1f} us plain, {b:7.1f} us with colsum", flush=True)


conv(1536, 384, 3)
conv(384, 1536, 3)
lin(384, 384)
lin(768, 384)
lin(384, 1152)
