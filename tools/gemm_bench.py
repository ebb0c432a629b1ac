"""Micro-benchmark + numerics check of the GEMM kernels on the A3T shapes (run on the GPU box).
torch matmul is used here only as an on-device checker."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16, F32

dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-9))


def run(B, T, Cin, Cout, taps, dt):
    M = B * T
    x = torch.randn(M, Cin, device=dev).to(dt)
    Wk = (torch.randn(Cout, taps, Cin, device=dev) * (Cin * taps) ** -0.5).to(dt)
    dy = torch.randn(M, Cout, device=dev).to(dt)
    out = torch.empty(M, Cout, device=dev, dtype=dt)
    pad = (taps - 1) // 2
    fl = 2.0 * M * Cout * Cin * taps
    # reference via unfold on fp32 copies
    xf = x.float().view(B, T, Cin)
    cols = torch.cat([torch.nn.functional.pad(xf, (0, 0, pad - t, t - pad))[:, pad - t + (t - pad if t > pad else 0):, :][:, :T] if False else
                      torch.roll(xf, shifts=pad - t, dims=1) * ((torch.arange(T, device=dev) + (t - pad) >= 0) & (torch.arange(T, device=dev) + (t - pad) < T))[None, :, None]
                      for t in range(taps)], dim=-1).view(M, taps * Cin)
    ref = cols @ Wk.float().view(Cout, -1).t()
    t = timeit(lambda: ops.conv_fwd(x, Wk, out, T, pad, compute=BF16))
    print(f"conv_fwd  M={M} N={Cout} K={taps*Cin} {str(dt)[6:]}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF  relerr {rel(out, ref):.2e}")
    dx = torch.empty(M, Cin, device=dev, dtype=dt)
    t = timeit(lambda: ops.conv_bwd_data(dy, Wk, dx, T, pad, compute=BF16))
    colsd = torch.cat([torch.roll(dy.float().view(B, T, Cout), shifts=t_ - pad, dims=1) * ((torch.arange(T, device=dev) - (t_ - pad) >= 0) & (torch.arange(T, device=dev) - (t_ - pad) < T))[None, :, None]
                       for t_ in range(taps)], dim=-1).view(M, taps * Cout)
    refdx = colsd @ Wk.float().permute(1, 0, 2).reshape(taps * Cout, Cin)
    print(f"conv_bwd_data                      : {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF  relerr {rel(dx, refdx):.2e}")
    dW = torch.zeros(Cout, taps, Cin, device=dev)
    def f():
        ops.conv_bwd_weight(dy, x, dW, T, pad, compute=BF16)
    dW.zero_(); f(); torch.cuda.synchronize()
    refdW = (dy.float().t() @ cols).view(Cout, taps, Cin)
    err = rel(dW, refdW)
    t = timeit(f)
    print(f"conv_bwd_weight                    : {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF  relerr {err:.2e}")


for dt in (torch.bfloat16, torch.float32):
    run(32, 1120, 384, 1536, 3, dt)
    run(32, 1120, 1536, 384, 3, dt)
    run(32, 1120, 384, 384, 1, dt)
run(4, 1000, 80, 256, 5, torch.bfloat16)
