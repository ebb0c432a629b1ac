"""Main-loop rate of the 256x256 kernels: one tile per CU (M = 65536, N = 256), large K, plain NT GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16
lib = _lib.load()
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(65536, 256, 8192), (65536, 256, 16384), (65536, 512, 8192), (35840, 1536, 1152), (65536, 1536, 1152)]:
    x = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.linear_fwd(x, W, out, compute=BF16)
    for mode in (0, 1):
        lib.a3t_gemm_p256_mode(mode)
        us = timeit(fn)
        print(f"{M}x{N}x{K} p256={mode}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s  per K-tile/tile-round {us / (K/64) / max(1, -(-((M+255)//256*((N+255)//256))//256)) * 1e3:.0f} ns [{lib.a3t_gemm_last_kernel().decode()}]", flush=True)
