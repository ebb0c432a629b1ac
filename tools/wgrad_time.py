"""The fused conv weight gradients of configs[1] alone (A3T_LIB_PATH selects an instrumented library)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16

lib = _lib.load()
B, T = 32, 1120
M = B * T
for N, Cin in ((1536, 384), (384, 1536)):
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, Cin, device="cuda").bfloat16()
    dW = torch.zeros(N, 3, Cin, device="cuda")
    fn = lambda: ops.conv_bwd_weight(dy, x, dW, T, 1, compute=BF16)
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
    print(f"[{os.path.basename(_lib.LIB_PATH)}] conv weight gradient {Cin}x3 -> {N}: {us:.1f} us ({2.0 * M * N * 3 * Cin / us / 1e6:.0f} TFLOP/s)  {lib.a3t_gemm_last_kernel().decode()}")
