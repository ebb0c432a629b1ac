"""Split-K sweep of the small Linear weight gradients of configs[1] (dW[N][K] += dy[M][N]^T x[M][K], M = 35840 tokens)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import ACC_ATOMIC, BF16

lib = _lib.load()
M = 32 * 1120
for N, K in ((384, 384), (768, 384), (1152, 384), (384, 1120)):
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    dW = torch.zeros(N, K, device="cuda")
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    line = []
    for sk in (8, 16, 24, 32, 48, 64, 86, 96, 112, 128, 160, 224):
        fn = lambda: ops.gemm(dy, x, dW, N, K, M, 1, N, 1, K, K, acc=ACC_ATOMIC, splitk=sk, compute=BF16)
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
        line.append(f"{sk}:{us:.0f}us[{lib.a3t_gemm_last_kernel().decode()[-16:]}]")
    print(f"N={N} K={K} tiles={tiles} default splitk={ops._splitk_for(tiles, M)}  " + "  ".join(line))
