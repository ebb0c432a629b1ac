"""The d_model-wide Linears (35 840 x 384 x K) alone on the panel kernel and on the 128-row kernel, with the epilogues the step uses
(fp32 output + fp32 residual + dropout: linear_out / pointwise_conv2 forward; bf16 output: their data gradients), over ROTATING
operand / output buffers (a loop over one buffer keeps it in the 256-MB Infinity Cache and flatters the write-heavy variants)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import BF16

lib = _lib.load()
M, N = 35840, 384
NB = 12
g = torch.Generator(device="cuda").manual_seed(0)
for K in (384, 768, 1152):
    xs = [torch.randn(M, K, device="cuda", generator=g).bfloat16() for _ in range(NB)]
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    Rs = [torch.randn(M, N, device="cuda", generator=g) for _ in range(NB)]
    o32 = [torch.empty(M, N, device="cuda") for _ in range(NB)]
    o16 = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    for mode in (1, 0):
        old = lib.a3t_gemm_pn_mode(mode)
        for name, fn in (("fp32 out + residual + dropout", lambda i: ops.gemm(xs[i], W, o32[i], M, N, K, K, 1, K, 1, N, bias=b, R=Rs[i], compute=BF16, drop=(0.1, 7))),
                         ("bf16 out", lambda i: ops.gemm(xs[i], W, o16[i], M, N, K, K, 1, K, 1, N, compute=BF16))):
            for i in range(NB):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(3):
                for i in range(NB):
                    fn(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (3 * NB) * 1e3
            print(f"K={K:5d} pn_mode {mode}  {name:32s} {us:6.1f} us   [{lib.a3t_gemm_last_kernel().decode()}]")
        lib.a3t_gemm_pn_mode(old)
    del xs, Rs, o32, o16
