"""Micro-benchmark of the GLU + depthwise-conv kernels at the C2 shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, T, C = 32, 1120, 384
M = B * T
for K in (7, 31):
    g2 = torch.randn(M, 2 * C, device=dev).bfloat16()
    w = torch.randn(C, K, device=dev) * 0.2
    b = torch.randn(C, device=dev)
    glu = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    z = torch.empty(M, C, device=dev)
    t = timeit(lambda: ops.glu_dwconv_fwd(g2, w, b, glu, z, T))
    print(f"K={K} fwd {t:7.1f} us")
    dz = torch.randn(M, C, device=dev)
    dg = torch.empty(M, 2 * C, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(C, K, device=dev); db = torch.zeros(C, device=dev); dgs = torch.zeros(2 * C, device=dev)
    t = timeit(lambda: ops.glu_dwconv_bwd(dz, g2, glu, w, dg, dw, db, T, dgsum=dgs))
    print(f"K={K} bwd {t:7.1f} us")
