"""Micro-benchmark of the LayerNorm kernels at the C2 shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
dev = "cuda"
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M, D = 35840, 384
x = torch.randn(M, D, device=dev); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
y = torch.empty(M, D, device=dev, dtype=torch.bfloat16); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
t = timeit(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, 1e-12))
print(f"ln fwd {t:6.1f} us  {(M*D*6)/t/1e6:5.2f} TB/s")
dy = torch.randn(M, D, device=dev).bfloat16(); dres = torch.randn(M, D, device=dev)
dx = torch.empty(M, D, device=dev); dx16 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); dxs = torch.zeros(D, device=dev)
t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres, dx, dg, db, dx16=dx16, dxsum=dxs, dxsum_scale=0.5))
print(f"ln bwd {t:6.1f} us  {(M*D*16)/t/1e6:5.2f} TB/s")
