import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16, ACT_RELU
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 35840
for (N, K) in [(1536, 1152), (384, 4608), (1536, 4608), (1152, 384)]:
    x = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    for odt in (torch.bfloat16, torch.float32):
        out = torch.empty(M, N, device=dev, dtype=odt)
        t = timeit(lambda: ops.linear_fwd(x, W, out, compute=BF16))
        k = _lib.load().a3t_gemm_last_kernel().decode()
        t2 = timeit(lambda: ops.linear_fwd(x, W, out, bias=bias, act=ACT_RELU, compute=BF16, drop=(0.2, 5)))
        print(f"NT {M}x{N}x{K} out={str(odt)[6:]:9s}: plain {t:7.1f} us {2.0*M*N*K/t/1e6:7.1f} TF | bias+relu+drop {t2:7.1f} us  [{k}]")
