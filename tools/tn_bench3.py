"""Weight-gradient (TN, split-K atomics) GEMM sweep; run with A3T_GEMM_8P_TN=0/1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16, ACC_ATOMIC
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
M = 35840
for (N, K) in [(1536, 1152), (384, 4608), (1536, 384)]:
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    ref = dy.float().t() @ x.float()
    for sk in (4, 7, 8, 10, 14, 16, 20, 28):
        dW = torch.zeros(N, K, device=dev)
        ops.gemm(dy, x, dW, N, K, M, 1, N, 1, K, K, acc=ACC_ATOMIC, splitk=sk, compute=BF16)
        err = float((dW - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: ops.gemm(dy, x, dW, N, K, M, 1, N, 1, K, K, acc=ACC_ATOMIC, splitk=sk, compute=BF16))
        print(f"TN {N}x{K} splitk={sk:3d}: {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF relerr {err:.1e}")
