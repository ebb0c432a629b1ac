"""Linear-layer weight gradients (few output tiles): split-K / buffering sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16, ACC_ATOMIC
lib = _lib.load()
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
for (N, K) in [(384, 384), (1152, 384), (768, 384), (384, 768)]:
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    for sk in sorted(set([ops._splitk_for(tiles, M), 440 // tiles, 640 // tiles, 800 // tiles, 1000 // tiles, 1300 // tiles])):
        dW = torch.zeros(N, K, device=dev)
        t = timeit(lambda: ops.gemm(dy, x, dW, N, K, M, 1, N, 1, K, K, acc=ACC_ATOMIC, splitk=sk, compute=BF16))
        print(f"TN {N}x{K} tiles {tiles:3d} splitk={sk:3d} ({tiles*sk:4d} wgs): {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF  {lib.a3t_gemm_last_kernel().decode()}"
              + ("   <- default" if sk == ops._splitk_for(tiles, M) else ""), flush=True)
