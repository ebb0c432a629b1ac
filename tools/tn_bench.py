import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16, ACC_ATOMIC
dev="cuda"
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
M, N, Cin, taps, T = 35840, 1536, 384, 3, 1120
dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, Cin, device=dev).bfloat16()
dW = torch.zeros(N, taps, Cin, device=dev)
for sk in (2, 4, 6, 9, 12, 18):
    t = timeit(lambda: ops.gemm(dy, x, dW, N, taps*Cin, M, 1, N, 1, Cin, taps*Cin, taps=taps, pad=1, dil=1, Tseq=T, acc=ACC_ATOMIC, splitk=sk, compute=BF16))
    print(f"TN wgrad splitk={sk:2d} blocks={12*9*sk:5d}: {t*1e6:8.1f} us {2.0*M*N*Cin*taps/t/1e12:7.1f} TF")
