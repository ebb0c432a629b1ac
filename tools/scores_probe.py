"""Where the time of the attention-score GEMMs (M = N = 1120, K = d_k = 192, batch 64, bf16 out) goes: K sweep isolates the
epilogue + launch floor from the K loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16
dev = "cuda"
B, H, T = 32, 2, 1120
bf = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for odt in (bf, torch.float32):
    for dk in (64, 128, 192, 384, 768):
        d = H * dk
        q = torch.randn(B * T, d, device=dev).to(bf)
        k = torch.randn(B * T, d, device=dev).to(bf)
        sc = torch.empty(B, H, T, T, device=dev, dtype=odt)
        t = timeit(lambda: ops.gemm(q, k, sc, T, T, dk, d, 1, d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk),
                                    b_bs=(T * d, dk), c_bs=(H * T * T, T * T), compute=BF16))
        print(f"out {str(odt)[6:]:8s} K={dk:4d}: {t:7.1f} us  {2.0*B*H*T*T*dk/t/1e6:6.1f} TF  write {sc.numel()*sc.element_size()/t/1e6:5.2f} TB/s", flush=True)
# memset yardstick: how fast can 160 MB be written at all
sc = torch.empty(B, H, T, T, device=dev, dtype=bf)
t = timeit(lambda: sc.zero_())
print(f"memset 160 MB: {t:.1f} us  {sc.numel()*2/t/1e6:.2f} TB/s")
