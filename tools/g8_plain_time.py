import torch, os
from a3t_amd import _lib, ops
from a3t_amd._lib import BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
M, N, K = 35840, 1536, 1152
import sys
mode = sys.argv[1] if len(sys.argv) > 1 else "randn"
if mode == "uniform":
    x = (torch.rand(M, K, device=DEV, generator=g) * 2 - 1).bfloat16()
    W = (torch.rand(N, K, device=DEV, generator=g) * 2 - 1).bfloat16()
elif mode == "zero":
    x, W = torch.zeros(M, K, device=DEV).bfloat16(), torch.zeros(N, K, device=DEV).bfloat16()
else:
    x, W = rn(M, K).bfloat16(), rn(N, K, sc=0.03).bfloat16()
h = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
lib.a3t_gemm_8p_mode(1)
f = lambda: ops.linear_fwd(x, W, h, compute=BF16)
for _ in range(1500): f()   # ~0.25 s: the clocks must have ramped up before anything is timed
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 100 * 1e3
print(mode, os.environ.get("A3T_LIB_PATH", "default").split("/")[-1], f"plain linear {t:.1f} us  {2.0*M*N*K/t/1e6:.0f} TF", lib.a3t_gemm_last_kernel().decode())
