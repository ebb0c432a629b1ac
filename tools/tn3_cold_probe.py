"""One FFN weight gradient on the 128x384-tile kernel, warm (one operand set, resident in the Infinity Cache) or cold (COLD=1: four
operand sets in rotation, 550 MB > 256 MiB): for rocprofv3 --pmc passes (tools/r05_tn3_pmc.sh)."""
import os, sys, torch
from a3t_amd import _lib, ops
from a3t_amd._lib import BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
lib.a3t_gemm_tn3_mode(1)
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
B, T = 32, 1120
M = B * T
cold = os.environ.get("COLD", "0") == "1"
cin, cout, taps = (384, 1536, 3) if os.environ.get("SHAPE", "A") == "A" else (1536, 384, 3)
sets = [(rn(M, cout).bfloat16(), rn(M, cin).bfloat16()) for _ in range(4)]
dW = torch.zeros(cout, taps, cin, device=DEV)
for i in range(int(os.environ.get("N", "24"))):
    dy, x = sets[i % 4 if cold else 0]
    ops.conv_bwd_weight(dy, x, dW, T, 1, compute=BF16)
torch.cuda.synchronize()
print(lib.a3t_gemm_last_kernel().decode(), "cold" if cold else "warm")
