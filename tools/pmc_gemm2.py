import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16
dev = "cuda"
B, T, Cin, Cout = 32, 1120, 384, 1536
M = B * T
x = torch.randn(M, Cin, device=dev).bfloat16(); Wk = (torch.randn(Cout, 3, Cin, device=dev) * 0.03).bfloat16()
xp = torch.randn(M, 3 * Cin, device=dev).bfloat16()
h = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv_fwd(x, Wk, h, T, 1, compute=BF16)                     # conv (dispatch 1..3)
for _ in range(3):
    ops.linear_fwd(xp, Wk.view(Cout, -1), h, compute=BF16)         # plain, same M,N,K
torch.cuda.synchronize()
