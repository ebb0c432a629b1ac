import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16, ACT_RELU
dev = "cuda"
B, T, Cin, Cout = 32, 1120, 384, 1536
M = B * T
x = torch.randn(M, Cin, device=dev).bfloat16(); Wk = (torch.randn(Cout, 3, Cin, device=dev) * 0.03).bfloat16()
bias = torch.randn(Cout, device=dev)
h = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv_fwd(x, Wk, h, T, 1, bias=bias, act=ACT_RELU, compute=BF16)
torch.cuda.synchronize()
dy = torch.randn(M, Cout, device=dev).bfloat16()
dx = torch.empty(M, Cin, device=dev, dtype=torch.bfloat16)
dh = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
W2 = (torch.randn(Cin, 3, Cout, device=dev) * 0.02).bfloat16()
g = torch.randn(M, Cin, device=dev).bfloat16()
dW = torch.zeros(Cout, 3, Cin, device=dev)
for _ in range(2):
    ops.conv_bwd_data(g, W2, dh, T, 1, S=h, alpha=0.5, compute=BF16)      # NN, N=1536 K=1152 (+relu mask)
    ops.conv_bwd_data(dy, Wk, dx, T, 1, compute=BF16)                     # NN, N=384 K=4608
    ops.conv_bwd_weight(dy, x, dW, T, 1, compute=BF16)                    # TN
torch.cuda.synchronize()
