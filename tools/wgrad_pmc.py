import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16
dev = "cuda"
B, T, d, ff = 32, 1120, 384, 1536
M = B * T
dh = torch.randn(M, ff, device=dev).bfloat16(); y = torch.randn(M, d, device=dev).bfloat16()
ga = torch.randn(M, d, device=dev).bfloat16(); h = torch.randn(M, ff, device=dev).bfloat16()
gW1, gW2 = torch.zeros(ff, 3, d, device=dev), torch.zeros(d, 3, ff, device=dev)
for _ in range(3):
    ops.conv_bwd_weight(dh, y, gW1, T, 1, compute=BF16)
    ops.conv_bwd_weight(ga, h, gW2, T, 1, alpha=0.5, compute=BF16)
torch.cuda.synchronize()
