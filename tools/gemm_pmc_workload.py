"""GEMM launches for the SQ-counter passes of tools/gemm_pmc.sh: the conv-FFN classes of BASELINE configs[1] on the
128x128 kernel (forward / data gradient / weight gradient) and on the 384-column panel kernel (second conv forward, both data
gradients through the transposed weights), and of configs[3] on the 8-phase kernel (forward convs with their epilogues, data
gradient with the keep-bit mask)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16, ACT_RELU
dev = "cuda"
lib = _lib.load()
rn = lambda *s, sc=1.0: torch.randn(*s, device=dev) * sc
# ---- configs[1]
B, T, Cin, Cout = 32, 1120, 384, 1536
M = B * T
x, Wk, bias = rn(M, Cin).bfloat16(), rn(Cout, 3, Cin, sc=0.03).bfloat16(), rn(Cout)
h = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
dy, W2, g = rn(M, Cout).bfloat16(), rn(Cin, 3, Cout, sc=0.02).bfloat16(), rn(M, Cin).bfloat16()
dx = torch.empty(M, Cin, device=dev, dtype=torch.bfloat16)
dh = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
dW = torch.zeros(Cout, 3, Cin, device=dev)
W1t, W2t = Wk.permute(2, 1, 0).flip(1).contiguous(), W2.permute(2, 1, 0).flip(1).contiguous()
b2, xres, xo, gb = rn(Cin), rn(M, Cin), torch.empty(M, Cin, device=dev), torch.zeros(Cout, device=dev)
for _ in range(3):
    ops.conv_fwd(h, W2, xo, T, 1, bias=b2, R=xres, alpha=0.5, compute=BF16, drop=(0.2, 5))      # panel kernel: conv 2 forward
    ops.conv_fwd(dy, W1t, dx, T, 1, compute=BF16)                                                # ... data gradient of conv 1
    ops.conv_fwd(g, W2t, dh, T, 1, S=h, alpha=0.5, compute=BF16, colsum=gb)                      # ... data gradient of conv 2
    ops.conv_fwd(x, Wk, h, T, 1, bias=bias, act=ACT_RELU, compute=BF16, drop=(0.2, 7))
    ops.conv_bwd_data(g, W2, dh, T, 1, S=h, alpha=0.5, compute=BF16)
    ops.conv_bwd_data(dy, Wk, dx, T, 1, compute=BF16)
    ops.conv_bwd_weight(dy, x, dW, T, 1, compute=BF16)
torch.cuda.synchronize()
del x, Wk, h, dy, W2, g, dx, dh, dW, W1t, W2t, xres, xo
# ---- configs[3]: M = 16 * 1800, d = 512, ff = 2048 -> the dispatcher's cost model picks the 8-phase kernel
B, T, d, ff = 16, 1800, 512, 2048
M = B * T
y, xres = rn(M, d).bfloat16(), rn(M, d)
W1, W2 = rn(ff, 3, d, sc=0.03).bfloat16(), rn(d, 3, ff, sc=0.02).bfloat16()
W2t = W2.permute(2, 1, 0).flip(1).contiguous()
b1, b2 = rn(ff), rn(d)
h = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
keep = torch.zeros(ops.gemm_keep_bytes(M, ff), dtype=torch.uint8, device=dev)
xo = torch.empty(M, d, device=dev)
ga = rn(M, d).bfloat16()
dh = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
gb = torch.zeros(ff, device=dev)
for _ in range(3):
    ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 7), keep_out=keep)
    ops.conv_fwd(h, W2, xo, T, 1, bias=b2, R=xres, alpha=0.5, compute=BF16, drop=(0.2, 5))
    ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.6, compute=BF16, keep_in=keep, colsum=gb)
torch.cuda.synchronize()
print(lib.a3t_gemm_last_kernel().decode())
