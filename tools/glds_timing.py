"""Where a K-step of the single-buffer direct-to-LDS GEMM spends its time (needs a build with -DGLDS_TIMING:
A3T_EXTRA_FLAGS=-DGLDS_TIMING python a3t_amd/build.py --force).  The tick unit of the cycle counter is not calibrated:
read the PERCENTAGES (measured: DMA issue 39 %, DMA wait 18 %, barriers 8 %, ds_read + MFMA issue 35 % of a K-step)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16
dev = "cuda"
lib = _lib.load()
rd = lib.a3t_debug_read_glds
rd.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
B, T = 32, 1120
M = B * T


def report(tag, nblocks):
    torch.cuda.synchronize()
    buf = np.zeros((16384, 8), dtype=np.uint64)
    rd(buf.ctypes.data, buf.nbytes)
    b = buf[:min(nblocks, 16384)].astype(np.float64)
    steps = b[:, 7]
    tot = (b[:, 6] - b[:, 5])
    names = ["WAR barrier", "DMA issue", "DMA wait (vmcnt)", "landed barrier", "ds_read+MFMA issue"]
    print(f"== {tag}: {nblocks} workgroups, {steps.mean():.1f} K-steps each; K loop {tot.mean()/100:.1f} us per workgroup "
          f"= {tot.mean()/steps.mean()/100:.2f} us per K-step")
    for i, n in enumerate(names):
        print(f"   {n:22s} {b[:, i].sum()/steps.sum()*10:8.1f} ns per K-step  ({b[:, i].sum()/tot.sum()*100:4.1f} %)")
    span = (b[:, 6].max() - b[:, 5].min()) / 100
    print(f"   first start -> last K-loop end: {span:.1f} us")


for (cin, cout) in [(384, 1536), (1536, 384)]:
    x = torch.randn(M, cin, device=dev).bfloat16()
    Wk = (torch.randn(cout, 3, cin, device=dev) * 0.03).bfloat16()
    out = torch.empty(M, cout, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.conv_fwd(x, Wk, out, T, 1, compute=BF16)
    report(f"conv fwd {cin}->{cout} k3 (NT)", ((M + 127) // 128) * ((cout + 127) // 128))
    dy = torch.randn(M, cout, device=dev).bfloat16()
    dx = torch.empty(M, cin, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.conv_bwd_data(dy, Wk, dx, T, 1, compute=BF16)
    report(f"conv dgrad {cout}->{cin} k3 (NN)", ((M + 127) // 128) * ((cin + 127) // 128))
