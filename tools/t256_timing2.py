"""Per-workgroup phase cycles of the 256x256 GEMM kernel for ONE launch (build with -DT256_TIMING)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16, ACT_RELU
dev = "cuda"
lib = _lib.load()
lib.a3t_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
M, N, K = 35840, 1536, 1152
conv = len(sys.argv) > 1 and sys.argv[1] == "conv"
if conv:
    x = torch.randn(M, 384, device=dev).bfloat16(); W = (torch.randn(N, 3, 384, device=dev) * 0.03).bfloat16()
else:
    x = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
torch.cuda.synchronize()
def run():
    if conv: ops.conv_fwd(x, W, out, 1120, 1, bias=bias, act=ACT_RELU, compute=BF16, drop=(0.2, 7))
    else: ops.linear_fwd(x, W, out, bias=bias, act=ACT_RELU, compute=BF16, drop=(0.2, 7))
for rep in range(2):
    run()
    torch.cuda.synchronize()
    nb = ((M + 255) // 256) * ((N + 255) // 256)
    buf = np.zeros((8192, 8), dtype=np.uint64)
    lib.a3t_debug_read(buf.ctypes.data, buf.nbytes)
    b = buf[:nb].astype(np.int64)
    t0 = b[:, 0].min()
    d = b[:, :5] - t0
    seg = np.diff(d, axis=1)
    order = np.argsort(d[:, 0])
    first = order[:256]; later = order[256:]
    print(f"rep {rep} {'conv' if conv else 'plain'}: {nb} blocks, span {d[:,4].max()} cycles")
    for name, idx in (("first round", first), ("later rounds", later)):
        s = seg[idx]
        print(f"  {name:12s}: setup {s[:,0].mean():7.0f}  prologue {s[:,1].mean():7.0f}  main {s[:,2].mean():7.0f}  epilogue {s[:,3].mean():7.0f}  total {s.sum(1).mean():7.0f}")
