"""Per-workgroup phase timestamps of the 256x256 GEMM kernel (needs a build with -DT256_TIMING:
A3T_EXTRA_FLAGS=-DT256_TIMING python a3t_amd/build.py --force)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16
dev = "cuda"
lib = _lib.load()
lib.a3t_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for (M, N, K) in [(35840, 1536, 1152), (8192, 8192, 8192)]:
    x = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear_fwd(x, W, out, compute=BF16)
    torch.cuda.synchronize()
    nb = ((M + 255) // 256) * ((N + 255) // 256)
    buf = np.zeros((8192, 8), dtype=np.uint64)
    lib.a3t_debug_read(buf.ctypes.data, buf.nbytes)
    b = buf[:min(nb, 8192)].astype(np.int64)
    t0 = b[:, 0].min()
    d = b[:, :5] - t0
    print(f"== {M}x{N}x{K}: {nb} blocks; clock ticks (s_memtime @100MHz) -> us = ticks/100")
    seg = np.diff(d, axis=1)
    print(" mean segment ticks: setup %.0f  prologue %.0f  main %.0f  epilogue %.0f" % tuple(seg.mean(0)))
    print(" start times (sorted, every 64th):", np.sort(d[:, 0])[::64][:20])
    print(" end times   (sorted, every 64th):", np.sort(d[:, 4])[::64][:20])
    print(" total span ticks:", d[:, 4].max())
