"""Per-workgroup wall-clock stamps of the streaming attention-backward GEMM (probe build: A3T_EXTRA_FLAGS=-DTT_TIMING or a library
built with it, A3T_LIB_PATH): start -> first K-tile landed -> K loop done -> epilogue done, in us (100 MHz counter)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16

lib = _lib.load()
B, H, T, dk = 32, 2, 1120, 192
d, M = H * dk, B * T
S = (torch.randn(B, H, T, T, device="cuda") * 0.1).bfloat16()
x = torch.randn(M, d, device="cuda").bfloat16()
out = torch.zeros(M, 3 * d, device="cuda").bfloat16()
zb = (H * T * T, T * T)
lib.a3t_gemm_tt_mode(1)
for variant in ("NN", "TN"):
    def run():
        if variant == "NN":
            ops.gemm(S, x, out, T, dk, T, T, 1, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=BF16)
        else:
            ops.gemm(S, x, out, T, dk, T, 1, T, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=BF16)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 4, dtype=np.uint64)
    lib.a3t_debug_read_tt.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.a3t_debug_read_tt(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(1024, 4)[:256].astype(np.float64) / 100.0
    t0 = st[:, 0].min()
    seg = np.stack([st[:, 0] - t0, st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]], 1)
    print(variant, lib.a3t_gemm_last_kernel().decode(), "start spread %.1f us | first tile %.1f (max %.1f) | K loop %.1f (max %.1f) | epilogue %.1f (max %.1f) | last end %.1f us"
          % (seg[:, 0].max(), seg[:, 1].mean(), seg[:, 1].max(), seg[:, 2].mean(), seg[:, 2].max(), seg[:, 3].mean(), seg[:, 3].max(), (st[:, 3] - t0).max()))
