"""A3T_LIB_PATH=a3t_amd/lib/liba3t_hip_timing.so python tools/g8_timing.py  -- per-tile stamps (10 ns ticks) of the 8-phase GEMM"""
import ctypes
import numpy as np
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import ACT_RELU, BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
lib.a3t_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
B, T, d, ff = 32, 1120, 384, 1536
M = B * T
y, W1, b1 = rn(M, d).bfloat16(), rn(ff, 3, d, sc=0.03).bfloat16(), rn(ff)
h = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
keep = torch.zeros(ops.gemm_keep_bytes(M, ff), dtype=torch.uint8, device=DEV)
gb = torch.zeros(ff, device=DEV)
lib.a3t_gemm_8p_mode(1)
cases = {
    "plain": lambda: ops.conv_fwd(y, W1, h, T, 1, compute=BF16),
    "bias+relu": lambda: ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16),
    "bias+relu+dropout+keep_out": lambda: ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 7), keep_out=keep),
    "keep_in+colsum": lambda: ops.conv_fwd(y, W1, h, T, 1, alpha=0.6, compute=BF16, keep_in=keep, colsum=gb),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = np.zeros(256 * 2 * 16, dtype=np.uint64)
    lib.a3t_debug_read(st.ctypes.data, st.nbytes)
    st = st.reshape(256, 2, 16).astype(np.int64)
    print(name)
    for blk in (0, 100, 255):
        for grp in (0, 1):
            q = st[blk, grp]
            print(f"   wg {blk:3d} group {grp}:", " ".join(str(int(v - q[0])) for v in q[1:8] if v))
