import ctypes, numpy as np, torch
from a3t_amd import _lib, ops
from a3t_amd._lib import BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
lib.a3t_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
B, T = 32, 1120
M = B * T
lib.a3t_gemm_8p_mode(1)
for (cin, cout) in [(384, 1536), (1536, 384)]:
    dy, x = rn(M, cout).bfloat16(), rn(M, cin).bfloat16()
    dW = torch.zeros(cout, 3, cin, device=DEV)
    for _ in range(300): ops.conv_bwd_weight(dy, x, dW, T, 1, compute=BF16)
    torch.cuda.synchronize()
    st = np.zeros(256 * 2 * 16, dtype=np.uint64)
    lib.a3t_debug_read(st.ctypes.data, st.nbytes)
    st = st.reshape(256, 2, 16).astype(np.int64)
    print(cin, cout, lib.a3t_gemm_last_kernel().decode())
    for blk in (0, 1, 100, 239):
        q = st[blk, 0]
        print(f"   wg {blk:3d}: prologue {q[1]-q[0]} loop {q[2]-q[1]} epilogue {q[3]-q[2]}  (10 ns)")
