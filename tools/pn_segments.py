"""Where a K-tile of the panel GEMM spends its cycles: reads the segment timers of an instrumented build (hipcc -DPN_TIMING on
gemm_bf16_pn.hip, linked as a3t_amd/lib/liba3t_hip_pn_timing.so; A3T_LIB_PATH selects it).  Per phase p = 0..3 of a K-tile:
3p+0 = fragment reads + DMA issue + counted vmcnt wait + first barrier + lgkmcnt(0), 3p+1 = the 15 MFMAs' issue, 3p+2 = second
barrier.  usage: A3T_LIB_PATH=.../liba3t_hip_pn_timing.so python tools/pn_segments.py [cin cout]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
lib.a3t_gemm_pn_mode(1)
lib.a3t_gemm_8p_mode(0)
cin, cout = (int(x) for x in (sys.argv[1:3] if len(sys.argv) >= 3 else (1536, 384)))
g = torch.Generator(device=DEV).manual_seed(0)
B, T, taps = 32, 1120, 3
M = B * T
xs = [torch.randn(M, cin, device=DEV, generator=g).bfloat16() for _ in range(3)]
W = (torch.randn(cout, taps, cin, device=DEV, generator=g) * 0.03).bfloat16()
o = torch.empty(M, cout, device=DEV, dtype=torch.bfloat16)
for i in range(9):
    ops.conv_fwd(xs[i % 3], W, o, T, 1, compute=BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(9):
    ops.conv_fwd(xs[i % 3], W, o, T, 1, compute=BF16)
e1.record()
torch.cuda.synchronize()
print(lib.a3t_gemm_last_kernel().decode(), f"{e0.elapsed_time(e1) / 9 * 1e3:.1f} us per launch (instrumented build)")
buf = np.zeros(256 * 2 * 16, np.uint64)
lib.a3t_debug_read_pn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.a3t_debug_read_pn(buf.ctypes.data, buf.nbytes) == 0
nk = taps * cin // 64
st = buf.reshape(256, 2, 16)[:224].astype(np.float64) / nk
names = ["reads + dma + vmcnt + bar1 + lgkm", "15 mfma", "bar2"]
for grp in (0, 1):
    print(f" wave group {grp}: {st[:, grp, :12].sum(1).mean():.0f} cycles per K-tile (mean over 224 workgroups)")
    for ph in range(4):
        print("   phase %d: " % (ph + 1) + "  ".join(f"{names[k]} {st[:, grp, 3 * ph + k].mean():5.0f} ({st[:, grp, 3 * ph + k].min():.0f}..{st[:, grp, 3 * ph + k].max():.0f})" for k in range(3)))
