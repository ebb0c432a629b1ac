"""Where a K-tile of the 128 x 384 token-reduction kernel spends its cycles: reads the segment timers of an instrumented build
(hipcc -DG8_TIMING [-DG8_TIMING_D] on gemm_bf16_8p.hip, linked to tools/ab/liba3t_hip_tn3_timing[_d].so; A3T_LIB_PATH selects it).
Slots per phase p = 0..2: 4p+0 load segment up to the first barrier (only with G8_TIMING_D: the stamp forces the fragment reads
home), 4p+1 first barrier + lgkmcnt(0) (without _D: the whole load segment too), 4p+2 the 16 MFMAs' issue, 4p+3 second barrier."""
import ctypes, os, sys
import numpy as np
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
lib.a3t_gemm_tn3_mode(1)
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
B, T = 32, 1120
M = B * T
cin, cout, taps = 384, 1536, 3
sets = [(rn(M, cout).bfloat16(), rn(M, cin).bfloat16()) for _ in range(4)]
dW = torch.zeros(cout, taps, cin, device=DEV)
for i in range(12):
    dy, x = sets[i % 4]
    ops.conv_bwd_weight(dy, x, dW, T, 1, compute=BF16)
torch.cuda.synchronize()
buf = np.zeros(256 * 2 * 16, np.uint64)
lib.a3t_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.a3t_debug_read(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
st = buf.reshape(256, 2, 16)[:252].astype(np.float64) / 80.0       # cycles per K-tile (80 K-tiles per workgroup)
names = ["load", "bar1+lgkm", "mfma", "bar2"]
print(lib.a3t_gemm_last_kernel().decode(), "cycles per K-tile, mean over 252 workgroups (min .. max)")
for grp in (0, 1):
    tot = st[:, grp, :12].sum(1).mean()
    print(f" wave group {grp}: total {tot:.0f} cycles per K-tile")
    for ph in range(3):
        sub = {0: (("reads", 12), ("dma", 13)), 1: (("dma+cursor", 14),), 2: (("dma", 15),)}[ph]
        if False:
            print("            load segment split: " + "  ".join(f"{n} {st[:, grp, k].mean():.0f}" for n, k in sub) + f"  vmcnt wait {st[:, grp, 4 * ph].mean():.0f}")
        print("   phase %d: " % (ph + 1) + "  ".join(f"{names[k]} {st[:, grp, 4 * ph + k].mean():6.0f} ({st[:, grp, 4 * ph + k].min():.0f}..{st[:, grp, 4 * ph + k].max():.0f})" for k in range(4)))
