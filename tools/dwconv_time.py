"""GLU + depthwise conv forward / backward (conv module of a conformer block) alone at the configs[1] shape, kernel sizes 7 and 31."""
import torch
from a3t_amd import _lib, ops
DEV = torch.device("cuda:0")
g_ = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g_)
B, T, C = 32, 1120, 384
M = B * T

def timeit(fn, n=40):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for K in (7, 31):
    g = rn(M, 2 * C).bfloat16(); wdw = rn(C, K) * 0.2; bdw = rn(C) * 0.1
    glu = torch.empty(M, C, device=DEV, dtype=torch.bfloat16); z = torch.empty(M, C, device=DEV)
    dz = rn(M, C); dg = torch.empty(M, 2 * C, device=DEV, dtype=torch.bfloat16)
    dw = torch.zeros(C, K, device=DEV); db = torch.zeros(C, device=DEV); dgs = torch.zeros(2 * C, device=DEV)
    tf = timeit(lambda: ops.glu_dwconv_fwd(g, wdw, bdw, glu, z, T))
    tb = timeit(lambda: ops.glu_dwconv_bwd(dz, g, glu, wdw, dg, dw, db, T, dgs))
    torch.cuda.synchronize()
    print(f"K={K}: forward {tf:.1f} us, backward {tb:.1f} us   (checksums z {float(z.double().abs().sum()):.6e} dg {float(dg.double().abs().sum()):.6e} dw {float(dw.double().abs().sum()):.6e})")
