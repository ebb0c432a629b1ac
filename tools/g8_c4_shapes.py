"""FFN GEMM classes of BASELINE configs[3] (d=512, ff=2048, B=16, T=1800) and configs[1]: 128x128 kernel vs 8-phase kernel."""
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import ACT_RELU, BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
def timeit(fn, n=50):
    for _ in range(300): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, T, d, ff) in [(16, 1800, 512, 2048), (32, 1120, 384, 1536)]:
    M = B * T
    y, x = rn(M, d).bfloat16(), rn(M, d)
    W1, W2 = rn(ff, 3, d, sc=0.03).bfloat16(), rn(d, 3, ff, sc=0.02).bfloat16()
    b1, b2 = rn(ff), rn(d)
    h = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
    hh = torch.relu(rn(M, ff)).bfloat16()
    xo = torch.empty(M, d, device=DEV)
    fl = 2.0 * M * ff * 3 * d
    for mode in (0, 1):
        lib.a3t_gemm_8p_mode(mode)
        t1 = timeit(lambda: ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 777)))
        k1 = lib.a3t_gemm_last_kernel().decode()
        t2 = timeit(lambda: ops.conv_fwd(hh, W2, xo, T, 1, bias=b2, R=x, alpha=0.5, compute=BF16, drop=(0.2, 5)))
        k2 = lib.a3t_gemm_last_kernel().decode()
        print(f"M={M} d={d} ff={ff} mode {mode}: conv1 fwd {t1:.1f} us ({fl/t1/1e6:.0f} TF) [{k1}] | conv2 fwd (+R, fp32) {t2:.1f} us ({fl/t2/1e6:.0f} TF) [{k2}]")
    lib.a3t_gemm_8p_mode(2)
    # correctness of the R path
    lib.a3t_gemm_8p_mode(0); ops.conv_fwd(hh, W2, xo, T, 1, bias=b2, R=x, alpha=0.5, compute=BF16, drop=(0.2, 5)); a = xo.clone()
    lib.a3t_gemm_8p_mode(1); ops.conv_fwd(hh, W2, xo, T, 1, bias=b2, R=x, alpha=0.5, compute=BF16, drop=(0.2, 5)); lib.a3t_gemm_8p_mode(2)
    torch.cuda.synchronize()
    print("   R path max diff vs 128^2:", float((a - xo).abs().max()), lib.a3t_gemm_last_kernel().decode())
