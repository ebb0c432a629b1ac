"""Data gradient of the second FFN conv: strided-W (NN) vs transposed-shadow (NT) on the 128x128 kernel vs the 8-phase kernel."""
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import ACT_RELU, BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
B, T, d, ff = 32, 1120, 384, 1536
M = B * T
h = torch.relu(rn(M, ff)).bfloat16()
ga, W2 = rn(M, d).bfloat16(), rn(d, 3, ff, sc=0.02).bfloat16()
W2t = W2.permute(2, 1, 0).flip(1).contiguous()
dh = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
gb = torch.zeros(ff, device=DEV)
keep = torch.zeros(ops.gemm_keep_bytes(M, ff), dtype=torch.uint8, device=DEV)
def timeit(fn, n=100):
    for _ in range(400): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def nt_S():
    ops.gemm(ga, W2t, dh, M, ff, 3 * d, d, 1, 3 * d, 1, ff, b_ts=d, S=h, taps=3, pad=1, Tseq=T, alpha=0.6, compute=BF16, colsum=gb)
lib.a3t_gemm_8p_mode(0)
t_nn = timeit(lambda: ops.conv_bwd_data(ga, W2, dh, T, 1, S=h, alpha=0.6, compute=BF16, colsum=gb))
k_nn = lib.a3t_gemm_last_kernel().decode()
t_nt = timeit(nt_S)
k_nt = lib.a3t_gemm_last_kernel().decode()
lib.a3t_gemm_8p_mode(1)
t_8p = timeit(lambda: ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.6, compute=BF16, keep_in=keep, colsum=gb))
print(f"NN strided W + S: {t_nn:.1f} us ({k_nn}) | NT transposed W + S: {t_nt:.1f} us ({k_nt}) | 8p NT + keep bits: {t_8p:.1f} us")
