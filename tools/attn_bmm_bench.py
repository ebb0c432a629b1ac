"""Attention batched GEMMs of the C2 step (B=32, H=2, T=1120, d_k=192) with the engine's strided views, timed alone.
Run twice to compare tile variants:  A3T_GEMM_WN3=0 python tools/attn_bmm_bench.py ; python tools/attn_bmm_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16
lib = _lib.load()
dev = "cuda"
B, H, T, dk = 32, 2, 1120, 192
d = H * dk
bf = torch.bfloat16
probs = (torch.randn(B, H, T, T, device=dev) * T ** -0.5).to(bf)
qkv = torch.randn(B * T, 3 * d, device=dev).to(bf)
qu = torch.randn(B * T, d, device=dev).to(bf)
P = torch.randn(T, d, device=dev).to(bf)
out_d = torch.empty(B * T, d, device=dev, dtype=bf)
out_qkv = torch.empty(B * T, 3 * d, device=dev, dtype=bf)
sc = torch.empty(B, H, T, T, device=dev, dtype=bf)
cs = torch.zeros(3 * d, device=dev)
zb = (H * T * T, T * T)
kk, vv = qkv.view(-1)[d:], qkv.view(-1)[2 * d:]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


cases = {
    "NT scores (q+u) k^T      ": lambda: ops.gemm(qu, kk, sc, T, T, dk, d, 1, 3 * d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk), b_bs=(T * 3 * d, dk), c_bs=zb, compute=BF16),
    "NT bd (q+v) P^T          ": lambda: ops.gemm(qu, P, sc, T, T, dk, d, 1, d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk), b_bs=(0, dk), c_bs=zb, compute=BF16),
    "NN ctx = probs V         ": lambda: ops.gemm(probs, vv, out_d, T, dk, T, T, 1, 1, 3 * d, d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * 3 * d, dk), c_bs=(T * d, dk), compute=BF16),
    "NN dqu = ds K (+colsum)  ": lambda: ops.gemm(probs, kk, out_d, T, dk, T, T, 1, 1, 3 * d, d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * 3 * d, dk), c_bs=(T * d, dk), compute=BF16, colsum=cs, colsum_bs1=dk),
    "NN dqv = dbd P (+colsum) ": lambda: ops.gemm(probs, P, out_d, T, dk, T, T, 1, 1, d, d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(0, dk), c_bs=(T * d, dk), compute=BF16, colsum=cs, colsum_bs1=dk),
    "TN dV = probs^T dctx (+cs)": lambda: ops.gemm(probs, qu, out_qkv.view(-1)[2 * d:], T, dk, T, 1, T, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, colsum=cs[2 * d:], colsum_bs1=dk),
}
dP = torch.zeros(T, d, device=dev)
cases["TN dP += dbd^T (q+v), atomic"] = lambda: ops.gemm(probs, qu, dP, T, dk, T, 1, T, 1, d, d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * d, dk), c_bs=(0, dk), acc=_lib.ACC_ATOMIC, compute=BF16)
sl = torch.zeros(16, 4 * d, device=dev)
sk = dict(colsum_bs1=dk, colsum_slots=16, colsum_ss=4 * d)
cases["NN dqu, 16 colsum slots  "] = lambda: ops.gemm(probs, kk, out_d, T, dk, T, T, 1, 1, 3 * d, d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * 3 * d, dk), c_bs=(T * d, dk), compute=BF16, colsum=sl[0], **sk)
cases["TN dV, 16 colsum slots   "] = lambda: ops.gemm(probs, qu, out_qkv.view(-1)[2 * d:], T, dk, T, 1, T, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, colsum=sl[0, 3 * d:], **sk)
cases["fold                     "] = lambda: ops.attn_bias_fold(sl, 16, d, cs[:d], cs[d:2 * d], cs)
for name, fn in cases.items():
    t = timeit(fn)
    print(f"{name}: {t*1e6:7.1f} us  {2.0*B*H*T*T*dk/t/1e12:6.1f} TF  {lib.a3t_gemm_last_kernel().decode()}", flush=True)
