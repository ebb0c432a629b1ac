"""The score-sized products of the attention backward on the streaming kernel (gemm_bf16_tt.hip) against the 128-row kernel:
results (max |difference| relative to the fp32 product's scale) and launch times.  tools/tt_gemm_check.py [B H T dk]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16

lib = _lib.load()
B, H, T, dk = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 2, 1120, 192)))
d, M = H * dk, B * T
dev = "cuda"
torch.manual_seed(0)
S = (torch.randn(B, H, T, T, device=dev) * 0.1).bfloat16()
qkv = torch.randn(M, 3 * d, device=dev).bfloat16()
x = torch.randn(M, d, device=dev).bfloat16()
P = torch.randn(T, d, device=dev).bfloat16()
zb = (H * T * T, T * T)
NS = int(os.environ.get("TT_SLOTS", "8"))
sl = torch.zeros(NS * 4 * d, device=dev)
csk = dict(colsum_bs1=dk, colsum_slots=NS, colsum_ss=4 * d)
NOCS = os.environ.get("TT_NOCS", "0") == "1"


def nn_qkv(out):      # dQu = dS K : A [m][k], B = k third of qkv, C = q third of out (bf16 store + column sums)
    ops.gemm(S, qkv.view(-1)[d:], out, T, dk, T, T, 1, 1, 3 * d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb,
             b_bs=(T * 3 * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, **({} if NOCS else dict(colsum=sl, **csk)))


def nn_pos_add(out):  # dQv = dBD P : B shared by the batch, C += (bf16, fp32 sum)
    ops.gemm(S, P, out, T, dk, T, T, 1, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(0, dk),
             c_bs=(T * 3 * d, dk), acc=ops.ACC_ADD, compute=BF16, **({} if NOCS else dict(colsum=sl[d:], **csk)))


def tn_x(out):        # dV / dK = S^T x : A [k][m], C = third of out
    ops.gemm(S, x, out.view(-1)[2 * d:], T, dk, T, 1, T, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb,
             b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, **({} if NOCS else dict(colsum=sl[3 * d:], **csk)))


def timeit(fn, out, reps=10):
    fn(out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn in (("dS K (NN, store)", nn_qkv), ("dBD P (NN, add)", nn_pos_add), ("S^T x (TN, store)", tn_x)):
    res = {}
    for mode in (0, 1):
        old = lib.a3t_gemm_tt_mode(mode)
        out = torch.zeros(M, 3 * d, device=dev).bfloat16() + (0.5 if "add" in name else 0.0)
        sl.zero_()
        fn(out)
        torch.cuda.synchronize()
        kern = lib.a3t_gemm_last_kernel().decode()
        res[mode] = (out.float().clone(), sl.clone(), kern, timeit(fn, torch.zeros(M, 3 * d, device=dev).bfloat16()))
        lib.a3t_gemm_tt_mode(old)
    (o0, s0, k0, t0), (o1, s1, k1, t1) = res[0], res[1]
    scale = o0.abs().max().item()
    s4 = s0.view(NS, -1).sum(0)
    print(f"{name:20s} {k0}: {t0:7.1f} us   {k1}: {t1:7.1f} us   max|diff| / max|C| = {(o0 - o1).abs().max().item() / scale:.2e}   "
          f"column sums {(s4 - s1.view(NS, -1).sum(0)).abs().max().item() / (s4.abs().max().item() + 1e-9):.2e}")
