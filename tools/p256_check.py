"""Persistent 256x256 GEMM (gemm_bf16_p256.hip) against the 128x128 kernel on the same descriptors: the conv-FFN launches
of the benchmark with their fused epilogues, then ragged shapes (M / N tails, left-over pieces of 1, 2 and 4 quadrants,
batched products).  Prints max |diff| and the timing of both kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops, _lib
from a3t_amd._lib import ACT_RELU, ACT_NONE, BF16

lib = _lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(name, make, flops, time=True):
    """make() -> (fn, outputs): run under both kernels, compare every output."""
    res = []
    for mode in (0, 1):
        lib.a3t_gemm_p256_mode(mode)
        fn, outs = make()
        for o in outs:
            if o.dtype == torch.float32 and getattr(o, "_acc", False):
                o.zero_()
        fn()
        torch.cuda.synchronize()
        kern = lib.a3t_gemm_last_kernel().decode()
        vals = [o.float().clone() for o in outs]
        us = timeit(fn) if time else 0.0
        res.append((kern, vals, us))
    lib.a3t_gemm_p256_mode(0)
    worst = 0.0
    for a, b in zip(res[0][1], res[1][1]):
        worst = max(worst, float((a - b).abs().max() / (a.abs().max() + 1e-9)))
    used = "p256" in res[1][0]
    tag = "OK " if (worst < 4e-3 and used) else ("SKIP" if not used else "FAIL")
    print(f"{tag} {name}: rel diff {worst:.2e}  128x128 {res[0][2]:7.1f} us  p256 {res[1][2]:7.1f} us "
          f"({flops / max(res[1][2], 1e-9) / 1e6:6.0f} TFLOP/s)  [{res[1][0]}]", flush=True)
    return tag != "FAIL"


ok = True
B, T, d, ff = 32, 1120, 384, 1536
M = B * T
y = rn(M, d).bfloat16()
x = rn(M, d)
W1 = rn(ff, 3, d, sc=0.03).bfloat16()
W2 = rn(d, 3, ff, sc=0.02).bfloat16()
b1, b2 = rn(ff), rn(d)
ga = rn(M, d).bfloat16()
hh = torch.relu(rn(M, ff)).bfloat16()
dh0 = rn(M, ff).bfloat16()
DR = (0.2, 777)
fl = 2.0 * M * d * ff * 3


def f1():
    h = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
    return (lambda: ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=DR)), [h]


def f2():
    xo = torch.empty(M, d, device=dev)
    return (lambda: ops.conv_fwd(hh, W2, xo, T, 1, bias=b2, R=x, alpha=0.5, compute=BF16, drop=DR)), [xo]


def b1_():
    dh = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
    gb = torch.zeros(ff, device=dev)

    def fn():
        gb.zero_()
        ops.conv_bwd_data(ga, W2, dh, T, 1, S=hh, alpha=0.625, compute=BF16, colsum=gb)
    return fn, [dh, gb]


def b3():
    dy = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    return (lambda: ops.conv_bwd_data(dh0, W1, dy, T, 1, compute=BF16)), [dy]


ok &= ab("F1 conv1 fwd  NT 35840x1536x1152 bias+relu+drop", f1, fl)
ok &= ab("F2 conv2 fwd  NT 35840x384x4608 bias+drop+res  ", f2, fl)
ok &= ab("B1 conv2 dgrad NN 35840x1536x1152 mask+colsum   ", b1_, fl)
ok &= ab("B3 conv1 dgrad NN 35840x384x4608               ", b3, fl)

# ragged shapes: plain linears (NT fwd, NN dgrad), M / N tails, every left-over piece size, few tiles
for (Mx, N, K) in [(35840, 384, 384), (35840, 1536, 384), (70000, 520, 192), (256 * 300 + 8, 256, 128), (256 * 70, 512, 256),
                   (256 * 33, 256, 640), (1000, 264, 128), (256 * 256 + 128, 1024, 64 * 5)]:
    xa = rn(Mx, K, sc=0.5).bfloat16()
    Wl = rn(N, K, sc=K ** -0.5).bfloat16()
    bl = rn(N)
    Rl = rn(Mx, N)
    dyl = rn(Mx, N, sc=0.5).bfloat16()

    def lf():
        o = torch.empty(Mx, N, device=dev)
        return (lambda: ops.linear_fwd(xa, Wl, o, bias=bl, R=Rl, act=ACT_RELU, alpha=0.7, compute=BF16)), [o]

    def lb():
        o = torch.empty(Mx, K, device=dev, dtype=torch.bfloat16)
        return (lambda: ops.linear_bwd_data(dyl, Wl, o, compute=BF16)), [o]

    ok &= ab(f"linear_fwd      {Mx}x{N}x{K}", lf, 2.0 * Mx * N * K)
    ok &= ab(f"linear_bwd_data {Mx}x{K}x{N}", lb, 2.0 * Mx * N * K)

# dilated / 5-tap convs with utterance boundaries inside tiles
for (Bc, Tc, Cin, Cout, taps, dil) in [(9, 1030, 128, 384, 5, 2), (64, 300, 64, 520, 3, 1), (3, 20000, 192, 256, 3, 1)]:
    Mx, pad = Bc * Tc, (taps - 1) // 2
    xa = rn(Mx, Cin, sc=0.5).bfloat16()
    Wk = rn(Cout, taps, Cin, sc=(Cin * taps) ** -0.5).bfloat16()
    dyl = rn(Mx, Cout, sc=0.5).bfloat16()

    def cf():
        o = torch.empty(Mx, Cout, device=dev, dtype=torch.bfloat16)
        return (lambda: ops.conv_fwd(xa, Wk, o, Tc, pad, dil, compute=BF16)), [o]

    def cb():
        o = torch.empty(Mx, Cin, device=dev)
        return (lambda: ops.conv_bwd_data(dyl, Wk, o, Tc, pad, dil, compute=BF16)), [o]

    ok &= ab(f"conv_fwd      B{Bc} T{Tc} {Cin}->{Cout} k{taps} d{dil}", cf, 2.0 * Mx * Cin * Cout * taps)
    ok &= ab(f"conv_bwd_data B{Bc} T{Tc} {Cout}->{Cin} k{taps} d{dil}", cb, 2.0 * Mx * Cin * Cout * taps)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
