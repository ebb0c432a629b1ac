"""8-phase GEMM (csrc/gemm_bf16_8p.hip) against the 128x128 kernel and torch math; timing of the FFN shapes.
   python tools/g8_check.py [quick]"""
import sys
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import ACT_NONE, ACT_RELU, BF16

DEV = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc


def both(fn):
    outs = []
    for mode in (0, 1):
        lib.a3t_gemm_8p_mode(mode)
        outs.append(fn())
        outs.append(lib.a3t_gemm_last_kernel().decode())
    lib.a3t_gemm_8p_mode(2)
    torch.cuda.synchronize()
    return outs


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-9))


def timeit(fn, n=50):
    for _ in range(300):      # the clocks must have ramped up before anything is timed
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


bad = 0
# ---- plain linear, tails in M and N, bias / relu / fp32 out
for (M, N, K, act, f32out) in [(256, 256, 128, ACT_NONE, False), (1000, 520, 384, ACT_RELU, False), (4096, 1024, 1024, ACT_NONE, True),
                               (777, 264, 256, ACT_RELU, True)]:
    x, W, b = rn(M, K).bfloat16(), rn(N, K, sc=0.05).bfloat16(), rn(N)

    def f():
        o = torch.empty(M, N, device=DEV, dtype=torch.float32 if f32out else torch.bfloat16)
        ops.linear_fwd(x, W, o, bias=b, act=act, alpha=0.7, compute=BF16)
        return o
    o0, k0, o1, k1 = both(f)
    ref = (x.float() @ W.float().t() + b)
    ref = (torch.relu(ref) if act == ACT_RELU else ref) * 0.7
    e = rel(o1, ref), rel(o1, o0)
    ok = "8p" in k1 and e[0] < 1e-2 and e[1] < 1e-2
    bad += not ok
    print(f"linear {M}x{N}x{K} act={act} f32={f32out}: {k1} err vs torch {e[0]:.2e} vs 128^2 {e[1]:.2e} {'ok' if ok else 'FAIL'}")

# ---- conv forward: bias + relu + dropout + keep bits; data gradient with keep bits + column sums
for (B, T, d, ff) in [(3, 200, 128, 512), (9, 1120, 384, 1536)] if len(sys.argv) < 2 else [(3, 200, 128, 512)]:
    M = B * T
    y = rn(M, d).bfloat16()
    W1 = rn(ff, 3, d, sc=0.03).bfloat16()
    b1 = rn(ff)
    keep = torch.zeros(ops.gemm_keep_bytes(M, ff), dtype=torch.uint8, device=DEV)

    def f():
        h = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
        ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 777))
        return h
    h0, k0, h1, k1 = both(f)
    e = rel(h1, h0)
    nz = float(((h1 != 0) != (h0 != 0)).float().mean())
    ok = "8p" in k1 and e < 1e-2 and nz < 1e-3
    bad += not ok
    print(f"conv fwd B={B} T={T} {d}->{ff}: {k1} err vs 128^2 {e:.2e}, zero-pattern mismatch {nz:.2e} {'ok' if ok else 'FAIL'}")
    if ff % 256 == 0:
        lib.a3t_gemm_8p_mode(1)
        h2 = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
        ops.conv_fwd(y, W1, h2, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 777), keep_out=keep)
        # data gradient of the second conv: dh = relu'(h) * (ga * W2^T) through the transposed weights, as a forward conv
        ga = rn(M, d).bfloat16()
        W2 = rn(d, 3, ff, sc=0.02).bfloat16()
        W2t = W2.permute(2, 1, 0).flip(1).contiguous()          # [ff][tap'][d]: NT operand of the data gradient
        dh1 = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
        gb1 = torch.zeros(ff, device=DEV)
        ops.conv_fwd(ga, W2t, dh1, T, 1, alpha=0.625, compute=BF16, keep_in=keep, colsum=gb1)
        k2 = lib.a3t_gemm_last_kernel().decode()
        lib.a3t_gemm_8p_mode(0)
        dh0 = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
        gb0 = torch.zeros(ff, device=DEV)
        ops.conv_bwd_data(ga, W2, dh0, T, 1, S=h2, alpha=0.625, compute=BF16, colsum=gb0)
        lib.a3t_gemm_8p_mode(2)
        torch.cuda.synchronize()
        e1, e2 = rel(dh1, dh0), rel(gb1, gb0)
        ok = "8p" in k2 and e1 < 1e-2 and e2 < 1e-2 and bool((h2 == h1).all())
        bad += not ok
        print(f"   data gradient through keep bits: {k2} err {e1:.2e}, colsum err {e2:.2e} {'ok' if ok else 'FAIL'}")

# ---- timing at the benchmark shapes
if len(sys.argv) < 2:
    B, T, d, ff = 32, 1120, 384, 1536
    M = B * T
    y, W1, b1 = rn(M, d).bfloat16(), rn(ff, 3, d, sc=0.03).bfloat16(), rn(ff)
    h = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
    keep = torch.zeros(ops.gemm_keep_bytes(M, ff), dtype=torch.uint8, device=DEV)
    ga, W2 = rn(M, d).bfloat16(), rn(d, 3, ff, sc=0.02).bfloat16()
    W2t = W2.permute(2, 1, 0).flip(1).contiguous()
    dh = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
    gb = torch.zeros(ff, device=DEV)
    fl = 2.0 * M * ff * 3 * d
    xl, Wl = rn(M, 3 * d).bfloat16(), rn(ff, 3 * d, sc=0.03).bfloat16()
    for mode in (0, 1):
        lib.a3t_gemm_8p_mode(mode)
        t0 = timeit(lambda: ops.linear_fwd(xl, Wl, h, compute=BF16))
        print(f"mode {mode}: plain linear {M}x{ff}x{3 * d}: {t0:.1f} us ({fl / t0 / 1e6:.0f} TF)")
        if mode:
            ta = timeit(lambda: ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.625, compute=BF16, keep_in=keep))
            tb = timeit(lambda: ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.625, compute=BF16, colsum=gb))
            print(f"        data-grad shape: keep_in only {ta:.1f} us, colsum only {tb:.1f} us")
        t1 = timeit(lambda: ops.conv_fwd(y, W1, h, T, 1, compute=BF16))
        t2 = timeit(lambda: ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16))
        t3 = timeit(lambda: ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 777)))
        if mode:
            t4 = timeit(lambda: ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 777), keep_out=keep))
            t5 = timeit(lambda: ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.625, compute=BF16, keep_in=keep, colsum=gb))
        else:
            t4 = float("nan")
            t5 = timeit(lambda: ops.conv_bwd_data(ga, W2, dh, T, 1, S=h, alpha=0.625, compute=BF16, colsum=gb))
        print(f"mode {mode}: conv1 fwd plain {t1:.1f} us ({fl / t1 / 1e6:.0f} TF) | +bias+relu {t2:.1f} | +dropout {t3:.1f} | +keep_out {t4:.1f} | "
              f"conv2 data grad (mask + colsum) {t5:.1f} us ({fl / t5 / 1e6:.0f} TF)")
    lib.a3t_gemm_8p_mode(2)
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
