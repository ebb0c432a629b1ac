"""384-column panel GEMM (csrc/gemm_bf16_pn.hip) against the 128x128 kernel and torch math; timing of the model's N = 384 shapes.
   python tools/pn_check.py [quick]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import ACT_NONE, ACT_RELU, BF16

DEV = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc


def both(fn):
    outs = []
    old8 = lib.a3t_gemm_8p_mode(0)
    for mode in (0, 1):
        lib.a3t_gemm_pn_mode(mode)
        outs.append(fn())
        outs.append(lib.a3t_gemm_last_kernel().decode())
    lib.a3t_gemm_pn_mode(2)
    lib.a3t_gemm_8p_mode(old8)
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
N = 384
# ---- plain linear: M tails, several rounds, bias / relu / fp32 out / residual / dropout / column sums
for (M, K, act, f32out, res, drop, cs) in [(160, 128, ACT_NONE, False, False, False, False), (1000, 384, ACT_RELU, False, False, False, True),
                                           (4099 * 8, 768, ACT_NONE, True, True, True, False), (45000, 256, ACT_NONE, False, False, True, True),
                                           (777, 1152, ACT_RELU, True, True, False, True)]:
    x, W, b = rn(M, K).bfloat16(), rn(N, K, sc=0.05).bfloat16(), rn(N)
    R = rn(M, N) if res else None

    def f():
        o = torch.empty(M, N, device=DEV, dtype=torch.float32 if f32out else torch.bfloat16)
        csum = torch.zeros(N, device=DEV) if cs else None
        ops.gemm(x, W, o, M, N, K, K, 1, K, 1, N, bias=b, R=R, alpha=0.7, act=act, compute=BF16, drop=(0.1, 99) if drop else None,
                 colsum=csum, colsum_scale=0.5)
        return (o, csum)
    (o0, c0), k0, (o1, c1), k1 = both(f)
    e = rel(o1, o0)
    exact = bool(torch.equal(o1, o0))
    ec = rel(c1, c0) if cs else 0.0
    if not drop:
        ref = x.float() @ W.float().t() + b
        ref = (torch.relu(ref) if act == ACT_RELU else ref) * 0.7 + (R if res else 0)
        et = rel(o1, ref)
    else:
        et = 0.0
    ok = "pn" in k1 and "pn" not in k0 and e < 1e-6 and et < 1e-2 and ec < 1e-4
    bad += not ok
    print(f"linear {M}x{N}x{K} act={act} f32={f32out} R={res} drop={drop} colsum={cs}: {k1} vs 128^2 {e:.1e} (bit-equal {exact}) "
          f"vs torch {et:.1e} colsum {ec:.1e} {'ok' if ok else 'FAIL'}")

# ---- conv over time (taps 3 and 5, utterance boundaries, M tail): conv2 forward and the data gradient through a transposed shadow
for (B, T, cin, taps) in [(3, 200, 128, 3), (7, 333, 256, 5)] + ([(32, 1120, 1536, 3)] if len(sys.argv) < 2 else []):
    M = B * T
    h = rn(M, cin).bfloat16()
    W2 = rn(N, taps, cin, sc=0.03).bfloat16()
    b2, xres = rn(N), rn(M, N)

    def f():
        o = torch.empty(M, N, device=DEV)
        ops.conv_fwd(h, W2, o, T, (taps - 1) // 2, bias=b2, R=xres, alpha=0.5, compute=BF16, drop=(0.1, 4242))
        return o
    o0, k0, o1, k1 = both(f)
    e = rel(o1, o0)
    ok = "pn" in k1 and e < 1e-6
    bad += not ok
    print(f"conv fwd B={B} T={T} {cin}x{taps}->{N}: {k1} vs 128^2 {e:.1e} (bit-equal {bool(torch.equal(o1, o0))}) {'ok' if ok else 'FAIL'}")

    def f2():
        o = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.conv_fwd(h, W2, o, T, taps - 1 - (taps - 1) // 2, compute=BF16)
        return o
    o0, k0, o1, k1 = both(f2)
    ref = torch.nn.functional.conv1d(h.float().view(B, T, cin).transpose(1, 2), W2.float().permute(0, 2, 1), padding=(taps - 1) // 2)
    et = rel(o1, ref.transpose(1, 2).reshape(M, N))
    ok = "pn" in k1 and rel(o1, o0) < 1e-6 and et < 1e-2
    bad += not ok
    print(f"   plain conv: vs 128^2 {rel(o1, o0):.1e} vs torch {et:.1e} {'ok' if ok else 'FAIL'}")

# ---- wider outputs: column chunks of 384 (N = 768 / 1152 / 1536, and a ragged N), the ReLU' mask tensor S + column sums
for (M, Nw, K, taps, smask) in [(2000, 768, 384, 1, False), (3000, 1152, 384, 1, False), (1800, 1536, 384, 3, False), (1800, 1536, 384, 3, True),
                                (1234, 1000, 256, 1, True), (41000, 768, 128, 1, False)]:
    cin = K // taps
    T = 100 if taps > 1 else 0
    a = rn(M, cin).bfloat16()
    W = rn(Nw, taps, cin, sc=0.05).bfloat16()
    b = rn(Nw)
    S = rn(M, Nw).bfloat16() if smask else None

    def f():
        o = torch.empty(M, Nw, device=DEV, dtype=torch.bfloat16)
        csum = torch.zeros(Nw, device=DEV)
        if taps > 1:
            ops.conv_fwd(a, W, o, T, 1, bias=None if smask else b, act=ACT_NONE if smask else ACT_RELU, alpha=0.8, compute=BF16,
                         drop=None if smask else (0.1, 5), S=S, colsum=csum)
        else:
            ops.gemm(a, W, o, M, Nw, K, K, 1, K, 1, Nw, bias=b, S=S, alpha=0.8, compute=BF16, colsum=csum)
        return (o, csum)
    (o0, c0), k0, (o1, c1), k1 = both(f)
    ok = "pn" in k1 and "pn" not in k0 and bool(torch.equal(o0, o1)) and rel(c1, c0) < 1e-4
    bad += not ok
    print(f"wide {M}x{Nw}x{K} taps={taps} S={smask}: {k1} bit-equal {bool(torch.equal(o0, o1))} colsum {rel(c1, c0):.1e} {'ok' if ok else 'FAIL'}")

print("FAILED" if bad else "all ok")

# ---- timing: the N = 384 GEMMs of configs[1] (B=32, T=1120)
if len(sys.argv) < 2:
    M, T = 35840, 1120
    xres = rn(M, N)
    b = rn(N)
    h = rn(M, 1536).bfloat16()
    for name, Nw, K, taps, kw in [("ffn conv1 fwd (bias+relu+drop)", 1536, 1152, 3, dict(relu=True, drop=True)),
                                  ("ffn conv2 dgrad (S mask + colsum)", 1536, 1152, 3, dict(S=True, cs=True)),
                                  ("qkv fwd (bias)", 1152, 384, 1, dict(bias=True)),
                                  ("pw1 fwd (bias)", 768, 384, 1, dict(bias=True))]:
        cin = K // taps
        a = rn(M, cin).bfloat16()
        W = rn(Nw, taps, cin, sc=0.03).bfloat16()
        bw = rn(Nw)
        o = torch.empty(M, Nw, device=DEV, dtype=torch.bfloat16)
        csum = torch.zeros(Nw, device=DEV)

        def run():
            ekw = dict(bias=bw if (kw.get("relu") or kw.get("bias")) else None, act=ACT_RELU if kw.get("relu") else ACT_NONE, compute=BF16,
                       drop=(0.1, 7) if kw.get("drop") else None)
            if taps > 1:
                ops.conv_fwd(a, W, o, T, 1, S=h if kw.get("S") else None, colsum=csum if kw.get("cs") else None, **ekw)
            else:
                ops.linear_fwd(a, W.view(Nw, cin), o, **ekw)
        old8 = lib.a3t_gemm_8p_mode(0)
        ts = []
        for mode in (0, 1, 2):
            lib.a3t_gemm_pn_mode(mode)
            ts.append(timeit(run))
            kn = lib.a3t_gemm_last_kernel().decode()
        lib.a3t_gemm_8p_mode(old8)
        fl = 2.0 * M * Nw * K
        print(f"{name:42s} N={Nw:4d} K={K:5d}: 128^2 {ts[0]:6.1f} us ({fl/ts[0]/1e6:5.0f} TF)  panel {ts[1]:6.1f} us ({fl/ts[1]/1e6:5.0f} TF)  "
              f"default {ts[2]:6.1f} us [{kn}]")
    for name, K, taps, kw in [("ffn conv2 fwd (bias+drop+R, fp32 out)", 4608, 3, dict(f32=True, R=True, drop=True)),
                              ("ffn conv1 dgrad (bf16 out)", 4608, 3, dict()),
                              ("linear_out / pw2 fwd (bias+drop+R, fp32)", 384, 1, dict(f32=True, R=True, drop=True)),
                              ("dgrad of linear_out / pw2", 384, 1, dict()),
                              ("dgrad of pw1", 768, 1, dict()),
                              ("dgrad of qkv", 1152, 1, dict())]:
        cin = K // taps
        a = rn(M, cin).bfloat16()
        W = rn(N, taps, cin, sc=0.03).bfloat16()
        o = torch.empty(M, N, device=DEV, dtype=torch.float32 if kw.get("f32") else torch.bfloat16)

        def run():
            ekw = dict(bias=b if kw.get("R") else None, R=xres if kw.get("R") else None, alpha=0.5, compute=BF16,
                       drop=(0.1, 7) if kw.get("drop") else None)
            if taps > 1:
                ops.conv_fwd(a, W, o, T, (taps - 1) // 2, **ekw)
            else:
                ops.linear_fwd(a, W.view(N, cin), o, **ekw)
        old8 = lib.a3t_gemm_8p_mode(0)
        ts = []
        for mode in (0, 1, 2):
            lib.a3t_gemm_pn_mode(mode)
            ts.append(timeit(run))
            kn = lib.a3t_gemm_last_kernel().decode()
        lib.a3t_gemm_8p_mode(old8)
        fl = 2.0 * M * N * K
        print(f"{name:42s} K={K:5d}: 128^2 {ts[0]:6.1f} us ({fl/ts[0]/1e6:5.0f} TF)  panel {ts[1]:6.1f} us ({fl/ts[1]/1e6:5.0f} TF)  "
              f"default {ts[2]:6.1f} us [{kn}]")
sys.exit(1 if bad else 0)
