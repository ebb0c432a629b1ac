"""Randomised sweeps of the row kernels (rel-pos softmax, LayerNorm, GLU+dwconv) against torch math."""
import sys, os, math, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from a3t_amd import ops
from oracle import a3t_oracle as O
dev = "cuda"
fails = 0


def close(name, got, ref, atol, rtol):
    global fails
    g, r = got.float().cpu(), ref.float().cpu()
    bad = ~torch.isclose(g, r, atol=atol, rtol=rtol)
    if bool(bad.any()) or not bool(torch.isfinite(g).all()):
        fails += 1
        print(f"FAIL {name}: max abs err {float((g - r).abs().max()):.3e} ({int(bad.sum())} elements)")


def softmax_case(rng):
    B, H = rng.randrange(1, 4), rng.randrange(1, 5)
    T = rng.choice([rng.randrange(1, 40), rng.randrange(40, 300), 8 * rng.randrange(1, 40)])
    bf = rng.random() < 0.5
    tdt = torch.bfloat16 if bf else torch.float32
    gen = torch.Generator().manual_seed(rng.randrange(1 << 30))
    ac = (torch.randn(B, H, T, T, generator=gen) * 3).to(tdt).float().requires_grad_(True)
    bd = (torch.randn(B, H, T, T, generator=gen) * 3).to(tdt).float().requires_grad_(True)
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    for b in range(B):
        r = rng.random()
        if r < 0.2:
            mask[b] = False
        elif r < 0.6:
            mask[b, 0, rng.randrange(0, T):] = False
    scale = rng.choice([0.3, 0.0721, 1.0])
    s = (ac + O.rel_shift_legacy(bd)) * scale
    m = mask.unsqueeze(1).eq(0)
    pr = torch.softmax(s.masked_fill(m, float(np.finfo(np.float32).min)), dim=-1).masked_fill(m, 0.0)
    dp = torch.randn(B, H, T, T, generator=gen).to(tdt).float()
    pr.backward(dp)
    probs = torch.empty(B, H, T, T, device=dev, dtype=tdt)
    ops.relpos_softmax_fwd(ac.detach().to(dev).to(tdt), bd.detach().to(dev).to(tdt), mask.view(B, T).to(dev).view(torch.uint8),
                           probs, B, H, T, scale)
    tag = f"softmax B{B} H{H} T{T} {'bf16' if bf else 'f32'}"
    close(tag + " fwd", probs, pr, *( (4e-3, 1e-2) if bf else (1e-6, 1e-4)))
    ds = torch.empty(B, H, T, T, device=dev, dtype=tdt)
    dbd = torch.full((B, H, T, T), 7.0, device=dev, dtype=tdt)
    pin = probs if bf else pr.detach().to(dev)
    ops.relpos_softmax_bwd(pin, dp.to(dev).to(tdt), ds, dbd, B, H, T, scale)
    close(tag + " ds", ds, ac.grad, *((8e-3, 3e-2) if bf else (1e-6, 1e-3)))
    close(tag + " dbd", dbd, bd.grad, *((8e-3, 3e-2) if bf else (1e-6, 1e-3)))


def layernorm_case(rng):
    M = rng.choice([rng.randrange(1, 70), rng.randrange(70, 3000)])
    D = rng.choice([8 * rng.randrange(1, 60), 128, 256, 384, 512, rng.randrange(3, 200)])
    gen = torch.Generator().manual_seed(rng.randrange(1 << 30))
    x = torch.randn(M, D, generator=gen).requires_grad_(True)
    g = (1 + 0.2 * torch.randn(D, generator=gen)).requires_grad_(True)
    b = (0.1 * torch.randn(D, generator=gen)).requires_grad_(True)
    eps = rng.choice([1e-12, 1e-5])
    y = F.layer_norm(x, (D,), g, b, eps)
    dy = torch.randn(M, D, generator=gen)
    dres = torch.randn(M, D, generator=gen)
    y.backward(dy)
    yd = torch.empty(M, D, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x.detach().to(dev), g.detach().to(dev), b.detach().to(dev), yd, mean, rstd, eps)
    tag = f"layernorm M{M} D{D}"
    close(tag + " fwd", yd, y, 2e-5, 1e-4)
    dx = torch.empty(M, D, device=dev)
    dg, db, cs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ops.layernorm_bwd(dy.to(dev), x.detach().to(dev), g.detach().to(dev), mean, rstd, dres.to(dev), dx, dg, db,
                      dxsum=cs if D % 128 == 0 or True else None, dxsum_scale=0.5)
    close(tag + " dx", dx, x.grad + dres, 5e-5, 1e-3)
    close(tag + " dgamma", dg, g.grad, 1e-3 * math.sqrt(M), 1e-3)
    close(tag + " dbeta", db, b.grad, 1e-3 * math.sqrt(M), 1e-3)
    close(tag + " dxsum", cs, 0.5 * (x.grad + dres).sum(0), 1e-3 * math.sqrt(M), 1e-3)


def dwconv_case(rng):
    B, T = rng.randrange(1, 5), rng.choice([rng.randrange(1, 70), rng.randrange(70, 400)])
    C = rng.choice([8 * rng.randrange(1, 30), 64, 128, 384])
    K = rng.choice([3, 5, 7, 15, 31])
    bf = rng.random() < 0.5
    gen = torch.Generator().manual_seed(rng.randrange(1 << 30))
    g = torch.randn(B, T, 2 * C, generator=gen)
    if bf:
        g = g.bfloat16().float()
    g.requires_grad_(True)
    w = (torch.randn(C, 1, K, generator=gen) * K ** -0.5).requires_grad_(True)
    b = torch.randn(C, generator=gen).requires_grad_(True)
    glu = F.glu(g.transpose(1, 2), dim=1)
    z = F.conv1d(glu, w, b, padding=(K - 1) // 2, groups=C).transpose(1, 2)
    dz = torch.randn(B, T, C, generator=gen)
    z.backward(dz)
    tdt = torch.bfloat16 if bf else torch.float32
    gd = g.detach().reshape(B * T, 2 * C).to(dev).to(tdt)
    wd, bd_ = w.detach().reshape(C, K).contiguous().to(dev), b.detach().to(dev)
    glud = torch.empty(B * T, C, device=dev, dtype=tdt)
    zd = torch.empty(B * T, C, device=dev)
    ops.glu_dwconv_fwd(gd, wd, bd_, glud, zd, T)
    tag = f"dwconv B{B} T{T} C{C} K{K} {'bf16' if bf else 'f32'}"
    close(tag + " z", zd.view(B, T, C), z, 5e-5, 1e-4)
    dg = torch.empty(B * T, 2 * C, device=dev, dtype=tdt)
    dw, dbb = torch.zeros(C, K, device=dev), torch.zeros(C, device=dev)
    ops.glu_dwconv_bwd(dz.reshape(B * T, C).to(dev), gd, glud, wd, dg, dw, dbb, T)
    close(tag + " dg", dg.view(B, T, 2 * C), g.grad, *((3e-2, 3e-2) if bf else (5e-5, 1e-4)))
    sc = float(w.grad.abs().max()) + 1e-6
    close(tag + " dw", dw, w.grad.view(C, K), (2e-2 if bf else 1e-3) * sc, 3e-2 if bf else 1e-3)
    close(tag + " db", dbb, b.grad, 1e-3 * (float(b.grad.abs().max()) + 1e-6) + 1e-4, 1e-3)


def run(seed=0, n=40, verbose=True):
    global fails
    fails = 0
    rng = random.Random(seed)
    for i in range(n):
        for fn in (softmax_case, layernorm_case, dwconv_case):
            try:
                fn(rng)
            except Exception as e:  # noqa: BLE001
                fails += 1
                print(f"EXC {fn.__name__} #{i}: {type(e).__name__}: {e}")
    torch.cuda.synchronize()
    if verbose:
        print(f"{3 * n} cases, {fails} failures")
    return fails


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
