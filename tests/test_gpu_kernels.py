"""GPU parity tests: every HIP kernel (called through the C ABI) against the CPU oracle /
plain torch-CPU fp32 math on identical seeded inputs.  Run with `pytest -m gpu` on an MI355X."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import a3t_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from a3t_amd import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


def _close(got, ref, atol, rtol, msg=""):
    got = got.detach().float().cpu().numpy()
    ref = ref.detach().float().cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=atol, rtol=rtol, err_msg=msg)


TOL = {"f32": dict(atol=2e-4, rtol=2e-4), "bf16": dict(atol=0.15, rtol=5e-2)}


@pytest.mark.parametrize("cmp", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 80, 384), (1120, 192, 96), (37, 50, 24), (333, 264, 1152)])
def test_linear_fwd_bwd(cmp, M, N, K):
    ops = _ops()
    from a3t_amd._lib import BF16, F32, ACT_RELU
    c = BF16 if cmp == "bf16" else F32
    if cmp == "bf16" and (K % 8 or N % 8):
        pytest.skip("bf16 path needs 8-element alignment")
    x, W, b, R = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3), _rand(M, N, seed=4)
    xd, Wd, bd, Rd = x.to(DEV), W.to(DEV), b.to(DEV), R.to(DEV)
    out = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xd, Wd, out, bias=bd, R=Rd, alpha=0.5, act=ACT_RELU, compute=c)
    ref = 0.5 * torch.relu(F.linear(x, W, b)) + R
    _close(out, ref, **TOL[cmp])
    dy = _rand(M, N, seed=5)
    dyd = dy.to(DEV)
    dx = torch.empty(M, K, device=DEV)
    ops.linear_bwd_data(dyd, Wd, dx, compute=c)
    _close(dx, dy @ W, **TOL[cmp])
    dW = torch.zeros(N, K, device=DEV)
    ops.linear_bwd_weight(dyd, xd, dW, compute=c)
    tol = dict(TOL[cmp])
    tol["atol"] *= math.sqrt(M)
    _close(dW, dy.t() @ x, **tol)


@pytest.mark.parametrize("cmp", ["f32", "bf16"])
@pytest.mark.parametrize("B,T,Cin,Cout,taps,dil", [(2, 56, 32, 64, 3, 1), (3, 37, 80, 256, 5, 1), (1, 300, 64, 128, 3, 4),
                                                   (2, 200, 384, 1536, 3, 1)])
def test_conv_as_gemm(cmp, B, T, Cin, Cout, taps, dil):
    ops = _ops()
    from a3t_amd._lib import BF16, F32
    c = BF16 if cmp == "bf16" else F32
    pad = (taps - 1) // 2
    x = _rand(B, T, Cin, seed=1).requires_grad_(True)
    W = _rand(Cout, Cin, taps, seed=2, scale=(Cin * taps) ** -0.5).requires_grad_(True)
    b = _rand(Cout, seed=3)
    y = F.conv1d(x.transpose(1, 2), W, b, padding=pad * dil, dilation=dil).transpose(1, 2)
    dy = _rand(B, T, Cout, seed=4)
    y.backward(dy)
    Wk = W.detach().permute(0, 2, 1).contiguous().to(DEV)
    xd = x.detach().reshape(B * T, Cin).to(DEV)
    out = torch.empty(B * T, Cout, device=DEV)
    ops.conv_fwd(xd, Wk, out, T, pad, dil, bias=b.to(DEV), compute=c)
    _close(out.view(B, T, Cout), y, **TOL[cmp])
    dyd = dy.reshape(B * T, Cout).to(DEV)
    dx = torch.empty(B * T, Cin, device=DEV)
    ops.conv_bwd_data(dyd, Wk, dx, T, pad, dil, compute=c)
    _close(dx.view(B, T, Cin), x.grad, **TOL[cmp])
    dWk = torch.zeros_like(Wk)
    ops.conv_bwd_weight(dyd, xd, dWk, T, pad, dil, compute=c)
    tol = dict(TOL[cmp])
    tol["atol"] *= math.sqrt(B * T)
    _close(dWk.permute(0, 2, 1), W.grad, **tol)


@pytest.mark.parametrize("M,D,eps", [(7, 384, 1e-12), (1000, 384, 1e-5), (130, 32, 1e-12), (65, 80, 1e-12)])
def test_layernorm(M, D, eps):
    ops = _ops()
    x = _rand(M, D, seed=1, scale=2.0).requires_grad_(True)
    g = (1 + 0.1 * _rand(D, seed=2)).requires_grad_(True)
    b = _rand(D, seed=3).requires_grad_(True)
    y = F.layer_norm(x, (D,), g, b, eps)
    dy, res = _rand(M, D, seed=4), _rand(M, D, seed=5)
    y.backward(dy)
    xd = x.detach().to(DEV)
    yd, mean, rstd = torch.empty(M, D, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(xd, g.detach().to(DEV), b.detach().to(DEV), yd, mean, rstd, eps)
    _close(yd, y, atol=2e-5, rtol=1e-5)
    dx = torch.empty(M, D, device=DEV)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dy.to(DEV), xd, g.detach().to(DEV), mean, rstd, res.to(DEV), dx, dg, db)
    _close(dx, x.grad + res, atol=5e-5, rtol=1e-4)
    _close(dg, g.grad, atol=1e-4 * math.sqrt(M), rtol=1e-4)
    _close(db, b.grad, atol=1e-4 * math.sqrt(M), rtol=1e-4)


@pytest.mark.parametrize("act", ["swish", "tanh", "none"])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("M,C", [(777, 96), (1003, 128), (50, 30)])   # generic / 16-byte vector / unaligned paths
def test_batchnorm_act(act, training, M, C):
    ops = _ops()
    from a3t_amd._lib import ACT_NONE, ACT_SWISH, ACT_TANH
    A = dict(swish=ACT_SWISH, tanh=ACT_TANH, none=ACT_NONE)[act]
    z = (_rand(M, C, seed=1) * 1.5 + 0.3).requires_grad_(True)
    g = (1 + 0.2 * _rand(C, seed=2)).requires_grad_(True)
    b = (0.1 * _rand(C, seed=3)).requires_grad_(True)
    rm, rv = 0.1 * _rand(C, seed=4), 0.5 + torch.rand(C)
    rm0, rv0 = rm.clone(), rv.clone()
    bn = F.batch_norm(z.t()[None], rm, rv, g, b, training, 0.1, 1e-5)[0].t()
    y = bn * torch.sigmoid(bn) if act == "swish" else (torch.tanh(bn) if act == "tanh" else bn)
    dy = _rand(M, C, seed=5)
    y.backward(dy)
    zd = z.detach().to(DEV)
    stats = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    if training:
        ops.col_reduce(zd, stats[:C], stats[C:], mode=1)
    rmd, rvd = rm0.to(DEV), rv0.to(DEV)
    mean, rstd, yd = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(M, C, device=DEV)
    ops.bn_act_fwd(zd, stats, g.detach().to(DEV), b.detach().to(DEV), rmd, rvd, mean, rstd, yd, 1e-5,
                   0.1 if training else 0.0, training, A)
    _close(yd, y, atol=2e-5, rtol=1e-4)
    _close(rmd, rm, atol=1e-6, rtol=1e-5)
    _close(rvd, rv, atol=1e-6, rtol=1e-5)
    dz = torch.empty(M, C, device=DEV)
    sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_act_bwd(dy.to(DEV), zd, mean, rstd, g.detach().to(DEV), b.detach().to(DEV), sums, dz, dg, db,
                   training, A)
    _close(dz, z.grad, atol=5e-5, rtol=1e-3)
    _close(dg, g.grad, atol=2e-3, rtol=1e-4)
    _close(db, b.grad, atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("B,T,C,K", [(2, 50, 32, 7), (3, 130, 96, 31), (1, 64, 384, 7)])
def test_glu_dwconv(B, T, C, K):
    ops = _ops()
    g = _rand(B, T, 2 * C, seed=1).requires_grad_(True)
    w = _rand(C, 1, K, seed=2, scale=K ** -0.5).requires_grad_(True)
    b = _rand(C, seed=3).requires_grad_(True)
    glu = F.glu(g.transpose(1, 2), dim=1)
    z = F.conv1d(glu, w, b, padding=(K - 1) // 2, groups=C).transpose(1, 2)
    dz = _rand(B, T, C, seed=4)
    z.backward(dz)
    gd, wd, bd = g.detach().reshape(B * T, 2 * C).to(DEV), w.detach().reshape(C, K).contiguous().to(DEV), b.detach().to(DEV)
    glud, zd = torch.empty(B * T, C, device=DEV), torch.empty(B * T, C, device=DEV)
    ops.glu_dwconv_fwd(gd, wd, bd, glud, zd, T)
    _close(zd.view(B, T, C), z, atol=2e-5, rtol=1e-4)
    dg = torch.empty(B * T, 2 * C, device=DEV)
    dw, dbb = torch.zeros(C, K, device=DEV), torch.zeros(C, device=DEV)
    ops.glu_dwconv_bwd(dz.reshape(B * T, C).to(DEV), gd, glud, wd, dg, dw, dbb, T)
    _close(dg.view(B, T, 2 * C), g.grad, atol=2e-5, rtol=1e-4)
    _close(dw, w.grad.view(C, K), atol=2e-4, rtol=1e-4)
    _close(dbb, b.grad, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("B,H,T", [(3, 2, 37), (2, 2, 64), (1, 4, 130)])
def test_relpos_softmax(B, H, T):
    ops = _ops()
    ac = _rand(B, H, T, T, seed=1, scale=3.0).requires_grad_(True)
    bd = _rand(B, H, T, T, seed=2, scale=3.0).requires_grad_(True)
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    if B > 1:
        mask[1, 0, T // 2:] = False
    if B > 2:
        mask[2] = False
    scale = 0.3
    s = (ac + O.rel_shift_legacy(bd)) * scale
    m = mask.unsqueeze(1).eq(0)
    pr = torch.softmax(s.masked_fill(m, float(np.finfo(np.float32).min)), dim=-1).masked_fill(m, 0.0)
    dp = _rand(B, H, T, T, seed=3)
    pr.backward(dp)
    probs = torch.empty(B, H, T, T, device=DEV)
    ops.relpos_softmax_fwd(ac.detach().to(DEV), bd.detach().to(DEV), mask.view(B, T).to(DEV).view(torch.uint8), probs,
                           B, H, T, scale)
    _close(probs, pr, atol=1e-6, rtol=1e-4)
    ds = dp.to(DEV).clone()
    dbd = torch.full((B, H, T, T), 7.0, device=DEV)
    ops.relpos_softmax_bwd(probs, ds, ds, dbd, B, H, T, scale)
    _close(ds, ac.grad, atol=1e-6, rtol=1e-3)
    _close(dbd, bd.grad, atol=1e-6, rtol=1e-3)


def test_loss_and_optimizer():
    ops = _ops()
    M, C = 500, 80
    before, after, y = _rand(M, C, seed=1).requires_grad_(True), _rand(M, C, seed=2).requires_grad_(True), _rand(M, C, seed=3)
    masked = torch.from_numpy(np.random.RandomState(4).rand(M) < 0.8)
    loss = O.mlm_loss(before[None], after[None], y[None], masked[None], O.A3TConfig())
    loss.backward()
    lo = torch.empty(1, device=DEV)
    db, da = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV)
    scratch = torch.empty(ops.loss_scratch_floats(M), device=DEV)
    ops.mlm_loss(before.detach().to(DEV), after.detach().to(DEV), y.to(DEV), masked.to(DEV).view(torch.uint8), lo, db,
                 da, scratch)
    assert abs(float(lo) - float(loss)) < 1e-4 * max(1.0, abs(float(loss)))
    _close(db, before.grad, atol=1e-8, rtol=1e-5)
    _close(da, after.grad, atol=1e-8, rtol=1e-5)
    # clip + Adam on a flat buffer vs the oracle restatement, 3 steps with Noam LR
    n = 100003
    p = _rand(n, seed=5)
    pd, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pr, mr, vr = [p.clone()], [torch.zeros(n)], [torch.zeros(n)]
    partial = torch.zeros(1024, dtype=torch.float64, device=DEV)
    norm = torch.zeros(1, device=DEV)
    for step in range(1, 4):
        g = _rand(n, seed=10 + step, scale=0.01 * step)
        lr = O.noam_lr(step, 1.0, 384, 4000)
        gn = O.clip_adam_step(pr, [g], mr, vr, step, lr, 1.0)
        gd = g.to(DEV)
        ops.sumsq(gd, partial)
        ops.clip_adam(pd, gd, m, v, partial, norm, lr, step, clip=1.0)
        assert abs(float(norm) - float(gn)) < 1e-4 * float(gn)
        _close(pd, pr[0], atol=1e-6, rtol=1e-5)


def test_dropout_statistics_replay_and_gemm_epilogue():
    ops = _ops()
    from a3t_amd._lib import F32
    n = 1 << 20
    x = torch.ones(n, device=DEV)
    y1, y2, y3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    ops.dropout(x, y1, 0.2, 1234)
    ops.dropout(x, y2, 0.2, 1234)
    ops.dropout(x, y3, 0.2, 99)
    assert torch.equal(y1, y2)
    keep = float((y1 > 0).float().mean())
    assert abs(keep - 0.8) < 5e-3
    assert abs(float(y1.mean()) - 1.0) < 1e-2
    assert not torch.equal(y1, y3)
    # no visible structure along rows / columns of a matrix view
    m = (y1.view(1024, 1024) > 0).float()
    assert float(m.mean(0).std()) < 0.03 and float(m.mean(1).std()) < 0.03
    # the GEMM epilogue regenerates exactly the mask of the stand-alone kernel (same key, same linear index)
    M, N, K = 300, 256, 64
    a, W = _rand(M, K, seed=1).to(DEV), _rand(N, K, seed=2).to(DEV)
    ref = torch.empty(M, N, device=DEV)
    ops.linear_fwd(a, W, ref, compute=F32)
    out = torch.empty(M, N, device=DEV)
    ops.linear_fwd(a, W, out, compute=F32, drop=(0.3, 777))
    exp = torch.empty(M, N, device=DEV)
    ops.dropout(ref, exp, 0.3, 777)
    _close(out, exp, atol=1e-6, rtol=1e-6)
    # backward companion: gm = g*mask/(1-p) and its column sums
    g = _rand(M, N, seed=3).to(DEV)
    gm = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    cs = torch.zeros(N, device=DEV)
    ops.dropout_bwd_cast(g, gm, 0.3, 777, colsum=cs, colsum_scale=0.5)
    gexp = torch.empty(M, N, device=DEV)
    ops.dropout(g, gexp, 0.3, 777)
    _close(gm, gexp, atol=2e-2, rtol=1e-2)
    _close(cs, 0.5 * gexp.sum(0), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("M,D", [(1003, 384), (64, 128), (77, 80)])
def test_layernorm_bwd_bf16_input_copy_and_fused_column_sum(M, D):
    ops = _ops()
    x = _rand(M, D, seed=1, scale=2.0).requires_grad_(True)
    g = (1 + 0.1 * _rand(D, seed=2)).requires_grad_(True)
    b = _rand(D, seed=3).requires_grad_(True)
    y = F.layer_norm(x, (D,), g, b, 1e-12)
    dy16 = _rand(M, D, seed=4).bfloat16()
    res = _rand(M, D, seed=5)
    y.backward(dy16.float())
    xd = x.detach().to(DEV)
    yd = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(xd, g.detach().to(DEV), b.detach().to(DEV), yd, mean, rstd, 1e-12)
    _close(yd, y, atol=3e-2, rtol=1e-2)
    dx, dx16 = torch.empty(M, D, device=DEV), torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dg, db, cs = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dy16.to(DEV), xd, g.detach().to(DEV), mean, rstd, res.to(DEV), dx, dg, db, dx16=dx16, dxsum=cs,
                      dxsum_scale=0.5)
    ref = x.grad + res
    _close(dx, ref, atol=5e-5, rtol=1e-4)
    _close(dx16, ref, atol=3e-2, rtol=1e-2)
    _close(dg, g.grad, atol=1e-4 * math.sqrt(M), rtol=1e-4)
    _close(db, b.grad, atol=1e-4 * math.sqrt(M), rtol=1e-4)
    _close(cs, 0.5 * ref.sum(0), atol=2e-4 * math.sqrt(M), rtol=1e-4)


@pytest.mark.parametrize("layout", ["nt", "nn"])
def test_gemm_large_k_many_tiles(layout):
    """A large GEMM (K = 4096, 232 256x256 tiles with a ragged last row tile): the k-contiguous layout is what the
    dispatcher's cost model routes to the persistent 8-phase kernel, the strided-W layout stays on the 128x128 kernel;
    bf16 operands, fp32 accumulate, checked against an fp32 matmul of the same bf16-rounded inputs.
    Transpose-detecting (non-square operands, M != N)."""
    ops = _ops()
    from a3t_amd._lib import BF16
    M, N, K = 3904 + 256 * 13, 2048, 4096          # 7232 rows: 29 row tiles (last one partial: 64 rows), 8 col tiles
    x = (_rand(M, K, seed=1) * 0.5).to(DEV).bfloat16()
    bias = _rand(N, seed=3).to(DEV)
    if layout == "nt":
        W = (_rand(N, K, seed=2) * K ** -0.5).to(DEV).bfloat16()
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.linear_fwd(x, W, out, bias=bias, compute=BF16)
        from a3t_amd import _lib
        assert "8p" in _lib.load().a3t_gemm_last_kernel().decode()
        ref = x.float() @ W.float().t() + bias
    else:
        W = (_rand(K, N, seed=2) * K ** -0.5).to(DEV).bfloat16()   # dx = dy @ W with dy := x, W stored [K][N]
        out = torch.empty(M, N, device=DEV, dtype=torch.float32)
        ops.linear_bwd_data(x, W, out, compute=BF16)
        ref = x.float() @ W.float()
    _close(out, ref, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("T,dk", [(200, 192), (72, 192), (328, 160), (1120, 192)])
def test_gemm_192_column_tile(T, dk):
    """The 128x192-tile variant of the direct-to-LDS GEMM (d_k = 192 attention operands: one column tile instead of
    one and a half).  Batched over (utterance, head) with the engine's strided views, all three layouts, ragged row
    tiles, a ragged column tile (d_k = 160), bf16 and fp32 outputs, fused column sums (the bias-gradient epilogue);
    checked against fp32 matmuls of the same bf16-rounded operands."""
    ops = _ops()
    from a3t_amd import _lib
    from a3t_amd._lib import BF16
    lib = _lib.load()
    B, H = (2, 2) if T > 1000 else (3, 2)
    d = H * dk
    bf = lambda t: t.to(DEV).bfloat16()
    probs = bf(_rand(B, H, T, T, seed=1, scale=T ** -0.5))
    qkv = bf(_rand(B * T, 3 * d, seed=2))
    vv = qkv.view(-1)[2 * d:]
    v4 = qkv.float().view(B, T, 3, H, dk)[:, :, 2].permute(0, 2, 1, 3)            # (B,H,T,dk)
    # NN: ctx[b,:,h,:] = probs[b,h] V[b,h]
    ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
    ops.gemm(probs, vv, ctx, T, dk, T, T, 1, 1, 3 * d, d, batch=B * H, batch_inner=H, a_bs=(H * T * T, T * T),
             b_bs=(T * 3 * d, dk), c_bs=(T * d, dk), compute=BF16)
    assert lib.a3t_gemm_last_kernel().decode().endswith(", 3>"), lib.a3t_gemm_last_kernel()
    ref = torch.matmul(probs.float(), v4).permute(0, 2, 1, 3).reshape(B * T, d)
    _close(ctx, ref, atol=2e-2, rtol=2e-2)
    # TN with fused column sums: dV[b,h] = probs[b,h]^T dctx[b,:,h,:]; colsum over rows and utterances
    dctx = bf(_rand(B * T, d, seed=3))
    dqkv = torch.zeros(B * T, 3 * d, device=DEV, dtype=torch.bfloat16)
    dvv = dqkv.view(-1)[2 * d:]
    cs = torch.zeros(d, device=DEV)
    ops.gemm(probs, dctx, dvv, T, dk, T, 1, T, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=(H * T * T, T * T),
             b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, colsum=cs, colsum_bs1=dk)
    assert lib.a3t_gemm_last_kernel().decode().endswith(", 3>")
    d4 = dctx.float().view(B, T, H, dk).permute(0, 2, 1, 3)
    ref = torch.matmul(probs.float().transpose(2, 3), d4)                          # (B,H,T,dk)
    _close(dqkv.view(B, T, 3, H, dk)[:, :, 2].permute(0, 2, 1, 3), ref, atol=3e-2, rtol=2e-2)
    assert float(dqkv.view(B, T, 3, H, dk)[:, :, :2].abs().max()) == 0.0           # q / k slices untouched
    _close(cs.view(H, dk), ref.sum((0, 2)), atol=0.05 * math.sqrt(B * T), rtol=2e-2)
    # the same with the column sums spread over S accumulator copies (slot = (row tile + utterance) % S) + the fold
    S = 4
    sl = torch.zeros(S, 4 * d, device=DEV)
    gu, gv, gb = torch.ones(d, device=DEV), torch.zeros(d, device=DEV), torch.zeros(3 * d, device=DEV)
    for part in (0, 3):     # pretend the same product is d(q+u) (part 0) and dV (part 3)
        ops.gemm(probs, dctx, dvv, T, dk, T, 1, T, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=(H * T * T, T * T),
                 b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, colsum=sl[0, part * d:], colsum_bs1=dk,
                 colsum_slots=S, colsum_ss=4 * d)
    assert int((sl.abs().sum(1) > 0).sum()) == min(S, (T + 127) // 128 + B - 1)    # slot = (row tile + utterance) % S
    ops.attn_bias_fold(sl, S, d, gu, gv, gb)
    tot = ref.sum((0, 2)).reshape(d)
    tol = dict(atol=0.05 * math.sqrt(B * T), rtol=2e-2)
    _close(gu, 1.0 + tot, **tol)
    _close(gv, torch.zeros(d), **tol)
    _close(gb, torch.cat([tot, torch.zeros(d, device=DEV), tot]), **tol)
    # NT: plain linear with N = d_k outputs, bias + relu, fp32 output with residual
    x = bf(_rand(B * T, 136, seed=4))
    W = bf(_rand(dk, 136, seed=5, scale=136 ** -0.5))
    bias, R = _rand(dk, seed=6).to(DEV), _rand(B * T, dk, seed=7).to(DEV)
    out = torch.empty(B * T, dk, device=DEV)
    ops.linear_fwd(x, W, out, bias=bias, R=R, act=_lib.ACT_RELU, alpha=0.5, compute=BF16)
    assert lib.a3t_gemm_last_kernel().decode().endswith(", 3>")
    _close(out, 0.5 * torch.relu(x.float() @ W.float().t() + bias) + R, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("B,H,T", [(3, 2, 72), (2, 2, 200), (1, 2, 1120), (2, 2, 37)])
def test_relpos_softmax_bf16_scores(B, H, T):
    """bf16 compute mode: ac / bd / dprobs / probs / ds / dbd all stored in bf16 (fp32 math inside).  T % 8 == 0
    takes the 16-byte vector kernels, T = 37 the scalar typed path; both against fp32 torch math on the SAME
    bf16-rounded inputs, with key padding, a fully padded utterance and attention dropout replayed from the saved
    dropped probabilities."""
    ops = _ops()
    bf = lambda t: t.bfloat16().float()
    ac = bf(_rand(B, H, T, T, seed=1, scale=3.0)).requires_grad_(True)
    bd = bf(_rand(B, H, T, T, seed=2, scale=3.0)).requires_grad_(True)
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    if B > 1:
        mask[1, 0, T // 2:] = False
    if B > 2:
        mask[2] = False
    scale, pdrop_p, key = 0.3, 0.25, 12345
    s = (ac + O.rel_shift_legacy(bd)) * scale
    m = mask.unsqueeze(1).eq(0)
    pr = torch.softmax(s.masked_fill(m, float(np.finfo(np.float32).min)), dim=-1).masked_fill(m, 0.0)
    probs = torch.empty(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    pd = torch.empty_like(probs)
    km = mask.view(B, T).to(DEV).view(torch.uint8)
    ops.relpos_softmax_fwd(ac.detach().to(DEV).bfloat16(), bd.detach().to(DEV).bfloat16(), km, probs, B, H, T, scale,
                           probs_drop=pd, drop=(pdrop_p, key))
    _close(probs, pr, atol=4e-3, rtol=1e-2)
    keep = (pd.float() != 0).cpu()
    frac = keep[pr > 1e-3].float().mean()
    assert abs(float(frac) - (1 - pdrop_p)) < 0.05
    _close(pd.float().cpu()[keep], (probs.float().cpu() / (1 - pdrop_p))[keep], atol=4e-3, rtol=1e-2)
    # backward: gradient arrives for the DROPPED probabilities
    dpd = bf(_rand(B, H, T, T, seed=3))
    (pr * keep.float() / (1 - pdrop_p)).backward(dpd)
    ds = torch.empty(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    dbd = torch.full((B, H, T, T), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.relpos_softmax_bwd(probs, dpd.to(DEV).bfloat16(), ds, dbd, B, H, T, scale, probs_drop=pd, drop_p=pdrop_p)
    _close(ds, ac.grad, atol=6e-3, rtol=3e-2)
    _close(dbd, bd.grad, atol=6e-3, rtol=3e-2)
    if T % 8 == 0:
        # round 3: the same gradients with the mask REGENERATED from the counter RNG (no read of the dropped probabilities),
        # into the head-major dBD layout [H][B][T][T] the engine uses
        ds2 = torch.empty_like(ds)
        dbd2 = torch.full((H, B, T, T), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.relpos_softmax_bwd(probs, dpd.to(DEV).bfloat16(), ds2, dbd2, B, H, T, scale, probs_drop=None, drop_p=pdrop_p,
                               dbd_head_major=True, drop_key=key)
        assert torch.equal(ds2, ds) and torch.equal(dbd2.permute(1, 0, 2, 3), dbd)


@pytest.mark.parametrize("B,T,C,K", [(2, 150, 64, 7), (3, 130, 128, 31), (1, 70, 384, 5), (2, 1120, 384, 31)])
def test_glu_dwconv_bf16_vector_path(B, T, C, K):
    """bf16 storage, C % 64 == 0 -> dwconv_vec.hip (16-byte lane accesses).  Reference: fp32 torch math on the same
    bf16-rounded g; the backward feeds the kernel's own bf16 glu (as the engine does)."""
    ops = _ops()
    g = _rand(B, T, 2 * C, seed=1).bfloat16().float().requires_grad_(True)
    w = _rand(C, 1, K, seed=2, scale=K ** -0.5).requires_grad_(True)
    b = _rand(C, seed=3).requires_grad_(True)
    glu = F.glu(g.transpose(1, 2), dim=1)
    z = F.conv1d(glu, w, b, padding=(K - 1) // 2, groups=C).transpose(1, 2)
    dz = _rand(B, T, C, seed=4)
    z.backward(dz)
    gd = g.detach().reshape(B * T, 2 * C).to(DEV).bfloat16()
    wd, bd = w.detach().reshape(C, K).contiguous().to(DEV), b.detach().to(DEV)
    glud = torch.empty(B * T, C, device=DEV, dtype=torch.bfloat16)
    zd = torch.empty(B * T, C, device=DEV)
    ops.glu_dwconv_fwd(gd, wd, bd, glud, zd, T)
    _close(glud.view(B, T, C), glu.transpose(1, 2), atol=1e-2, rtol=1e-2)
    _close(zd.view(B, T, C), z, atol=2e-5, rtol=1e-4)       # the conv reads the un-rounded GLU window
    dg = torch.empty(B * T, 2 * C, device=DEV, dtype=torch.bfloat16)
    dw, dbb, dgs = torch.zeros(C, K, device=DEV), torch.zeros(C, device=DEV), torch.zeros(2 * C, device=DEV)
    ops.glu_dwconv_bwd(dz.reshape(B * T, C).to(DEV), gd, glud, wd, dg, dw, dbb, T, dgsum=dgs)
    _close(dg.view(B, T, 2 * C), g.grad, atol=2e-2, rtol=2e-2)
    scale = float(w.grad.abs().max())
    _close(dw, w.grad.view(C, K), atol=1e-2 * scale, rtol=2e-2)   # glu enters the weight gradient bf16-rounded
    _close(dbb, b.grad, atol=1e-3 * float(b.grad.abs().max()) + 1e-4, rtol=1e-4)
    _close(dgs, g.grad.reshape(-1, 2 * C).sum(0), atol=2e-2 * float(g.grad.abs().sum(dim=(0, 1)).max()) / 10, rtol=5e-2)


def test_layernorm_bwd_fused_consumer_dropout():
    """a3t_layernorm_bwd(drop=(p, key)): dx stays unmasked, dx_bf16 / dx_colsum carry exactly what the separate
    dropout_bwd_cast pass over dx would produce (same counter RNG, same element index)."""
    ops = _ops()
    M, D = 517, 384
    x = _rand(M, D, seed=1).to(DEV)
    g = _rand(D, seed=2).to(DEV)
    b = _rand(D, seed=3).to(DEV)
    y = torch.empty(M, D, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(x, g, b, y, mean, rstd, 1e-12)
    dy = _rand(M, D, seed=4).to(DEV).bfloat16()
    dres = _rand(M, D, seed=5).to(DEV)
    p, key = 0.2, 0xBEEF1234
    dx_a, dx16_a = torch.empty(M, D, device=DEV), torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dg_a, db_a, cs_a = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dres, dx_a, dg_a, db_a, dx16=dx16_a, dxsum=cs_a, dxsum_scale=0.5, drop=(p, key))
    dx_b, dx16_b = torch.empty(M, D, device=DEV), torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dg_b, db_b, cs_b = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dres, dx_b, dg_b, db_b, dx16=dx16_b)
    gm = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    ops.dropout_bwd_cast(dx_b, gm, p, key, colsum=cs_b, colsum_scale=0.5)
    assert torch.equal(dx_a, dx_b)
    assert torch.equal(dx16_a, gm)
    keep = float((gm.float() != 0).float().mean())
    assert abs(keep - (1 - p)) < 0.01
    _close(cs_a, cs_b, atol=1e-3, rtol=1e-4)
    _close(dg_a, dg_b, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_add_pos_bias_fwd_bwd(dt):
    """q + pos_bias_u / q + pos_bias_v out of the fused qkv projection and the backward sum into d(qkv)[:, :d]
    (attention.py:190-199); bf16 takes the 16-byte vector kernels."""
    ops = _ops()
    M, d = 333, 384
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    qkv = _rand(M, 3 * d, seed=1).to(DEV).to(tdt)
    u, v = _rand(d, seed=2).to(DEV), _rand(d, seed=3).to(DEV)
    qu, qv = torch.empty(M, d, device=DEV, dtype=tdt), torch.empty(M, d, device=DEV, dtype=tdt)
    ops.add_pos_bias(qkv, u, v, qu, qv)
    tol = dict(atol=2e-2, rtol=2e-2) if dt == "bf16" else dict(atol=1e-6, rtol=1e-6)
    _close(qu, qkv[:, :d].float() + u, **tol)
    _close(qv, qkv[:, :d].float() + v, **tol)
    dqu, dqv = _rand(M, d, seed=4).to(DEV).to(tdt), _rand(M, d, seed=5).to(DEV).to(tdt)
    dqkv = torch.full((M, 3 * d), 7.0, device=DEV, dtype=tdt)
    ops.add_pos_bias_bwd(dqu, dqv, dqkv)
    _close(dqkv[:, :d], dqu.float() + dqv.float(), **tol)
    assert float((dqkv[:, d:].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("M,C", [(1000, 128), (333, 96)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_col_reduce_modes(M, C, mode):
    ops = _ops()
    x, y = _rand(M, C, seed=1).to(DEV), _rand(M, C, seed=2).to(DEV)
    o0 = torch.zeros(C, dtype=torch.float64, device=DEV)
    o1 = torch.zeros(C, dtype=torch.float64, device=DEV)
    ops.col_reduce(x, o0, o1 if mode else None, y=y if mode == 2 else None, mode=mode)
    _close(o0, x.double().sum(0), atol=1e-4, rtol=1e-6)
    if mode == 1:
        _close(o1, (x.double() ** 2).sum(0), atol=1e-4, rtol=1e-6)
    if mode == 2:
        _close(o1, (x.double() * y.double()).sum(0), atol=1e-4, rtol=1e-6)


@pytest.mark.parametrize("taps,Cin,Cout,T", [(3, 128, 1536, 1030), (5, 64, 1536, 1100), (3, 1536, 128, 6 * 1030)])
def test_conv_large_grid_single_buffer_variants(taps, Cin, Cout, T):
    """Conv1d shapes whose grid is in the single-buffer / 4-workgroups-per-CU regime (>= 768 tiles, the production FFN
    regime; the small shapes above take the double-buffered variants).  T is not a multiple of the 128-row tile, so
    utterance boundaries fall inside tiles.  Forward (NT, fast-conv) and data gradient (NN, fast-conv)."""
    ops = _ops()
    from a3t_amd._lib import BF16, ACT_RELU
    B = 8 if Cout >= 1536 else 16
    T = T if Cout >= 1536 else 1030
    M, pad = B * T, (taps - 1) // 2
    x = (_rand(B, T, Cin, seed=1) * 0.5).bfloat16().float()
    Wt = (_rand(Cout, Cin, taps, seed=2) * (Cin * taps) ** -0.5).bfloat16().float()      # torch layout (out, in, k)
    bias = _rand(Cout, seed=3)
    ref = F.relu(F.conv1d(x.transpose(1, 2), Wt, bias, padding=pad)).transpose(1, 2).reshape(M, Cout)
    Wk = Wt.permute(0, 2, 1).contiguous().to(DEV).bfloat16()                              # [out][tap][in]
    xd = x.reshape(M, Cin).to(DEV).bfloat16()
    out = torch.empty(M, Cout, device=DEV, dtype=torch.bfloat16)
    ops.conv_fwd(xd, Wk, out, T, pad, bias=bias.to(DEV), act=ACT_RELU, compute=BF16)
    _close(out, ref, atol=3e-2, rtol=3e-2)
    # data gradient: dx[m][c] = sum_tap sum_n dy[m - (tap - pad)][n] W[n][tap][c]
    dy = (_rand(B, T, Cout, seed=4) * 0.5).bfloat16().float()
    xr = x.clone().requires_grad_(True)
    F.conv1d(xr.transpose(1, 2), Wt, None, padding=pad).transpose(1, 2).backward(dy)
    dx = torch.empty(M, Cin, device=DEV, dtype=torch.float32)
    ops.conv_bwd_data(dy.reshape(M, Cout).to(DEV).bfloat16(), Wk, dx, T, pad, compute=BF16)
    scale = float(xr.grad.abs().max())
    _close(dx, xr.grad.reshape(M, Cin), atol=2e-2 * scale, rtol=3e-2)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_relpos_softmax_long_rows_generic_three_pass_path(dt):
    """T > 2048 (a > 25 s utterance): rows no longer fit the register-cached kernels, the generic three-pass kernel
    runs (fp32 and bf16-typed), including the shifted-BD indexing and a partially padded utterance."""
    ops = _ops()
    B, H, T = 2, 1, 2056
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    ac = _rand(B, H, T, T, seed=1, scale=2.0).to(tdt).float().requires_grad_(True)
    bd = _rand(B, H, T, T, seed=2, scale=2.0).to(tdt).float().requires_grad_(True)
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    mask[1, 0, T - 300:] = False
    scale = 0.2
    s = (ac + O.rel_shift_legacy(bd)) * scale
    m = mask.unsqueeze(1).eq(0)
    pr = torch.softmax(s.masked_fill(m, float(np.finfo(np.float32).min)), dim=-1).masked_fill(m, 0.0)
    dp = _rand(B, H, T, T, seed=3).to(tdt).float()
    pr.backward(dp)
    probs = torch.empty(B, H, T, T, device=DEV, dtype=tdt)
    ops.relpos_softmax_fwd(ac.detach().to(DEV).to(tdt), bd.detach().to(DEV).to(tdt), mask.view(B, T).to(DEV).view(torch.uint8),
                           probs, B, H, T, scale)
    tol = dict(atol=4e-3, rtol=1e-2) if dt == "bf16" else dict(atol=1e-6, rtol=1e-4)
    _close(probs, pr, **tol)
    ds = torch.empty(B, H, T, T, device=DEV, dtype=tdt)
    dbd = torch.full((B, H, T, T), 7.0, device=DEV, dtype=tdt)
    ops.relpos_softmax_bwd(probs, dp.to(DEV).to(tdt), ds, dbd, B, H, T, scale)
    tol = dict(atol=6e-3, rtol=3e-2) if dt == "bf16" else dict(atol=1e-6, rtol=1e-3)
    _close(ds, ac.grad, **tol)
    _close(dbd, bd.grad, **tol)


@pytest.mark.parametrize("l2", [False, True])
@pytest.mark.parametrize("frac", [0.6, 0.0])
def test_mlm_loss_l1_and_mse_variants(l2, frac):
    """_calc_mlm_loss (sedit_model.py:320-340): L1 (recipe) or MSE (lsm_weight > 50, :105-108) summed over the 80 bins of
    the masked frames / (n_masked + 1e-10), before + after terms; frac = 0: no masked frame -> loss 0, zero gradients."""
    ops = _ops()
    M, C = 515, 80
    before = _rand(M, C, seed=1).requires_grad_(True)
    after = _rand(M, C, seed=2).requires_grad_(True)
    y = _rand(M, C, seed=3)
    masked = torch.from_numpy(np.random.RandomState(4).rand(M) < frac)
    crit = (lambda a, b: (a - b) ** 2) if l2 else (lambda a, b: (a - b).abs())
    w = masked.float()[:, None]
    loss = (crit(before, y) * w).sum() / (masked.float().sum() + 1e-10) + (crit(after, y) * w).sum() / (masked.float().sum() + 1e-10)
    loss.backward()
    lo = torch.empty(1, device=DEV)
    db, da = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV)
    scratch = torch.empty(ops.loss_scratch_floats(M), device=DEV)
    ops.mlm_loss(before.detach().to(DEV), after.detach().to(DEV), y.to(DEV), masked.to(DEV).view(torch.uint8), lo, db, da,
                 scratch, l2=l2, gscale=1.0)
    assert abs(float(lo) - float(loss)) < 1e-4 * max(1.0, abs(float(loss)))
    _close(db, before.grad, atol=1e-7, rtol=1e-4)
    _close(da, after.grad, atol=1e-7, rtol=1e-4)


def test_gemm_dispatch_randomised_sweep():
    """60 random linear / conv problems (ragged M, N, K, taps 3/5, dilation, batch, every epilogue option) through the
    a3t_gemm dispatcher against fp32 torch math: no call may be rejected and every result must agree to bf16 accuracy;
    the sweep must reach the plain / fast-conv / generic bookkeeping variants of all three layouts."""
    import fuzz_gemm as mod
    fails, seen = mod.run(seed=3, n_cases=60, verbose=False)
    assert fails == 0
    glds = [k for k in seen if k.startswith("gemm_bf16_glds_kernel")]
    assert len(glds) >= 8, seen


def test_8phase_gemm_randomised_sweep_and_ab():
    """The persistent 8-phase 256x256 GEMM (gemm_bf16_8p.hip) must agree with fp32 torch math on the random sweep (forced
    on for every problem it accepts) and, bit for bit, with the 128x128 kernel on the model's conv-FFN launches with their
    fused epilogues: bias + relu + dropout (+ keep bits out); the data gradient through the transposed weight shadow with the
    keep-bit mask and fused column sums against the strided-W / S = h formulation; M / N tails; fp32 output."""
    import fuzz_gemm as mod
    from a3t_amd import _lib
    from a3t_amd._lib import ACT_RELU, BF16
    ops = _ops()
    lib = _lib.load()
    old = lib.a3t_gemm_8p_mode(1)
    try:
        fails, seen = mod.run(seed=11, n_cases=60, verbose=False)
        assert fails == 0
        assert sum(v for k, v in seen.items() if k.startswith("gemm_bf16_8p_kernel")) >= 5, seen
    finally:
        lib.a3t_gemm_8p_mode(old)
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
    B, T, d, ff = 9, 1120, 384, 1536          # 40 row tiles x 6 column tiles = 240 tiles: one partial round on 256 CUs
    M = B * T
    y = rn(M, d).bfloat16()
    W1, W2 = rn(ff, 3, d, sc=0.03).bfloat16(), rn(d, 3, ff, sc=0.02).bfloat16()
    W2t = W2.permute(2, 1, 0).flip(1).contiguous()
    b1 = rn(ff)
    ga = rn(M, d).bfloat16()
    Wl, bl = rn(544, d, sc=0.05).bfloat16(), rn(544)
    Mt = M - 8

    def run_all(keepbits):
        outs = []
        h = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
        keep = torch.zeros(ops.gemm_keep_bytes(M, ff), dtype=torch.uint8, device=DEV) if keepbits else None
        ops.conv_fwd(y, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.2, 777), keep_out=keep)
        outs.append((lib.a3t_gemm_last_kernel().decode(), h))
        dh, gb = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16), torch.zeros(ff, device=DEV)
        if keepbits:
            ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.625, compute=BF16, keep_in=keep, colsum=gb)
        else:
            ops.conv_bwd_data(ga, W2, dh, T, 1, S=h, alpha=0.625, compute=BF16, colsum=gb)
        outs.append((lib.a3t_gemm_last_kernel().decode(), dh))
        outs.append(("colsum", gb))
        o = torch.empty(Mt, 544, device=DEV)
        ops.linear_fwd(y[:Mt], Wl, o, bias=bl, act=ACT_RELU, alpha=0.7, compute=BF16)
        outs.append((lib.a3t_gemm_last_kernel().decode(), o))
        torch.cuda.synchronize()
        return outs

    lib.a3t_gemm_8p_mode(0)
    try:
        ref = run_all(False)
        lib.a3t_gemm_8p_mode(1)
        got = run_all(True)
    finally:
        lib.a3t_gemm_8p_mode(old)
    assert sum("8p" in k for k, _ in got) == 3 and not any("8p" in k for k, _ in ref), [k for k, _ in got]
    for (kr, a), (kg, b) in zip(ref, got):
        if kr == "colsum":      # fp32 atomics in a different order
            assert float((a - b).abs().max() / (a.abs().max() + 1e-9)) < 1e-5
        else:                   # same products in the same k order: identical results
            assert torch.equal(a, b), (kr, kg, float((a.float() - b.float()).abs().max()))
    # keep bits without the 8-phase kernel must be refused, not ignored
    lib.a3t_gemm_8p_mode(0)
    try:
        with pytest.raises(_lib.A3TLibraryError):
            ops.conv_fwd(y, W1, torch.empty(M, ff, device=DEV, dtype=torch.bfloat16), T, 1, compute=BF16,
                         keep_out=torch.zeros(ops.gemm_keep_bytes(M, ff), dtype=torch.uint8, device=DEV))
    finally:
        lib.a3t_gemm_8p_mode(old)


def test_8phase_tn_weight_gradients():
    """The reduction-strided (TN) variant of the 8-phase GEMM: fused Conv1d weight gradients (output columns = (tap, c), token
    shift per 128-column half, zeros across utterance boundaries) and Linear weight gradients, token counts that are NOT a
    multiple of the 64-token K-tile, M / N tails, against fp32 torch math and the 128x128 kernel."""
    from a3t_amd import _lib
    from a3t_amd._lib import BF16
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(3)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    old = lib.a3t_gemm_8p_mode(1)
    try:
        for (B, T, cin, cout, taps) in [(3, 200, 128, 512, 3), (5, 1120, 384, 1536, 3), (4, 264, 256, 264, 1), (2, 1000, 384, 1152, 1)]:
            M = B * T
            dy, x = rn(M, cout).bfloat16(), rn(M, cin).bfloat16()
            outs = []
            for mode in (0, 1):
                lib.a3t_gemm_8p_mode(mode)
                if taps > 1:
                    dW = torch.zeros(cout, taps, cin, device=DEV)
                    ops.conv_bwd_weight(dy, x, dW, T, 1, alpha=0.5, compute=BF16)
                else:
                    dW = torch.zeros(cout, cin, device=DEV)
                    ops.linear_bwd_weight(dy, x, dW, alpha=0.5, compute=BF16)
                outs.append((dW, lib.a3t_gemm_last_kernel().decode()))
            torch.cuda.synchronize()
            assert "8p_tn" in outs[1][1] and "8p" not in outs[0][1], outs[1][1]
            if taps > 1:
                xs, dyf = x.float().view(B, T, cin), dy.float().view(B, T, cout)
                ref = torch.zeros(cout, taps, cin, device=DEV)
                for t in range(taps):
                    xsft = torch.zeros_like(xs)
                    sh = t - 1
                    if sh < 0:
                        xsft[:, -sh:] = xs[:, :sh]
                    elif sh > 0:
                        xsft[:, :-sh] = xs[:, sh:]
                    else:
                        xsft = xs
                    ref[:, t, :] = 0.5 * torch.einsum("btn,btc->nc", dyf, xsft)
            else:
                ref = 0.5 * dy.float().t() @ x.float()
            for dW, _ in outs:
                assert float((dW - ref).abs().max() / ref.abs().max()) < 1e-4
    finally:
        lib.a3t_gemm_8p_mode(old)


def test_row_kernels_randomised_sweep():
    """75 random problems for the rel-pos softmax (T from 1 to 320 incl. T % 8 != 0, fully / partially padded
    utterances, fp32 and bf16 storage), LayerNorm (any M, D incl. the 16-byte vector widths) and GLU + depthwise conv
    (K in 3..31, any C, bf16 vector path and generic path), forward and backward against torch math."""
    import fuzz_rowkernels as mod
    assert mod.run(seed=5, n=25, verbose=False) == 0


def test_dropout_masks_of_two_keys_are_not_shifted_copies_of_each_other():
    """Counter RNG (csrc/dtype_io.h rng_pair).  With the key entering ADDITIVELY (round 3) the masks of two dropout sites were
    the same sequence read at an offset: keys k and k + 7 gave masks that agree 100 % at an index shift of 14.  Round 4: the key
    enters by XOR and again as the increment.  Checked here: (a) for adjacent raw keys no index shift reproduces the other mask
    (agreement stays far from 1), (b) for keys as the engine derives them (murmur-mixed hash of step seed and site, engine._drop)
    the agreement at every small shift sits at the chance level p^2 + (1-p)^2, and the keep rate at 1 - p."""
    import zlib
    from a3t_amd import ops
    n, p = 1 << 20, 0.2
    x = torch.ones(n, device=DEV)

    def mask(key):
        y = torch.empty_like(x)
        ops.dropout(x, y, p, key)
        m = (y != 0)
        assert abs(float(m.float().mean()) - (1 - p)) < 3e-3
        return m

    def engine_key(step, tag):
        h = (zlib.crc32(tag.encode()) ^ ((step * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF
        h = ((h ^ (h >> 16)) * 0x85EBCA6B) & 0xFFFFFFFF
        h = ((h ^ (h >> 13)) * 0xC2B2AE35) & 0xFFFFFFFF
        return (h ^ (h >> 16)) & 0xFFFFFFFF

    chance = p * p + (1 - p) * (1 - p)
    shifts = list(range(0, 65)) + [1 << 10]

    def worst_dev(ma, mb):
        w, top = 0.0, 0.0
        for sh in shifts:
            m = n - sh
            for u, v in ((ma, mb), (mb, ma)):
                agree = float((u[:m] == v[sh:sh + m]).float().mean())
                w, top = max(w, abs(agree - chance)), max(top, agree)
        return w, top

    a, b = mask(1000), mask(1007)                       # (a) raw adjacent keys: no shifted copy
    w, top = worst_dev(a, b)
    print(f"raw keys 1000 / 1007: largest agreement at any shift {top:.4f} (chance {chance:.4f})")
    assert top < chance + 0.05
    ks = [engine_key(3, "enc.0.mha.att"), engine_key(3, "enc.1.mha.att"), engine_key(4, "enc.0.mha.att")]
    ms = [mask(k) for k in ks]                          # (b) production keys: chance level
    worst = max(worst_dev(ms[i], ms[j])[0] for i, j in ((0, 1), (0, 2), (1, 2)))
    print(f"engine keys: largest deviation of mask agreement from chance: {worst:.4f}")
    assert worst < 5e-3


def test_tn3_weight_gradient_kernel_and_its_deterministic_fold():
    """Round 5: the 128 x 384-tile token-reduction kernel (gemm_bf16_8p_tn3_kernel) with split-K partials folded in a fixed order
    (no atomics).  Every weight-gradient shape family of the model -- fused Conv1d gradients whose 384-column tile lies in one tap
    (cin = 384, 1536), spans taps (cin = 128, 512) or is cut by N (cin = 128 x 3 taps = one tile; N = 256, 640), row tails
    (cout = 136, 200, 264), token counts that are no multiple of the 64-token K-tile or of the utterance length -- against fp32
    torch math and the 128 x 128 kernel; run twice: bit-identical (the atomics of the kernels it replaces are not); store /
    add / atomic accumulation modes."""
    from a3t_amd import _lib
    from a3t_amd._lib import ACC_ATOMIC, BF16
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(5)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    old8, old3 = lib.a3t_gemm_8p_mode(1), lib.a3t_gemm_tn3_mode(1)
    try:
        for (B, T, cin, cout, taps) in [(3, 200, 128, 512, 3), (5, 1120, 384, 1536, 3), (5, 1120, 1536, 384, 3), (4, 264, 256, 264, 1),
                                        (2, 1800, 512, 2048, 3), (7, 333, 384, 384, 1), (3, 77, 128, 136, 3), (9, 1120, 640, 200, 1)]:
            M = B * T
            dy, x = rn(M, cout).bfloat16(), rn(M, cin).bfloat16()

            def run(mode8, base=None):
                lib.a3t_gemm_8p_mode(mode8)
                dW = (torch.zeros(cout, taps, cin, device=DEV) if taps > 1 else torch.zeros(cout, cin, device=DEV)) if base is None else base.clone()
                if taps > 1:
                    ops.conv_bwd_weight(dy, x, dW, T, 1, alpha=0.5, compute=BF16)
                else:
                    ops.linear_bwd_weight(dy, x, dW, alpha=0.5, compute=BF16)
                return dW, lib.a3t_gemm_last_kernel().decode()
            d128, k128 = run(0)
            d3, k3 = run(1)
            d3b, _ = run(1)
            base = rn(*d3.shape)
            d3acc, _ = run(1, base)
            torch.cuda.synchronize()
            assert "8p_tn3" in k3 and "8p" not in k128, (k3, k128)
            assert torch.equal(d3, d3b), "the fold sums the K splits in a fixed order"
            if taps > 1:
                xs, dyf = x.float().view(B, T, cin), dy.float().view(B, T, cout)
                ref = torch.zeros(cout, taps, cin, device=DEV)
                for t in range(taps):
                    xsft = torch.zeros_like(xs)
                    sh = t - 1
                    if sh < 0:
                        xsft[:, -sh:] = xs[:, :sh]
                    elif sh > 0:
                        xsft[:, :-sh] = xs[:, sh:]
                    else:
                        xsft = xs
                    ref[:, t, :] = 0.5 * torch.einsum("btn,btc->nc", dyf, xsft)
            else:
                ref = 0.5 * dy.float().t() @ x.float()
            sc = float(ref.abs().max())
            assert float((d3 - ref).abs().max()) / sc < 1e-4 and float((d128 - ref).abs().max()) / sc < 1e-4
            assert float((d3acc - (base + ref)).abs().max()) / sc < 1e-4        # accumulates onto what the buffer holds
    finally:
        lib.a3t_gemm_8p_mode(old8)
        lib.a3t_gemm_tn3_mode(old3)


def test_grouped_linear_weight_gradients_equal_the_single_launches():
    """a3t_gemm_tn3_group: the Linear weight gradients of a Conformer block (3 + 9 + 3 + 6 tiles of 128 x 384) in ONE launch of
    the token-reduction kernel.  Same partial sums, same fold order as the single launches of the same kernel -> bit-identical to
    them; against fp32 torch math 1e-4; accumulates onto what the buffers hold; a member outside the kernel's contract (fp32
    operands) makes the library decline and the wrapper fall back to single launches."""
    from a3t_amd import _lib
    from a3t_amd._lib import BF16
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(6)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    old8, old3 = lib.a3t_gemm_8p_mode(1), lib.a3t_gemm_tn3_mode(1)
    try:
        for M, couts in ((5 * 1120, (384, 1152, 384, 768)), (3 * 200, (136, 384)), (32 * 1120, (384, 1152, 384, 768))):
            items = [(rn(M, co).bfloat16(), rn(M, 384).bfloat16(), rn(co, 384), 1.0 if i % 2 else 0.5) for i, co in enumerate(couts)]
            base = [dW.clone() for _, _, dW, _ in items]
            single = []
            for (dy, x, _, al), b in zip(items, base):
                dW = b.clone()
                ops.linear_bwd_weight(dy, x, dW, alpha=al, compute=BF16)
                assert "tn3" in lib.a3t_gemm_last_kernel().decode()
                single.append(dW)
            assert ops.linear_bwd_weight_group(items, compute=BF16) is True
            torch.cuda.synchronize()
            for (dy, x, dW, al), b, sg in zip(items, base, single):
                ref = b + al * (dy.float().t() @ x.float())
                assert float((dW - ref).abs().max() / ref.abs().max()) < 1e-4
                assert torch.equal(dW, sg) or float((dW - sg).abs().max() / ref.abs().max()) < 2e-6   # (other K split count -> other rounding)
        # a member the kernel does not take (132 output channels: not a multiple of 8): the wrapper reports the fall-back to single
        # launches and the results are still right
        its = [(rn(640, 384).bfloat16(), rn(640, 384).bfloat16(), torch.zeros(384, 384, device=DEV), 1.0),
               (rn(640, 132).bfloat16(), rn(640, 384).bfloat16(), torch.zeros(132, 384, device=DEV), 1.0)]
        assert ops.linear_bwd_weight_group(its, compute=BF16) is False
        torch.cuda.synchronize()
        for dy, x, dW, _ in its:
            ref = dy.float().t() @ x.float()
            assert float((dW - ref).abs().max() / ref.abs().max()) < 1e-4
    finally:
        lib.a3t_gemm_8p_mode(old8)
        lib.a3t_gemm_tn3_mode(old3)


def test_bf16_output_accumulates_in_fp32_with_one_rounding_and_keeps_its_own_column_sum():
    """Round 6: A3T_ACC_ADD on a bf16 C (the second attention product adds d(q+v) onto d(q+u) in the q third of dqkv): the sum is
    formed in fp32 from the stored bf16 value and rounded once; the fused column sum takes the INCREMENT alone (the bias gradient
    of pos_bias_v is the column sum of d(q+v), not of the total); bf16 atomics / split-K accumulation are refused."""
    from a3t_amd import _lib
    from a3t_amd._lib import ACC_ADD, ACC_ATOMIC, BF16
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(21)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    B, H, T, dk = 3, 2, 264, 192
    d = H * dk
    A1, A2 = rn(B, H, T, T).bfloat16(), rn(B, H, T, T).bfloat16()
    kk, P = rn(B * T, d).bfloat16(), rn(T, d).bfloat16()
    out = torch.zeros(B * T, 3 * d, device=DEV, dtype=torch.bfloat16)          # q third of a q | k | v row
    cs = torch.zeros(2 * d, device=DEV)
    zb = (H * T * T, T * T)
    ops.gemm(A1, kk, out, T, dk, T, T, 1, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * d, dk), c_bs=(T * 3 * d, dk),
             compute=BF16, colsum=cs[:d], colsum_bs1=dk)
    first = out[:, :d].clone()
    ops.gemm(A2, P, out, T, dk, T, T, 1, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(0, dk), c_bs=(T * 3 * d, dk),
             acc=ACC_ADD, compute=BF16, colsum=cs[d:], colsum_bs1=dk)
    torch.cuda.synchronize()
    r1 = torch.einsum("bhij,bjhc->bihc", A1.float(), kk.float().view(B, T, H, dk)).reshape(B * T, d)
    r2 = torch.einsum("bhij,jhc->bihc", A2.float(), P.float().view(T, H, dk)).reshape(B * T, d)
    assert torch.equal(first, r1.bfloat16()) or float((first.float() - r1).abs().max()) < 2e-2 * float(r1.abs().max())
    want = (first.float() + r2).bfloat16()                                    # one rounding of (stored bf16 + fp32 increment)
    diff = (out[:, :d].float() - want.float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(want.float().abs().max()), float(diff.max())      # <= 1 bf16 ulp (fp32 sum order)
    assert float((out[:, d:].float()).abs().max()) == 0.0                     # the k | v thirds are not touched
    sc = float(r2.abs().sum(0).max())
    assert float((cs[d:] - r2.sum(0)).abs().max()) < 2e-3 * sc and float((cs[:d] - r1.sum(0)).abs().max()) < 2e-3 * float(r1.abs().sum(0).max())
    with pytest.raises(_lib.A3TLibraryError if hasattr(_lib, "A3TLibraryError") else RuntimeError):
        ops.gemm(A2, P, out, T, dk, T, T, 1, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(0, dk), c_bs=(T * 3 * d, dk),
                 acc=ACC_ATOMIC, compute=BF16)


def test_release_workspaces_frees_the_split_k_slabs_and_the_next_launch_allocates_again():
    """a3t_release_workspaces (round 6): the grow-only slabs of the weight-gradient kernels and the attention key-split
    workspace are freed; the next weight gradient allocates again and gives the same bits."""
    from a3t_amd import _lib
    from a3t_amd._lib import BF16
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(8)
    dy, x = torch.randn(5 * 1120, 1536, device=DEV, generator=g).bfloat16(), torch.randn(5 * 1120, 384, device=DEV, generator=g).bfloat16()

    def run():
        dW = torch.zeros(1536, 3, 384, device=DEV)
        ops.conv_bwd_weight(dy, x, dW, 1120, 1, compute=BF16)
        torch.cuda.synchronize()
        return dW
    a = run()
    free0 = torch.cuda.mem_get_info()[0]
    assert lib.a3t_release_workspaces() == 0
    free1 = torch.cuda.mem_get_info()[0]
    b = run()
    assert free1 >= free0 and torch.equal(a, b)
    assert lib.a3t_release_workspaces() == 0 and lib.a3t_release_workspaces() == 0        # idempotent


@pytest.mark.parametrize("B,H,T,dk", [(2, 2, 328, 192), (3, 4, 520, 128), (1, 2, 1120, 192), (2, 1, 264, 96), (1, 3, 648, 160)])
def test_streaming_attention_backward_gemm_matches_the_128_row_kernel_and_the_fp32_product(B, H, T, dk):
    """gemm_bf16_tt.hip (round 6): the score-sized products of the attention backward (espnet attention.py:64-96,145-209) --
    dS K and dBD P ([m][k] operand, the second one added to the stored bf16 in fp32), P^T dctx / dS^T (q+u) ([k][m] operand) --
    on the one-workgroup-per-CU streaming kernel: same bf16 bits as the 128-row kernel (both sum the K-tiles in ascending order
    in fp32), column sums equal to fp32 accumulation noise, and both within bf16 rounding of the fp32 product.  Shapes cover
    K tails (T % 32 = 8), row tiles that end inside a tile, N = 96 / 128 / 160 / 192 and a batch-shared B operand."""
    from a3t_amd import _lib
    from a3t_amd._lib import ACC_ADD, ACC_STORE, BF16
    ops = _ops()
    lib = _lib.load()
    d, M = H * dk, B * T
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + T + dk)
    S = (torch.randn(B, H, T, T, device=DEV, generator=g) * 0.1).bfloat16()
    qkv = torch.randn(M, 3 * d, device=DEV, generator=g).bfloat16()
    x = torch.randn(M, d, device=DEV, generator=g).bfloat16()
    P = torch.randn(T, d, device=DEV, generator=g).bfloat16()
    zb = (H * T * T, T * T)
    NS = 4
    Sf = S.float()

    def products(mode):
        old = lib.a3t_gemm_tt_mode(mode)
        try:
            sl = torch.zeros(NS * 4 * d, device=DEV)
            csk = dict(colsum_bs1=dk, colsum_slots=NS, colsum_ss=4 * d)
            out = torch.full((M, 3 * d), 0.25, device=DEV).bfloat16()
            names = []
            ops.gemm(S, qkv.view(-1)[d:], out, T, dk, T, T, 1, 1, 3 * d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb,
                     b_bs=(T * 3 * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, colsum=sl, **csk)
            names.append(lib.a3t_gemm_last_kernel().decode())
            ops.gemm(S, P, out, T, dk, T, T, 1, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(0, dk),
                     c_bs=(T * 3 * d, dk), acc=ACC_ADD, alpha=0.5, compute=BF16, colsum=sl[d:], **csk)
            names.append(lib.a3t_gemm_last_kernel().decode())
            ops.gemm(S, x, out.view(-1)[2 * d:], T, dk, T, 1, T, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb,
                     b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, colsum=sl[3 * d:], **csk)
            names.append(lib.a3t_gemm_last_kernel().decode())
            torch.cuda.synchronize()
            return out, sl.view(NS, 4 * d).sum(0), names
        finally:
            lib.a3t_gemm_tt_mode(old)

    o0, s0, n0 = products(0)
    o1, s1, n1 = products(1)
    assert all("gemm_bf16_tt_kernel" not in n for n in n0) and all("gemm_bf16_tt_kernel" in n for n in n1), (n0, n1)
    assert n1[0].startswith("gemm_bf16_tt_kernel<false") and n1[2].startswith("gemm_bf16_tt_kernel<true")
    assert torch.equal(o0, o1)
    assert torch.allclose(s0, s1, rtol=1e-4, atol=1e-3 * float(s0.abs().max()))
    # fp32 products of the same bf16 operands
    K_ = qkv.float().view(B, T, 3, H, dk)[:, :, 1].permute(0, 2, 1, 3)            # [B][H][T][dk]
    X_ = x.float().view(B, T, H, dk).permute(0, 2, 1, 3)
    P_ = P.float().view(T, H, dk).permute(1, 0, 2)
    q1 = torch.matmul(Sf, K_)
    q2 = 0.5 * torch.matmul(Sf, P_.unsqueeze(0))
    v = torch.matmul(Sf.transpose(2, 3), X_)
    got = o1.float().view(B, T, 3, H, dk)
    ref_q = (q1.bfloat16().float() + q2).permute(0, 2, 1, 3)       # the first product is stored in bf16, the second added in fp32
    ref_v = v.permute(0, 2, 1, 3)
    assert (got[:, :, 0] - ref_q).abs().max() <= 2e-2 * float(ref_q.abs().max())
    assert (got[:, :, 2] - ref_v).abs().max() <= 2e-2 * float(ref_v.abs().max())
    assert torch.equal(got[:, :, 1], torch.full_like(got[:, :, 1], 0.25))         # the k third was nobody's output
    ref_cs = torch.cat([q1.sum(dim=(0, 2)).reshape(-1), q2.sum(dim=(0, 2)).reshape(-1)])      # sums of the increments, [h][n]
    assert torch.allclose(s1[:2 * d], ref_cs, rtol=2e-2, atol=2e-2 * float(ref_cs.abs().max()))


@pytest.mark.parametrize("B,H,T,dk", [(32, 2, 1120, 192), (16, 4, 1120, 128), (3, 2, 616, 96), (2, 2, 328, 160)])
def test_two_products_in_one_streaming_launch(B, H, T, dk):
    """a3t_gemm_desc::A2 (round 6): dq = dS K + dBD P (attention.py:190-203 on its way back) as ONE launch of the streaming kernel --
    both K loops into one accumulator set, one bf16 rounding, the column sums of the two products taken apart at the hand-over.
    Against the two-launch form (second product added in its epilogue: two roundings) and the fp32 products of the same operands;
    the k third and the v third of the output row are nobody's.  Anything the streaming kernel does not take is refused."""
    from a3t_amd import _lib
    from a3t_amd._lib import ACC_ADD, BF16
    ops = _ops()
    lib = _lib.load()
    d, M = H * dk, B * T
    g = torch.Generator(device=DEV).manual_seed(B * 77 + T + dk)
    S1 = (torch.randn(B, H, T, T, device=DEV, generator=g) * 0.1).bfloat16()
    S2 = (torch.randn(B, H, T, T, device=DEV, generator=g) * 0.1).bfloat16()
    qkv = torch.randn(M, 3 * d, device=DEV, generator=g).bfloat16()
    P = torch.randn(T, d, device=DEV, generator=g).bfloat16()
    zb = (H * T * T, T * T)
    NS = 4
    csk = dict(colsum_bs1=dk, colsum_slots=NS, colsum_ss=4 * d)
    old = lib.a3t_gemm_tt_mode(1)
    try:
        assert ops.gemm_tt_supported(T, dk, T, B * H)
        o2 = torch.full((M, 3 * d), 0.25, device=DEV).bfloat16()
        s2 = torch.zeros(NS * 4 * d, device=DEV)
        ops.gemm(S1, qkv.view(-1)[d:], o2, T, dk, T, T, 1, 1, 3 * d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * 3 * d, dk),
                 c_bs=(T * 3 * d, dk), compute=BF16, colsum=s2, **csk)
        ops.gemm(S2, P, o2, T, dk, T, T, 1, 1, d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(0, dk), c_bs=(T * 3 * d, dk),
                 acc=ACC_ADD, compute=BF16, colsum=s2[d:], **csk)
        o1 = torch.full((M, 3 * d), 0.25, device=DEV).bfloat16()
        s1 = torch.zeros(NS * 4 * d, device=DEV)
        ops.gemm(S1, qkv.view(-1)[d:], o1, T, dk, T, T, 1, 1, 3 * d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * 3 * d, dk),
                 c_bs=(T * 3 * d, dk), compute=BF16, colsum=s1, second=(S2, P, d, (0, dk), s1[d:]), **csk)
        torch.cuda.synchronize()
        name = lib.a3t_gemm_last_kernel().decode()
        assert name.startswith("gemm_bf16_tt_kernel<false") and name.endswith("true>"), name
    finally:
        lib.a3t_gemm_tt_mode(old)
    K_ = qkv.float().view(B, T, 3, H, dk)[:, :, 1].permute(0, 2, 1, 3)
    P_ = P.float().view(T, H, dk).permute(1, 0, 2).unsqueeze(0)
    q1, q2 = torch.matmul(S1.float(), K_), torch.matmul(S2.float(), P_)
    ref = (q1 + q2).permute(0, 2, 1, 3)
    got1, got2 = o1.float().view(B, T, 3, H, dk), o2.float().view(B, T, 3, H, dk)
    sc = float(ref.abs().max())
    e1, e2 = float((got1[:, :, 0] - ref).abs().max()) / sc, float((got2[:, :, 0] - ref).abs().max()) / sc
    assert e1 <= 6e-3 and e1 <= e2 * 1.05 + 1e-4, (e1, e2)         # one rounding instead of two
    assert torch.equal(got1[:, :, 1:], torch.full_like(got1[:, :, 1:], 0.25))
    c1, c2 = s1.view(NS, 4 * d).sum(0), s2.view(NS, 4 * d).sum(0)
    ref_cs = torch.cat([q1.sum(dim=(0, 2)).reshape(-1), q2.sum(dim=(0, 2)).reshape(-1)])
    tol = 2e-3 * float(ref_cs.abs().max()) + 1e-3
    assert torch.allclose(c1[:2 * d], ref_cs, rtol=0, atol=tol), float((c1[:2 * d] - ref_cs).abs().max())
    assert torch.allclose(c1[:2 * d], c2[:2 * d], rtol=0, atol=2e-2 * float(ref_cs.abs().max()))
    assert float(c1[2 * d:].abs().max()) == 0.0
    # refused: the [k][m] operand, an accumulating output, a shape the streaming kernel leaves to the 128-row kernel
    with pytest.raises(Exception):
        ops.gemm(S1, qkv.view(-1)[d:], o1, T, dk, T, 1, T, 1, 3 * d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * 3 * d, dk),
                 c_bs=(T * 3 * d, dk), compute=BF16, second=(S2, P, d, (0, dk), None))
    with pytest.raises(Exception):
        ops.gemm(S1, qkv.view(-1)[d:], o1, T, dk, T, T, 1, 1, 3 * d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * 3 * d, dk),
                 c_bs=(T * 3 * d, dk), acc=ACC_ADD, compute=BF16, second=(S2, P, d, (0, dk), None))
    old = lib.a3t_gemm_tt_mode(0)
    try:
        assert not ops.gemm_tt_supported(T, dk, T, B * H)
        with pytest.raises(Exception):
            ops.gemm(S1, qkv.view(-1)[d:], o1, T, dk, T, T, 1, 1, 3 * d, 3 * d, batch=B * H, batch_inner=H, a_bs=zb,
                     b_bs=(T * 3 * d, dk), c_bs=(T * 3 * d, dk), compute=BF16, second=(S2, P, d, (0, dk), None))
    finally:
        lib.a3t_gemm_tt_mode(old)


def test_streaming_attention_backward_gemm_cost_model_and_unsupported_epilogues_fall_back():
    """Default mode: the streaming kernel takes the configs[1] attention shape (256 workgroups = one round) and leaves a grid that
    fills a third of the chip, a short reduction and every epilogue it does not implement (bias, fp32 output) to the 128-row kernel."""
    from a3t_amd import _lib
    from a3t_amd._lib import BF16
    ops = _ops()
    lib = _lib.load()
    old = lib.a3t_gemm_tt_mode(2)
    try:
        def kernel(B, H, T, dk, out_dtype=torch.bfloat16, bias=False):
            d, M = H * dk, B * T
            S = torch.zeros(B, H, T, T, device=DEV).bfloat16()
            x = torch.zeros(M, d, device=DEV).bfloat16()
            out = torch.zeros(M, d, device=DEV, dtype=out_dtype)
            ops.gemm(S, x, out, T, dk, T, T, 1, 1, d, d, batch=B * H, batch_inner=H, a_bs=(H * T * T, T * T), b_bs=(T * d, dk),
                     c_bs=(T * d, dk), compute=BF16, bias=torch.zeros(dk, device=DEV) if bias else None)
            torch.cuda.synchronize()
            return lib.a3t_gemm_last_kernel().decode()
        assert kernel(32, 2, 1120, 192).startswith("gemm_bf16_tt_kernel<false, 3,")
        assert kernel(16, 4, 1120, 128).startswith("gemm_bf16_tt_kernel<false, 2,")
        assert "tt_kernel" not in kernel(4, 2, 1120, 192)          # 32 workgroups
        assert "tt_kernel" not in kernel(16, 4, 1800, 128)         # 384 workgroups = 1.5 rounds (configs[3])
        assert "tt_kernel" not in kernel(32, 2, 384, 192)          # K = 384
        assert "tt_kernel" not in kernel(32, 2, 1120, 192, out_dtype=torch.float32)
        assert "tt_kernel" not in kernel(32, 2, 1120, 192, bias=True)
    finally:
        lib.a3t_gemm_tt_mode(old)
