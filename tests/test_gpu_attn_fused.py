"""Fused legacy rel-pos attention (a3t_attn_fwd / a3t_attn_bwd) against the materialised path (batched GEMMs +
a3t_relpos_softmax_*, itself pinned by block384.npz / the oracle) on the same bf16 operands, and against the oracle's
fp32 attention on the reference's own block fixture."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import a3t_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def _inputs(B, H, T, dk, seed, lengths=None):
    rs = np.random.RandomState(seed)
    d = H * dk
    M = B * T
    mk = lambda *s, sc=1.0: torch.from_numpy((rs.standard_normal(s) * sc).astype(np.float32)).to(DEV).bfloat16()
    qkv = mk(M, 3 * d)
    qu = (qkv[:, :d].float() + mk(1, d, sc=0.3).float()).bfloat16().contiguous()
    qv = (qkv[:, :d].float() + mk(1, d, sc=0.3).float()).bfloat16().contiguous()
    P = mk(T, d)
    keymask = torch.ones(B, T, dtype=torch.uint8, device=DEV)
    if lengths is not None:
        for b, n in enumerate(lengths):
            keymask[b, n:] = 0
    return qkv, qu, qv, P, keymask


def _materialised(qkv, qu, qv, P, keymask, B, H, T, dk, drop):
    """The round-1 path: ac / bd batched GEMMs -> relpos softmax (bf16 logits, fp32 math) -> probs @ V."""
    from a3t_amd import ops
    from a3t_amd._lib import BF16
    d = H * dk
    M = B * T
    ac = torch.empty(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    bd = torch.empty_like(ac)
    kk = qkv.view(-1)[d:]
    vv = qkv.view(-1)[2 * d:]
    ops.gemm(qu, kk, ac, T, T, dk, d, 1, 3 * d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk), b_bs=(T * 3 * d, dk),
             c_bs=(H * T * T, T * T), compute=BF16)
    ops.gemm(qv, P, bd, T, T, dk, d, 1, d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk), b_bs=(0, dk),
             c_bs=(H * T * T, T * T), compute=BF16)
    probs = torch.empty_like(ac)
    pdrop = torch.empty_like(ac) if drop[0] > 0 else None
    ops.relpos_softmax_fwd(ac, bd, keymask, probs, B, H, T, 1.0 / math.sqrt(dk), probs_drop=pdrop, drop=drop)
    ctx = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    ops.gemm(pdrop if pdrop is not None else probs, vv, ctx, T, dk, T, T, 1, 1, 3 * d, d, batch=B * H, batch_inner=H,
             a_bs=(H * T * T, T * T), b_bs=(T * 3 * d, dk), c_bs=(T * d, dk), compute=BF16)
    return ctx, probs, pdrop


def _exact(qkv, qu, qv, P, keymask, B, H, T, dk):
    """fp64 evaluation of the reference formula on the SAME bf16 operands (rel_shift via the oracle's restatement of
    attention.py:145-165)."""
    d = H * dk
    f = lambda t: t.double().cpu()
    q_u = f(qu).view(B, T, H, dk).transpose(1, 2)
    q_v = f(qv).view(B, T, H, dk).transpose(1, 2)
    k = f(qkv[:, d:2 * d]).view(B, T, H, dk).transpose(1, 2)
    v = f(qkv[:, 2 * d:]).view(B, T, H, dk).transpose(1, 2)
    p = f(P).view(T, H, dk).transpose(0, 1)[None]
    ac = q_u @ k.transpose(-1, -2)
    bd = O.rel_shift_legacy(q_v @ p.transpose(-1, -2))
    sc = (ac + bd) / math.sqrt(dk)
    m = keymask.cpu().bool()[:, None, None, :]
    sc = sc.masked_fill(~m, -float("inf"))
    lse = torch.logsumexp(sc, dim=-1)
    pr = torch.softmax(sc, dim=-1).masked_fill(~m, 0.0)
    pr = torch.nan_to_num(pr, nan=0.0)
    ctx = (pr @ v).transpose(1, 2).reshape(B * T, d)
    return ctx, pr, lse


CASES = [  # B, H, T, dk, lengths
    (2, 2, 136, 32, None),
    (3, 2, 200, 64, [200, 131, 0]),          # ragged + a fully masked utterance
    (2, 4, 264, 128, [264, 97]),
    (2, 2, 1120, 192, [1120, 1000]),         # the benchmark shape per utterance
    (1, 1, 40, 96, [37]),
]


@pytest.mark.parametrize("B,H,T,dk,lengths", CASES)
def test_fused_forward_matches_exact_formula_and_materialised_path(B, H, T, dk, lengths):
    from a3t_amd import ops
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=T + dk, lengths=lengths)
    d = H * dk
    ctx = torch.full((B * T, d), 7.0, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=DEV)
    ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk))
    torch.cuda.synchronize()
    ref, _, rlse = _exact(qkv, qu, qv, P, keymask, B, H, T, dk)
    got = ctx.float().cpu().double()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max()) / scale
    mat, _, _ = _materialised(qkv, qu, qv, P, keymask, B, H, T, dk, (0.0, 0))
    err_mat = float((mat.float().cpu().double() - ref).abs().max()) / scale
    print(f"[B{B} H{H} T{T} dk{dk}] fused max err {err:.2e}, materialised {err_mat:.2e} (of scale {scale:.2f})")
    assert err < 1.2e-2, err                      # bf16 probabilities x bf16 V, fp32 accumulate
    assert err <= max(1.5 * err_mat, 6e-3)        # no worse than the path it replaces
    valid = torch.isfinite(rlse)
    l = lse.cpu().double()
    assert bool(torch.isinf(l[~valid]).all()) and bool((l[~valid] > 0).all())
    assert float((l[valid] - rlse[valid]).abs().max()) < 2e-3


@pytest.mark.parametrize("B,H,T,dk,lengths", CASES[:4])
def test_fused_forward_dropout_uses_the_same_mask_as_the_materialised_path(B, H, T, dk, lengths):
    from a3t_amd import ops
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=3 * T + dk, lengths=lengths)
    d = H * dk
    drop = (0.2, 0x9E3779B1)
    ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=DEV)
    ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk), drop=drop)
    mat, probs, pdrop = _materialised(qkv, qu, qv, P, keymask, B, H, T, dk, drop)
    torch.cuda.synchronize()
    # exact formula with the mask read off the materialised dropped probabilities
    _, pr, _ = _exact(qkv, qu, qv, P, keymask, B, H, T, dk)
    keep = (pdrop.float().cpu() != 0) | (probs.float().cpu() == 0)
    v = qkv[:, 2 * d:].double().cpu().view(B, T, H, dk).transpose(1, 2)
    ref = ((pr * keep / (1.0 - drop[0])) @ v).transpose(1, 2).reshape(B * T, d)
    scale = max(1.0, float(ref.abs().max()))
    err = float((ctx.float().cpu().double() - ref).abs().max()) / scale
    err_mat = float((mat.float().cpu().double() - ref).abs().max()) / scale
    print(f"[B{B} H{H} T{T} dk{dk}] dropout: fused {err:.2e}, materialised {err_mat:.2e}")
    assert err < 1.5e-2 and err <= max(1.5 * err_mat, 8e-3)
