"""Fused legacy rel-pos attention (a3t_attn_fwd / a3t_attn_fwd_train / a3t_attn_bwd_ds) against the materialised path (batched GEMMs +
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


def _exact_bwd(qkv, qu, qv, P, keymask, dctx, B, H, T, dk, keep=None, pdrop=0.0):
    """fp64 autograd of the reference formula on the same bf16 operands -> d(q+u), d(q+v), dK, dV, d linear_pos(pos)."""
    d = H * dk
    f = lambda t: t.double().cpu().clone().requires_grad_(True)
    qu_, qv_, k_, v_, p_ = f(qu), f(qv), f(qkv[:, d:2 * d]), f(qkv[:, 2 * d:]), f(P)
    hv = lambda t: t.view(B, T, H, dk).transpose(1, 2)
    ac = hv(qu_) @ hv(k_).transpose(-1, -2)
    bd = O.rel_shift_legacy(hv(qv_) @ p_.view(T, H, dk).transpose(0, 1)[None].transpose(-1, -2))
    sc = (ac + bd) / math.sqrt(dk)
    m = keymask.cpu().bool()[:, None, None, :]
    sc = sc.masked_fill(~m, -1e300)
    pr = torch.softmax(sc, dim=-1).masked_fill(~m, 0.0)
    if keep is not None:
        pr = pr * keep / (1.0 - pdrop)
    ctx = (pr @ hv(v_)).transpose(1, 2).reshape(B * T, d)
    ctx.backward(dctx.double().cpu())
    return dict(ctx=ctx.detach(), dqu=qu_.grad, dqv=qv_.grad, dk=k_.grad, dv=v_.grad, dpos=p_.grad)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


def test_forward_only_passes_take_the_fused_kernel_by_default(monkeypatch):
    """Default A3T_FUSED_ATTN=auto: an eval / need_grad=False forward with >= 64 attention workgroups runs the fused forward
    kernel (no T x T tensor), a training forward and a tiny batch stay on the materialised path, and both give the same
    mel output to bf16 accuracy and the oracle's output within the bf16 tolerance of the parity tests."""
    from a3t_amd.config import A3TConfig
    from a3t_amd.engine import MLMEngine
    from a3t_amd.params import ParamStore
    monkeypatch.delenv("A3T_FUSED_ATTN", raising=False)
    oc = O.A3TConfig(adim=128, heads=2, ff=256, enc_blocks=2, dec_blocks=1, postnet_layers=2, postnet_chans=32)
    c = A3TConfig(adim=128, heads=2, ff=256, enc_blocks=2, dec_blocks=1, postnet_layers=2, postnet_chans=32, vocab=oc.vocab)
    state = O.procedural_state(O.param_shapes(oc), 5)
    B = 16
    cpu_batch = O.synthetic_batch(oc, B=B, T_mel=232, T_phn=24, seed=9, lengths=[232] * 10 + [141, 77, 200, 99, 232, 8],
                                  text_lengths=[24] * 10 + [17, 9, 24, 11, 24, 3])
    batch = {k: v.to(DEV) for k, v in cpu_batch.items()}
    store = ParamStore(c, DEV)
    store.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    eng = MLMEngine(c, store, compute="bf16", training=False)
    assert eng.fused_attn_auto
    out = eng.forward(batch, need_grad=False)                       # 16 * 2 * 2 = 64 workgroups
    assert sum(k.endswith(".fused") for k in eng.sv) == 3
    fused_after = out["after"].float().clone()
    small = {k: v[:2].contiguous() for k, v in batch.items()}
    eng.forward(small, need_grad=False)                             # 8 workgroups: materialised
    assert not any(k.endswith(".fused") for k in eng.sv)
    eng.forward(batch, need_grad=True)                              # gradients wanted: fused forward that saves probabilities
    assert not any(k.endswith(".fused") for k in eng.sv) and sum(k.endswith(".rs") for k in eng.sv) == 3
    eng.fused_attn_auto = False
    mat_after = eng.forward(batch, need_grad=False)["after"].float()
    scale = float(mat_after.abs().max())
    assert float((fused_after - mat_after).abs().max()) < 3e-2 * scale
    with torch.no_grad():
        _, _, ra = O.forward_loss(O.to_torch_state(state), cpu_batch, oc, False)
    err = (fused_after.cpu().reshape(ra.shape) - ra)
    assert float(err.pow(2).mean().sqrt()) < 1e-2 * scale and float(err.abs().max()) < 6e-2 * scale


# ---- round 4: the training forward (a3t_attn_fwd_train) -------------------------------------------------------------------
@pytest.mark.parametrize("drop_p", [0.0, 0.2])
@pytest.mark.parametrize("B,H,T,dk,lengths", CASES[:4] + [(2, 2, 328, 192, [328, 0]), (1, 2, 296, 64, [200])])
def test_training_forward_saves_probabilities_the_materialised_backward_can_use(B, H, T, dk, lengths, drop_p):
    """probs * rowscale = the exact softmax (bf16 accuracy), probs_drop = probs under the counter-RNG mask x 1/(1-p) (the mask
    a3t_relpos_softmax_fwd draws), ctx / lse identical to a3t_attn_fwd, all-masked utterances -> zeros, rowscale 0."""
    from a3t_amd import ops
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=5 * T + dk, lengths=lengths)
    if lengths is not None and len(lengths) == 1:       # a hole in the middle of the key mask and masked leading tiles
        keymask[0, :40] = 0
        keymask[0, 120:150] = 0
    d = H * dk
    drop = (drop_p, 0x1234567)
    ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
    ctx2 = torch.zeros_like(ctx)
    lse, lse2 = torch.zeros(B, H, T, device=DEV), torch.zeros(B, H, T, device=DEV)
    probs = torch.full((B, H, T, T), 9.0, device=DEV, dtype=torch.bfloat16)
    pdrop = torch.full((B, H, T, T), 9.0, device=DEV, dtype=torch.bfloat16) if drop_p else None
    rs = torch.full((B, H, T), -1.0, device=DEV)
    ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, 1.0 / math.sqrt(dk), drop=drop)
    ops.attn_fwd(qu, qv, qkv, P, keymask, ctx2, lse2, B, H, T, 1.0 / math.sqrt(dk), drop=drop)
    torch.cuda.synchronize()
    assert torch.equal(ctx, ctx2) and torch.equal(lse, lse2)
    _, pr, rlse = _exact(qkv, qu, qv, P, keymask, B, H, T, dk)
    got = probs.float().cpu().double() * rs.cpu().double()[..., None]
    assert bool(torch.isfinite(got).all())
    err = float((got - pr).abs().max())
    print(f"[B{B} H{H} T{T} dk{dk} p{drop_p}] normalised saved probabilities: max err {err:.2e}")
    assert err < 6e-3                                  # bf16 probabilities (relative 2^-9) of rows that sum to 1
    valid = torch.isfinite(rlse)
    assert float(rs.cpu()[~valid].abs().max() if (~valid).any() else 0.0) == 0.0
    if drop_p:
        mat, mprobs, mpdrop = _materialised(qkv, qu, qv, P, keymask, B, H, T, dk, drop)
        keep = (mpdrop.float().cpu() != 0) | (mprobs.float().cpu() == 0)
        want = probs.float().cpu() * keep / (1.0 - drop_p)
        gotd = pdrop.float().cpu()
        rel = float(((gotd - want).abs() / (want.abs() + 1e-20)).max())
        assert rel < 1e-2, rel                        # same mask; one more bf16 rounding


@pytest.mark.parametrize("bwd_ds", ["0", "1"])
@pytest.mark.parametrize("dropout", [False, True])
def test_engine_training_step_with_the_fused_forward_matches_the_materialised_step(dropout, bwd_ds, monkeypatch):
    """Default training path of round 4 (fused forward that saves un-normalised probabilities + materialised backward) against
    the fully materialised step: same dropout masks, loss and every parameter gradient to bf16 accuracy.  bwd_ds = 1: the
    opt-in score-gradient kernel (a3t_attn_bwd_ds) in place of the dprobs GEMM + softmax backward."""
    from a3t_amd.config import A3TConfig
    from a3t_amd.engine import MLMEngine
    from a3t_amd.params import ParamStore
    monkeypatch.delenv("A3T_FUSED_ATTN", raising=False)
    monkeypatch.setenv("A3T_ATTN_BWD_DS", bwd_ds)
    oc = O.A3TConfig(adim=128, heads=2, ff=256, enc_blocks=2, dec_blocks=1, postnet_layers=2, postnet_chans=32)
    c = A3TConfig(adim=128, heads=2, ff=256, enc_blocks=2, dec_blocks=1, postnet_layers=2, postnet_chans=32, vocab=oc.vocab,
                  dropout_rate=0.2, positional_dropout_rate=0.2, attention_dropout_rate=0.2, postnet_dropout_rate=0.5)
    state = O.procedural_state(O.param_shapes(oc), 5)
    B = 16
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(oc, B=B, T_mel=232, T_phn=24, seed=9, lengths=[232] * 10 + [141, 77, 200, 99, 232, 8],
                                                        text_lengths=[24] * 10 + [17, 9, 24, 11, 24, 3]).items()}
    res = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("A3T_FUSED_ATTN_TRAIN", fused)
        store = ParamStore(c, DEV)
        store.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
        eng = MLMEngine(c, store, compute="bf16", training=True, dropout=dropout)
        assert eng.fused_attn_train == (fused == "1") and eng.attn_bwd_ds == (bwd_ds == "1")
        loss = float(eng.forward(batch)["loss"])
        assert sum(k.endswith(".rs") for k in eng.sv) == (3 if fused == "1" else 0)
        store.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        res[fused] = (loss, store.state_dict(grads=True))
    l0, g0 = res["0"]
    l1, g1 = res["1"]
    print(f"loss materialised {l0:.5f} fused-forward {l1:.5f}")
    assert abs(l0 - l1) < 5e-3 * abs(l0), (l0, l1)
    bad = []
    for k in g0:
        a, b_ = g1[k].double().flatten(), g0[k].double().flatten()
        nb = float(b_.norm())
        if nb < 1e-6 or k.endswith("depthwise_conv.bias") or k.endswith("linear_k.bias"):
            continue
        cos = float((a * b_).sum() / (a.norm() * b_.norm() + 1e-30))
        ratio = float(a.norm()) / nb
        if cos < 0.985 or not (0.93 < ratio < 1.07):
            bad.append((k, round(cos, 4), round(ratio, 4)))
    assert not bad, bad[:10]


@pytest.mark.parametrize("train", [False, True])
def test_fixed_reference_maximum_overflow_falls_back_and_stays_exact(train):
    """The 32-query forward fixes a row's reference maximum at the first key tile.  Keys whose scores exceed it by more than
    ~88 overflow the row sum; the block then raises its redo flag and is recomputed -- by the rescaling 16-query kernel for
    forward-only passes, by the same kernel with a first sweep for the true maximum for the training forward (it has to
    re-write the saved probabilities).  Forced here: the keys of the second half of one utterance are scaled by 40."""
    from a3t_amd import ops
    B, H, T, dk = 2, 2, 328, 64
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=99)
    d = H * dk
    qkv = qkv.clone()
    kview = qkv.view(B, T, 3 * d)
    kview[1, 200:, d:2 * d] = (kview[1, 200:, d:2 * d].float() * 40.0).bfloat16()
    ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=DEV)
    if train:
        probs = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
        rs = torch.zeros(B, H, T, device=DEV)
        ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, None, rs, B, H, T, 1.0 / math.sqrt(dk))
    else:
        ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk))
    torch.cuda.synchronize()
    ref, pr, rlse = _exact(qkv, qu, qv, P, keymask, B, H, T, dk)
    assert float(rlse.max()) > 100.0                       # the scenario really has scores far beyond the first tile's
    got = ctx.float().cpu().double()
    assert bool(torch.isfinite(got).all())
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max()) / scale
    assert err < 2e-2, err
    assert float((lse.cpu().double() - rlse).abs().max()) < 2e-2
    if train:
        pn = probs.float().cpu().double() * rs.cpu().double()[..., None]
        assert bool(torch.isfinite(pn).all()) and float((pn - pr).abs().max()) < 8e-3


def _split_off():
    from a3t_amd import _lib
    return _lib.load().a3t_attn_split_mode(0)


def _split_restore(old):
    from a3t_amd import _lib
    _lib.load().a3t_attn_split_mode(old)


@pytest.mark.parametrize("train,drop_p", [(False, 0.0), (True, 0.0), (True, 0.2)])
@pytest.mark.parametrize("B,H,T,dk,lengths", [
    (16, 2, 1120, 64, None),                                   # 288 blocks = 256 + 32 -> the tail runs as 4 key ranges
    (12, 3, 1000, 32, [1000, 640, 0, 977] + [1000] * 8),        # 288 blocks; an empty utterance, ragged lengths in and out of the tail
    (43, 1, 896, 96, [896] * 40 + [100, 896, 30]),              # 301 blocks = 256 + 45 -> 4 ranges; keys end inside the first range
    (16, 2, 1120, 192, [1120] * 15 + [777]),                    # the benchmark's instantiation (d_k = 192), 288 blocks
    (9, 4, 1000, 128, [1000] * 8 + [520]),                      # d_k = 128 (configs[3]'s head width), 288 blocks
])
def test_key_split_tail_blocks_equal_the_unsplit_launch(B, H, T, dk, lengths, train, drop_p):
    """A launch whose last round of 128-query blocks fills at most half the chip runs those blocks split into key ranges
    (launch_fwd16 in attn_fused.hip).  Every range uses the block's reference maximum, so the saved probabilities are
    bit-identical to the unsplit launch and ctx / lse / rowscale differ by the fp32 summation order only."""
    from a3t_amd import ops
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("block counts chosen for 256 CUs")
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=7 + B, lengths=lengths)
    d = H * dk
    drop = (drop_p, 4242) if drop_p else (0.0, 0)

    def run():
        ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(B, H, T, device=DEV)
        if not train:
            ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk), drop=drop)
            torch.cuda.synchronize()
            return ctx, lse
        probs = torch.full((B, H, T, T), 7.0, device=DEV, dtype=torch.bfloat16)
        pdrop = torch.full((B, H, T, T), 7.0, device=DEV, dtype=torch.bfloat16) if drop_p else None
        rs = torch.zeros(B, H, T, device=DEV)
        ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, 1.0 / math.sqrt(dk), drop=drop)
        torch.cuda.synchronize()
        return ctx, lse, rs, probs, pdrop

    got = run()
    old = _split_off()
    try:
        ref = run()
    finally:
        _split_restore(old)
    assert old == 1
    assert bool(torch.isfinite(got[0].float()).all())
    assert float((got[0].float() - ref[0].float()).abs().max()) <= 2.0 ** -7 * max(1.0, float(ref[0].float().abs().max()))
    fin = torch.isfinite(ref[1])
    assert bool((torch.isfinite(got[1]) == fin).all())
    assert float((got[1][fin] - ref[1][fin]).abs().max()) < 1e-5
    assert bool((got[1][~fin] == ref[1][~fin]).all())
    if train:
        assert float(((got[2] - ref[2]).abs() / ref[2].abs().clamp_min(1e-30)).max()) < 1e-5
        assert torch.equal(got[3], ref[3])
        if drop_p:
            assert torch.equal(got[4], ref[4])


@pytest.mark.parametrize("train", [False, True])
def test_key_split_tail_block_that_overflows_is_recomputed(train):
    """The overflow flag of a split block is raised by the fold kernel from the summed row sums; the fixup launch then
    recomputes the whole block unsplit.  Keys of the last utterance (its blocks are the tail of the launch) scaled by 40."""
    from a3t_amd import ops
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("block counts chosen for 256 CUs")
    B, H, T, dk = 29, 2, 640, 64                           # 290 blocks = 256 + 34
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=5)
    d = H * dk
    qkv = qkv.clone()
    kview = qkv.view(B, T, 3 * d)
    kview[B - 1, 300:, d:2 * d] = (kview[B - 1, 300:, d:2 * d].float() * 40.0).bfloat16()
    ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=DEV)
    if train:
        probs = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
        rs = torch.zeros(B, H, T, device=DEV)
        ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, None, rs, B, H, T, 1.0 / math.sqrt(dk))
    else:
        ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk))
    torch.cuda.synchronize()
    sl = slice(B - 2, B)                                   # check the last two utterances against the exact formula
    M0 = (B - 2) * T
    ref, pr, rlse = _exact(qkv[M0:], qu[M0:], qv[M0:], P, keymask[sl], 2, H, T, dk)
    assert float(rlse.max()) > 100.0
    got = ctx[M0:].float().cpu().double()
    assert bool(torch.isfinite(got).all())
    assert float((got - ref).abs().max()) / max(1.0, float(ref.abs().max())) < 2e-2
    assert float((lse[sl].cpu().double() - rlse).abs().max()) < 2e-2
    if train:
        pn = probs[sl].float().cpu().double() * rs[sl].cpu().double()[..., None]
        assert bool(torch.isfinite(pn).all()) and float((pn - pr).abs().max()) < 8e-3


@pytest.mark.parametrize("hm", [False, True])
@pytest.mark.parametrize("drop_p", [0.0, 0.2])
@pytest.mark.parametrize("B,H,T,dk,lengths", CASES[:3] + [(2, 2, 328, 192, [328, 0]), (1, 2, 296, 64, [200]), (3, 2, 1120, 96, [1120, 900, 1000])])
def test_score_gradients_from_saved_probabilities_in_one_launch(B, H, T, dk, lengths, drop_p, hm):
    """a3t_attn_bwd_ds against the definition on the same operands (fp64): ds = p (keep/(1-p) dctx V^T - delta) scale with
    p = probs * rowscale, keep read off probs_drop; dbd = ds through the inverse legacy skew (attention.py:145-165), every
    entry written; and against the two kernels it replaces (dprobs GEMM + a3t_relpos_softmax_bwd)."""
    from a3t_amd import ops
    from a3t_amd._lib import BF16
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=3 * T + dk + 1, lengths=lengths)
    if lengths is not None and len(lengths) == 1:
        keymask[0, :40] = 0
        keymask[0, 120:150] = 0
    d = H * dk
    scale = 1.0 / math.sqrt(dk)
    drop = (drop_p, 0x5151) if drop_p else (0.0, 0)
    ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=DEV)
    probs = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    pdrop = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16) if drop_p else None
    rs = torch.zeros(B, H, T, device=DEV)
    ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, scale, drop=drop)
    g = torch.Generator(device="cpu").manual_seed(T + dk)
    dctx = torch.randn(B * T, d, generator=g).to(DEV).bfloat16()
    delta = (dctx.double() * ctx.double()).view(B, T, H, dk).sum(-1).transpose(1, 2)      # the row term the kernel forms itself
    ds = torch.full((B, H, T, T), 3.0, device=DEV, dtype=torch.bfloat16)
    dshape = (H, B, T, T) if hm else (B, H, T, T)
    dbd = torch.full(dshape, 3.0, device=DEV, dtype=torch.bfloat16)
    ops.attn_bwd_ds(dctx, ctx, qkv, probs, rs, ds, dbd, B, H, T, scale, drop=drop, dbd_head_major=hm)
    torch.cuda.synchronize()
    # definition in fp64
    v = qkv[:, 2 * d:].double().view(B, T, H, dk).transpose(1, 2)
    dc = dctx.double().view(B, T, H, dk).transpose(1, 2)
    dP = dc @ v.transpose(-1, -2)
    if drop_p:
        keep = ((pdrop.float() != 0) | (probs.float() == 0)).double()
        dP = dP * keep / (1.0 - drop_p)
    pn = probs.double() * rs.double()[..., None]
    want = pn * (dP - delta[..., None]) * scale
    got = ds.double()
    sc = max(1e-6, float(want.abs().max()))
    err = float((got - want).abs().max()) / sc
    print(f"[B{B} H{H} T{T} dk{dk} p{drop_p}] ds max err / max |ds| = {err:.2e}")
    assert err < 6e-3                                  # bf16 outputs
    # dbd = the inverse skew of the ds the kernel wrote, bit for bit; row 0, columns 0..T-2 zero
    dsb = ds if not hm else ds
    dbv = dbd.transpose(0, 1) if hm else dbd
    ref = torch.zeros(B, H, T + 1, T, device=DEV, dtype=torch.bfloat16)
    ii = torch.arange(T, device=DEV)[:, None].expand(T, T)
    jj = torch.arange(T, device=DEV)[None, :].expand(T, T)
    lo = jj <= ii
    up = jj >= ii + 2
    ref[:, :, ii[lo], (T - 1 - ii + jj)[lo]] = dsb[:, :, ii[lo], jj[lo]]
    ref[:, :, (ii + 1)[up], (jj - ii - 2)[up]] = dsb[:, :, ii[up], jj[up]]
    assert torch.equal(dbv.contiguous(), ref[:, :, :T].contiguous())
    # the kernels it replaces
    dpr = torch.empty(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    vv = qkv.view(-1)[2 * d:]
    zb = (H * T * T, T * T)
    ops.gemm(dctx, vv, dpr, T, T, dk, d, 1, 3 * d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk), b_bs=(T * 3 * d, dk),
             c_bs=zb, compute=BF16)
    ds2 = torch.empty_like(ds)
    dbd2 = torch.zeros_like(dbd)
    ops.relpos_softmax_bwd(probs, dpr, ds2, dbd2, B, H, T, scale, probs_drop=None, drop_p=drop[0], dbd_head_major=hm,
                           drop_key=drop[1], rowscale=rs)
    torch.cuda.synchronize()
    err2 = float((ds2.double() - want).abs().max()) / sc
    assert err <= err2 + 2e-3, (err, err2)              # no worse than the path through a bf16 dprobs
    assert float((ds2.float() - ds.float()).abs().max()) / sc < 2e-2


@pytest.mark.parametrize("B,H,T,dk", [(32, 2, 1120, 192), (16, 4, 1800, 128)])
def test_score_gradient_kernel_is_deterministic_at_full_size(B, H, T, dk):
    """a3t_attn_bwd_ds at configs[1]'s and configs[3]'s attention shapes, five launches: bit-identical outputs.  (Regression: a
    counted vmcnt in front of the LAST V tiles of a workgroup let a tile through before its DMA had landed -- results moved in
    the 4th digit from run to run; the small-shape parity tests never saw it.)"""
    from a3t_amd import ops
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=11)
    d = H * dk
    scale = 1.0 / math.sqrt(dk)
    drop = (0.2, 31337)
    ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=DEV)
    probs = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    pdrop = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    rs = torch.zeros(B, H, T, device=DEV)
    ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, scale, drop=drop)
    del pdrop
    dctx = torch.randn(B * T, d, device=DEV).bfloat16()
    outs = []
    for _ in range(5):
        ds = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
        dbd = torch.zeros(H, B, T, T, device=DEV, dtype=torch.bfloat16)
        ops.attn_bwd_ds(dctx, ctx, qkv, probs, rs, ds, dbd, B, H, T, scale, drop=drop, dbd_head_major=True)
        torch.cuda.synchronize()
        outs.append((ds, dbd))
    for ds, dbd in outs[1:]:
        assert torch.equal(ds, outs[0][0]) and torch.equal(dbd, outs[0][1])
    assert bool(torch.isfinite(outs[0][0].float()).all()) and float(outs[0][0].float().abs().max()) > 0


@pytest.mark.parametrize("B,H,T,dk", [(32, 2, 1120, 192), (16, 4, 1800, 128)])
def test_fused_forward_is_deterministic_at_full_size(B, H, T, dk):
    """The training forward (counted vmcnt waits, DMA ring, key-split tail at configs[1]'s shape) five times at configs[1]'s and
    configs[3]'s attention shapes: ctx, lse, rowscale and both saved probability tensors bit-identical every time."""
    from a3t_amd import ops
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=12, lengths=[T] * (B - 2) + [T - 100, T // 2])
    d = H * dk
    scale = 1.0 / math.sqrt(dk)
    drop = (0.2, 4711)
    outs = []
    for _ in range(5):
        ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros(B, H, T, device=DEV)
        probs = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
        pdrop = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
        rs = torch.zeros(B, H, T, device=DEV)
        ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, scale, drop=drop)
        torch.cuda.synchronize()
        outs.append((ctx, lse, rs, probs, pdrop))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)
    assert bool(torch.isfinite(outs[0][0].float()).all())


@pytest.mark.parametrize("B,H,T,dk,lengths", [(2, 2, 328, 192, [328, 211]), (3, 4, 264, 128, None), (2, 2, 104, 64, [104, 57]),
                                              (32, 2, 1120, 192, None)])
def test_positional_biases_added_inside_the_kernels_equal_the_add_pos_bias_tensors_bit_for_bit(B, H, T, dk, lengths):
    """Round 6: with bias_u / bias_v the fused kernels read q from the q|k|v projection and form bf16(q + pos_bias_u) /
    bf16(q + pos_bias_v) as they load their query fragments (attention.py:190-194; biases staged in LDS).  Every output --
    ctx, lse, the saved probabilities and their dropped copy, 1 / row sum -- is bit-identical to the launch on the two
    tensors a3t_add_pos_bias stores, for the training forward, the forward-only pass and (full size) the key-split tail."""
    from a3t_amd import ops
    rs = np.random.RandomState(7 * T + dk)
    d = H * dk
    qkv, _, _, P, keymask = _inputs(B, H, T, dk, seed=T + dk, lengths=lengths)
    u = torch.from_numpy((rs.standard_normal(d) * 0.3).astype(np.float32)).to(DEV)
    v = torch.from_numpy((rs.standard_normal(d) * 0.3).astype(np.float32)).to(DEV)
    qu = torch.empty(B * T, d, device=DEV, dtype=torch.bfloat16)
    qv = torch.empty_like(qu)
    ops.add_pos_bias(qkv, u, v, qu, qv)
    scale, drop = 1.0 / math.sqrt(dk), (0.2, 0xBEEF)

    def train(pos_bias):
        ctx = torch.zeros(B * T, d, device=DEV, dtype=torch.bfloat16)
        lse, rsc = torch.zeros(B, H, T, device=DEV), torch.zeros(B, H, T, device=DEV)
        probs = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
        pdrop = torch.zeros_like(probs)
        ops.attn_fwd_train(None if pos_bias else qu, None if pos_bias else qv, qkv, P, keymask, ctx, lse, probs, pdrop, rsc, B, H, T,
                           scale, drop=drop, pos_bias=pos_bias)
        ctx2, lse2 = torch.zeros_like(ctx), torch.zeros_like(lse)
        ops.attn_fwd(None if pos_bias else qu, None if pos_bias else qv, qkv, P, keymask, ctx2, lse2, B, H, T, scale, drop=drop,
                     pos_bias=pos_bias)
        torch.cuda.synchronize()
        return ctx, lse, rsc, probs, pdrop, ctx2, lse2
    a, b = train(None), train((u, v))
    for x, y, name in zip(a, b, ("ctx", "lse", "rowscale", "probs", "pdrop", "ctx (forward only)", "lse (forward only)")):
        assert torch.equal(x, y), name
    assert float(a[0].float().abs().max()) > 0


def _sign_bits(x):
    return (x.view(torch.int16) < 0)


@pytest.mark.parametrize("B,H,T,dk,lengths,boost", [c + (False,) for c in CASES[:4]] + [
    (2, 2, 328, 192, [328, 0], False), (3, 2, 1120, 96, [1120, 900, 1000], False), (32, 2, 1120, 192, None, False),
    (2, 2, 328, 64, None, True)])        # boost: keys scaled by 40 -> row sums overflow, the two-pass fixup re-writes the tagged tensor
def test_single_tensor_save_carries_the_dropout_mask_in_the_sign_bits(B, H, T, dk, lengths, boost):
    """Round 6: a3t_attn_fwd_train without probs_drop stores ONE score-sized tensor -- exp(s - m_ref) with the sign bit set where
    attention dropout dropped the element (attention.py:84-96).  Against the two-tensor launch on the same inputs: ctx / lse /
    1 / row sum bit-identical, |x| = probs bit for bit, sign = the mask the dropped copy shows; a3t_attn_bwd_ds reading the mask
    off the sign bits writes the same dS / dBD bits as with the regenerated mask; and the dV product over the tagged tensor
    (a3t_gemm_desc::a_signmask, alpha = 1 / (1 - p)) equals the product over the dropped copy to bf16 rounding -- on the
    128-row kernel and on the streaming kernel (same bits), incl. the full configs[1] shape with its key-split tail blocks."""
    from a3t_amd import _lib, ops
    from a3t_amd._lib import BF16
    lib = _lib.load()
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=3 * T + dk, lengths=lengths)
    d, M = H * dk, B * T
    if boost:
        qkv = qkv.clone()
        kview = qkv.view(B, T, 3 * d)
        kview[1, 200:, d:2 * d] = (kview[1, 200:, d:2 * d].float() * 40.0).bfloat16()
    scale, drop = 1.0 / math.sqrt(dk), (0.2, 0xC0FFEE)
    inv = 1.0 / (1.0 - drop[0])

    def fwd(two):
        ctx = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
        lse, rs = torch.zeros(B, H, T, device=DEV), torch.zeros(B, H, T, device=DEV)
        probs = torch.full((B, H, T, T), 5.0, device=DEV, dtype=torch.bfloat16)
        pdrop = torch.full((B, H, T, T), 5.0, device=DEV, dtype=torch.bfloat16) if two else None
        ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, scale, drop=drop)
        torch.cuda.synchronize()
        return ctx, lse, rs, probs, pdrop
    ctx, lse, rs, probs, pdrop = fwd(True)
    ctx1, lse1, rs1, sp, _ = fwd(False)
    assert torch.equal(lse, lse1) and torch.equal(rs, rs1)
    # ctx: the one-tensor kernel multiplies bf16(p) * keep with V and applies 1 / (1 - p_drop) to the fp32 sums, the two-tensor kernel
    # multiplies bf16(p / (1 - p_drop)) * keep: equal to the rounding of the P operand (2^-9 per element, averaged over the keys)
    cs_ = float(ctx.float().abs().max())
    assert float((ctx.float() - ctx1.float()).abs().max()) <= 1e-2 * cs_ + 1e-6, float((ctx.float() - ctx1.float()).abs().max()) / cs_      # (one bf16 ulp of the largest output is 7.8e-3 of it)
    assert torch.equal((sp.view(torch.int16) & 0x7fff), probs.view(torch.int16)), "|x| is the saved probability"
    if boost:
        assert float(lse.max()) > 100.0 and bool(torch.isfinite(probs.float()).all()) and bool(torch.isfinite(ctx.float()).all())
    sgn = _sign_bits(sp)
    nz = probs.view(torch.int16) != 0
    assert torch.equal(sgn & nz, (pdrop.view(torch.int16) == 0) & nz), "sign = dropped (where the probability is not zero)"
    assert not bool((sgn & (pdrop.view(torch.int16) != 0)).any())
    frac = float((sgn & nz).sum()) / max(1, int(nz.sum()))
    assert abs(frac - drop[0]) < 0.02, frac
    del nz

    # score gradients: mask off the sign bits == mask from the counter RNG, bit for bit (and |x| is taken either way)
    g = torch.Generator(device=DEV).manual_seed(T + dk)
    dctx = torch.randn(M, d, device=DEV, generator=g).bfloat16()
    outs = []
    for pr, sg in ((probs, False), (sp, True), (sp, False)):
        ds = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
        dbd = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
        ops.attn_bwd_ds(dctx, ctx, qkv, pr, rs, ds, dbd, B, H, T, scale, drop=drop, signed_probs=sg)
        torch.cuda.synchronize()
        outs.append((ds, dbd))
    for ds, dbd in outs[1:]:
        assert torch.equal(ds, outs[0][0]) and torch.equal(dbd, outs[0][1])
    assert float(outs[0][0].float().abs().max()) > 0
    del outs

    # dV = dropped(P)^T (rs * dctx): the tagged tensor through a_signmask against the dropped copy
    dcs = torch.empty_like(dctx)
    ops.attn_scale_rows(dctx, rs, dcs, B, H, T)
    zb = (H * T * T, T * T)

    def dv(A, mode, **kw):
        old = lib.a3t_gemm_tt_mode(mode)
        try:
            out = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
            cs = torch.zeros(d, device=DEV)
            ops.gemm(A, dcs, out, T, dk, T, 1, T, 1, d, d, batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * d, dk),
                     c_bs=(T * d, dk), compute=BF16, colsum=cs, colsum_bs1=dk, **kw)
            torch.cuda.synchronize()
            return out, cs, lib.a3t_gemm_last_kernel().decode()
        finally:
            lib.a3t_gemm_tt_mode(old)
    ref, cs_ref, _ = dv(pdrop, 0)
    got0, cs0, n0 = dv(sp, 0, a_signmask=True, alpha=inv)
    assert "gemm_bf16_glds_kernel<2" in n0, n0
    tol = 2e-2 * float(ref.float().abs().max())
    assert float((got0.float() - ref.float()).abs().max()) <= tol
    assert torch.allclose(cs0, cs_ref, rtol=2e-2, atol=2e-2 * float(cs_ref.abs().max()))
    if T >= 512 and dk >= 96:
        got1, cs1, n1 = dv(sp, 1, a_signmask=True, alpha=inv)
        assert n1.startswith("gemm_bf16_tt_kernel<true") and n1.endswith(", true, false>"), n1
        assert torch.equal(got0, got1)
        assert torch.allclose(cs0, cs1, rtol=1e-4, atol=1e-3 * float(cs0.abs().max()))
    # the exact product of the bf16 operands (small cases): sum over queries of keep * p * dcs / (1 - p_drop)
    if B * H * T * T <= 4e6:
        keep = (~sgn).double().cpu()
        Pm = sp.double().abs().cpu() * keep * inv
        X = dcs.double().cpu().view(B, T, H, dk).permute(0, 2, 1, 3)
        want = torch.matmul(Pm.transpose(2, 3), X).permute(0, 2, 1, 3).reshape(M, d)
        assert float((got0.double().cpu() - want).abs().max()) <= 1e-2 * float(want.abs().max()) + 1e-6
    # anything but an m-contiguous bf16 A refuses the flag
    with pytest.raises(Exception):
        ops.gemm(sp, dcs, torch.zeros(M, d, device=DEV, dtype=torch.bfloat16), T, dk, T, T, 1, 1, d, d, batch=B * H, batch_inner=H,
                 a_bs=zb, b_bs=(T * d, dk), c_bs=(T * d, dk), compute=BF16, a_signmask=True)


@pytest.mark.parametrize("signed", ["0", "1"])
def test_engine_step_with_one_saved_probability_tensor_matches_the_two_tensor_step(signed, monkeypatch):
    """The default training step (sign-tagged probabilities, no dropped copy: A3T_ATTN_SIGNED=1) against the two-tensor step
    (=0): same dropout masks, loss and every parameter gradient to bf16 accuracy (ctx and dV sum bf16(p) x keep and apply
    1 / (1 - p_drop) in fp32 instead of multiplying bf16(p / (1 - p_drop)): one rounding less)."""
    from a3t_amd.config import A3TConfig
    from a3t_amd.engine import MLMEngine
    from a3t_amd.params import ParamStore
    monkeypatch.delenv("A3T_FUSED_ATTN", raising=False)
    monkeypatch.setenv("A3T_FUSED_ATTN_TRAIN", "2")
    oc = O.A3TConfig(adim=128, heads=2, ff=256, enc_blocks=2, dec_blocks=1, postnet_layers=2, postnet_chans=32)
    c = A3TConfig(adim=128, heads=2, ff=256, enc_blocks=2, dec_blocks=1, postnet_layers=2, postnet_chans=32, vocab=oc.vocab,
                  dropout_rate=0.2, positional_dropout_rate=0.2, attention_dropout_rate=0.2, postnet_dropout_rate=0.5)
    state = O.procedural_state(O.param_shapes(oc), 5)
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(oc, B=8, T_mel=232, T_phn=24, seed=9, lengths=[232] * 5 + [141, 77, 8],
                                                        text_lengths=[24] * 5 + [17, 9, 3]).items()}
    res = {}
    for sg in ("0", signed):
        monkeypatch.setenv("A3T_ATTN_SIGNED", sg)
        store = ParamStore(c, DEV)
        store.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
        eng = MLMEngine(c, store, compute="bf16", training=True, dropout=True)
        assert eng.attn_signed == (sg == "1")
        loss = float(eng.forward(batch)["loss"])
        n_pdrop = sum(1 for k, v in eng.sv.items() if isinstance(v, tuple) and len(v) == 9 and v[8] is not None)
        assert n_pdrop == (0 if sg == "1" else 3), n_pdrop
        store.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        res[sg] = (loss, store.state_dict(grads=True))
    l0, g0 = res["0"]
    l1, g1 = res[signed]
    assert abs(l0 - l1) <= 2e-3 * abs(l0), (l0, l1)
    for k in g0:
        a, b_ = g1[k].double().flatten(), g0[k].double().flatten()
        nb = float(b_.norm())
        if nb < 1e-6 or k.endswith("linear_k.bias"):       # (zero in exact arithmetic: the softmax is invariant to a key bias)
            continue
        cos = float((a * b_).sum() / (a.norm() * b_.norm() + 1e-30))
        assert cos > 0.999 and 0.98 < float(a.norm()) / nb < 1.02, (k, cos, float(a.norm()) / nb)


@pytest.mark.parametrize("B,H,T,dk,lengths", [(2, 2, 136, 32, None), (3, 2, 200, 64, [200, 131, 0]), (2, 2, 328, 192, [328, 211]),
                                              (3, 2, 1120, 96, [1120, 900, 1000]), (32, 2, 1120, 192, None)])
def test_compact_dbd_matrix_read_as_a_view_of_ds(B, H, T, dk, lengths):
    """Round 6: the compact dBD matrix is the flat dS sequence shifted by T - 1 elements (the inverse of the legacy rel_shift,
    attention.py:145-165): dbd[r][c] = ds_flat[r (T + 1) + c - (T - 1)].  a3t_attn_bwd_ds with dbd = NULL writes dS only, into (b, h)
    blocks with T zeros in front of each; torch.as_strided over that buffer (row stride T + 1, base one element into the zeros) IS
    the matrix the kernel used to store -- bit for bit -- and its two consumers read it that way through the 16-byte LDS-DMA from
    2-byte aligned rows: the dq launch (a3t_gemm_desc::A2 with a2_rs = T + 1; same bits as over the stored matrix) and the
    gradient of linear_pos (A as a view on the 128-row kernel, fp32 atomics: same sums)."""
    from a3t_amd import _lib, ops
    from a3t_amd._lib import ACC_ATOMIC, BF16
    lib = _lib.load()
    qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=5 * T + dk, lengths=lengths)
    d, M = H * dk, B * T
    scale, drop = 1.0 / math.sqrt(dk), (0.2, 0xD0D0)
    ctx = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
    lse, rs = torch.zeros(B, H, T, device=DEV), torch.zeros(B, H, T, device=DEV)
    sp = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, sp, None, rs, B, H, T, scale, drop=drop)
    g = torch.Generator(device=DEV).manual_seed(T)
    dctx = torch.randn(M, d, device=DEV, generator=g).bfloat16()
    ds0 = torch.zeros(B, H, T, T, device=DEV, dtype=torch.bfloat16)
    dbd0 = torch.full((B, H, T, T), 3.0, device=DEV, dtype=torch.bfloat16)
    ops.attn_bwd_ds(dctx, ctx, qkv, sp, rs, ds0, dbd0, B, H, T, scale, drop=drop, signed_probs=True)
    bs = T + T * T
    flat = torch.zeros(B * H * bs, device=DEV, dtype=torch.bfloat16)
    ops.attn_bwd_ds(dctx, ctx, qkv, sp, rs, flat[T:], None, B, H, T, scale, drop=drop, signed_probs=True, ds_bs=bs)
    torch.cuda.synchronize()
    blocks = flat.view(B * H, bs)
    assert torch.equal(blocks[:, T:].reshape(B, H, T, T), ds0)
    assert not bool(blocks[:, :T].any()), "the zeros in front of the blocks are nobody's output"
    view = torch.as_strided(flat, (B * H, T, T), (bs, T + 1, 1), storage_offset=1)
    assert torch.equal(view.reshape(B, H, T, T), dbd0), "the stored dBD matrix is this view of dS"
    # ---- dq = dS K + dBD P: A2 as the view against A2 as the stored matrix
    zb, zv = (H * T * T, T * T), (H * bs, bs)
    NS = 4
    csk = dict(colsum_bs1=dk, colsum_slots=NS, colsum_ss=4 * d)
    old = lib.a3t_gemm_tt_mode(1)
    try:
        outs = []
        for A, A2, zz, extra in ((ds0, dbd0, zb, ()), (flat[T:], flat[1:], zv, (T + 1,))):
            o = torch.full((M, 3 * d), 0.25, device=DEV).bfloat16()
            s = torch.zeros(NS * 4 * d, device=DEV)
            ops.gemm(A, qkv.view(-1)[d:], o, T, dk, T, T, 1, 1, 3 * d, 3 * d, batch=B * H, batch_inner=H, a_bs=zz, b_bs=(T * 3 * d, dk),
                     c_bs=(T * 3 * d, dk), compute=BF16, colsum=s, second=(A2, P, d, (0, dk), s[d:]) + extra, **csk)
            torch.cuda.synchronize()
            assert lib.a3t_gemm_last_kernel().decode().endswith("true>")
            outs.append((o, s))
    finally:
        lib.a3t_gemm_tt_mode(old)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-4 * float(outs[0][1].abs().max()) + 1e-7)
    assert float(outs[0][0].float().view(B, T, 3, H, dk)[:, :, 0].abs().max()) > 0
    # ---- d linear_pos: dP_h += sum_b dbd^T (q + v), the view as the [k][m] operand of the 128-row kernel
    res = []
    for A, acs, zz, vw in ((dbd0, T, zb, False), (flat[1:], T + 1, zv, True)):
        dP = torch.zeros(T, d, device=DEV)
        ops.gemm(A, qv, dP, T, dk, T, 1, acs, 1, d, d, batch=B * H, batch_inner=H, a_bs=zz, b_bs=(T * d, dk), c_bs=(0, dk),
                 acc=ACC_ATOMIC, compute=BF16, a_view=vw)
        torch.cuda.synchronize()
        res.append(dP)
    assert torch.allclose(res[0], res[1], rtol=1e-4, atol=1e-4 * float(res[0].abs().max()) + 1e-7)
    assert float(res[0].abs().max()) > 0
    # an odd leading stride without the flag is refused
    with pytest.raises(Exception):
        ops.gemm(flat[1:], qv, torch.zeros(T, d, device=DEV), T, dk, T, 1, T + 1, 1, d, d, batch=B * H, batch_inner=H, a_bs=zv,
                 b_bs=(T * d, dk), c_bs=(0, dk), acc=ACC_ATOMIC, compute=BF16)


@pytest.mark.parametrize("knob", ["A3T_ATTN_DBD_VIEW", "A3T_ATTN_DQ_DUAL"])
def test_engine_step_without_a_stored_dbd_matrix_matches_the_step_with_one(knob, monkeypatch):
    """The default training step (dq in one launch, dBD read as a view of dS) against its twins (the stored matrix; dq as two
    launches): same masks, same loss bits (nothing in the forward changes), every parameter gradient equal to bf16 accuracy -- with
    the view the dq launch sums the same values in the same order (bit-equal products), only the atomics of d linear_pos reorder."""
    from a3t_amd.config import A3TConfig
    from a3t_amd.engine import MLMEngine
    from a3t_amd.params import ParamStore
    from a3t_amd import _lib
    lib = _lib.load()
    monkeypatch.delenv("A3T_FUSED_ATTN", raising=False)
    monkeypatch.setenv("A3T_FUSED_ATTN_TRAIN", "2")
    for k_ in ("A3T_ATTN_DBD_VIEW", "A3T_ATTN_DQ_DUAL", "A3T_ATTN_SIGNED", "A3T_ATTN_BWD_DS"):      # (the switch under test alone differs)
        monkeypatch.setenv(k_, "1")
    oc = O.A3TConfig(adim=192, heads=2, ff=256, enc_blocks=2, dec_blocks=1, postnet_layers=2, postnet_chans=32)
    c = A3TConfig(adim=192, heads=2, ff=256, enc_blocks=2, dec_blocks=1, postnet_layers=2, postnet_chans=32, vocab=oc.vocab,
                  dropout_rate=0.2, positional_dropout_rate=0.2, attention_dropout_rate=0.2, postnet_dropout_rate=0.5)
    state = O.procedural_state(O.param_shapes(oc), 5)
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(oc, B=8, T_mel=560, T_phn=56, seed=9, lengths=[560] * 5 + [341, 177, 8],
                                                        text_lengths=[56] * 5 + [37, 19, 3]).items()}
    old = lib.a3t_gemm_tt_mode(1)        # (the streaming kernel whenever legal: these shapes are below its cost model's sizes)
    try:
        res = {}
        for val in ("0", "1"):
            monkeypatch.setenv(knob, val)
            store = ParamStore(c, DEV)
            store.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
            eng = MLMEngine(c, store, compute="bf16", training=True, dropout=True)
            loss = float(eng.forward(batch)["loss"])
            store.zero_grad()
            eng.backward()
            torch.cuda.synchronize()
            has_view = any(k[0].endswith("tmp.dsv") or "tmp.dsv" in k[0] for k in eng.ws.bufs)
            has_dbd = any("tmp.dbd16" in k[0] for k in eng.ws.bufs)
            res[val] = (loss, store.state_dict(grads=True), has_view, has_dbd)
    finally:
        lib.a3t_gemm_tt_mode(old)
    (l0, g0, v0, d0), (l1, g1, v1, d1) = res["0"], res["1"]
    assert v1 and not d1, (v1, d1)
    if knob == "A3T_ATTN_DBD_VIEW":
        assert (not v0) and d0, (v0, d0)          # (dq as two launches reads the view too: no stored matrix on either side)
    else:
        assert v0 and not d0, (v0, d0)
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
    for k in g0:
        a, b_ = g1[k].double().flatten(), g0[k].double().flatten()
        nb = float(b_.norm())
        if nb < 1e-6 or k.endswith("linear_k.bias"):
            continue
        cos = float((a * b_).sum() / (a.norm() * b_.norm() + 1e-30))
        assert cos > 0.9995 and 0.99 < float(a.norm()) / nb < 1.01, (k, cos, float(a.norm()) / nb)
