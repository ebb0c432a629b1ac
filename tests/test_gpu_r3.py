"""Round-3 GPU tests: x-vector conditioning (BASELINE configs[3]; an extension without reference behaviour -> property tests),
bias gradient of the speech-embedding Linear in dropout-on steps."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _tiny(spk=0, **kw):
    from a3t_amd.config import A3TConfig
    return A3TConfig(adim=32, heads=2, ff=64, enc_blocks=1, dec_blocks=1, enc_kernel=7, dec_kernel=7, postnet_layers=2,
                     postnet_chans=16, postnet_filts=5, spk_embed_dim=spk, **kw)


def _store(c, seed=0):
    from a3t_amd.init import xavier_init_
    from a3t_amd.params import ParamStore
    st = ParamStore(c, DEV)
    xavier_init_(st, seed=seed, bn_gamma=1.0)
    g = torch.Generator().manual_seed(seed + 1)
    for k, v in st.p.items():        # non-degenerate biases / norms
        if v.dim() == 1 and not k.endswith((".g",)):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    return st


def test_xvector_conditioning_properties():
    """spembs (B, S) -> Linear -> added to every token after the embedding prologue (config spk_embed_dim > 0):
    (1) a zero projection leaves the model exactly where it is without conditioning, (2) the loss depends on the speaker
    vectors, (3) the gradients of the projection match central finite differences of the fp32 loss, (4) the state_dict carries
    the two extension keys and a reference checkpoint without them still loads."""
    from a3t_amd.collate import synthetic_batch
    from a3t_amd.engine import MLMEngine
    c1 = _tiny(spk=24)
    st1 = _store(c1)
    batch = synthetic_batch(c1, 3, 48, 8, seed=3, device=DEV)
    assert batch["spembs"].shape == (3, 24)
    e1 = MLMEngine(c1, st1, compute="f32", training=True, dropout=False)
    # (1) zero projection == no conditioning
    c0 = _tiny(spk=0)
    st0 = _store(c0)
    sd = st1.state_dict()
    assert "spk_proj.weight" in sd and "spk_proj.bias" in sd
    missing, unexpected = st0.load_state_dict({k: v for k, v in sd.items() if not k.startswith("spk_proj")})
    assert not missing and not unexpected
    e0 = MLMEngine(c0, st0, compute="f32", training=True, dropout=False)
    b0 = {k: v for k, v in batch.items() if k != "spembs"}
    l0 = float(e0.forward(b0)["loss"])
    w_saved, b_saved = st1.p["spk.w"].clone(), st1.p["spk.b"].clone()
    st1.p["spk.w"].zero_(), st1.p["spk.b"].zero_()
    assert float(e1.forward(batch)["loss"]) == l0
    # a reference-format checkpoint (no spk_proj.* keys) loads into the conditioned model
    missing, unexpected = st1.load_state_dict(st0.state_dict())
    assert not missing and not unexpected
    st1.p["spk.w"].copy_(w_saved), st1.p["spk.b"].copy_(b_saved)
    # (2) the loss sees the speaker vectors
    l1 = float(e1.forward(batch)["loss"])
    b2 = dict(batch, spembs=batch["spembs"].flip(0).contiguous())
    assert abs(float(e1.forward(b2)["loss"]) - l1) > 1e-6 * abs(l1)
    # (3) gradients vs central differences
    e1.forward(batch)
    st1.zero_grad()
    e1.backward()
    torch.cuda.synchronize()
    gw, gb = st1.g["spk.w"].clone(), st1.g["spk.b"].clone()
    assert float(gw.abs().max()) > 0 and float(gb.abs().max()) > 0

    def loss_at(t, idx, dv):
        old = float(t[idx])
        t[idx] = old + dv
        l = float(e1.forward(batch)["loss"])
        t[idx] = old
        return l

    rs = np.random.RandomState(0)
    for t, g in ((st1.p["spk.b"], gb), (st1.p["spk.w"], gw)):
        for _ in range(4):
            idx = tuple(int(rs.randint(0, s)) for s in t.shape)
            h = 2e-2
            fd = (loss_at(t, idx, h) - loss_at(t, idx, -h)) / (2 * h)
            assert abs(fd - float(g[idx])) < 2e-2 * max(abs(fd), abs(float(g[idx])), 0.05), (idx, fd, float(g[idx]))


def test_xvector_bf16_step_and_plugin_surface():
    """bf16 schedule with conditioning: finite, the projection gets a gradient, the plugin model passes spembs through."""
    from a3t_amd.collate import synthetic_batch
    from a3t_amd.espnet_model import ESPnetMLMEncAsDecoderModel
    c = _tiny(spk=16)
    m = ESPnetMLMEncAsDecoderModel([str(i) for i in range(c.vocab)], c.odim, None, None, c, device=DEV, compute="bf16",
                                   dropout=False)
    from a3t_amd.init import xavier_init_
    xavier_init_(m.store, seed=1, bn_gamma=1.0)
    batch = synthetic_batch(c, 2, 48, 8, seed=5, device=DEV)
    m.train()
    loss, stats, weight = m(**batch)
    loss.backward()
    assert math.isfinite(float(loss)) and int(weight) == 2
    assert float(m.store.g["spk.w"].abs().max()) > 0
    with torch.no_grad():
        m.eval()
        l_a = float(m(**batch)[0])
        l_b = float(m(**dict(batch, spembs=batch["spembs"] * 0))[0])
    assert math.isfinite(l_a) and l_a != l_b


def test_embedding_bias_gets_its_gradient_with_dropout_on():
    """Round-3 regression: with the recipe's dropout on, the speech-embedding Linear's bias (a layer WITHOUT output dropout)
    got no gradient at all; every trained parameter must receive one, in both numeric modes."""
    from a3t_amd.collate import synthetic_batch
    from a3t_amd.engine import MLMEngine
    c = _tiny()
    for compute in ("f32", "bf16"):
        st = _store(c, seed=2)
        e = MLMEngine(c, st, compute=compute, training=True, dropout=True)
        batch = synthetic_batch(c, 2, 48, 8, seed=7, device=DEV)
        e.forward(batch)
        st.zero_grad()
        e.backward()
        torch.cuda.synchronize()
        dead = [k for k, v in st.g.items() if float(v.abs().max()) == 0.0]
        assert not dead, (compute, dead)


def test_first_postnet_conv_gradient_with_split_forward_operand():
    """bf16 mode runs the FIRST postnet conv on (hi, lo) = (bf16(before), bf16(before - hi)) in the forward pass but keeps only
    hi for its weight gradient (engine.py, A3T_POST_F32_FIRST): the gradient of post.0.w must still track the fp32 engine's
    (the operand mismatch is one bf16 ulp of the input and was not bounded by any test), as must sfc.w's, whose
    forward now runs on the exact-fp32 MFMA while its gradient uses the bf16 operands."""
    from a3t_amd.collate import synthetic_batch
    from a3t_amd.engine import MLMEngine
    c = _tiny(postnet_dropout_rate=0.0)
    grads = {}
    batch = synthetic_batch(c, 3, 96, 16, seed=9, device=DEV)
    for compute in ("f32", "bf16"):
        st = _store(c, seed=4)
        e = MLMEngine(c, st, compute=compute, training=True, dropout=False)
        e.forward(batch)
        st.zero_grad()
        e.backward()
        torch.cuda.synchronize()
        grads[compute] = {k: st.g[k].clone() for k in ("post.0.w", "post.1.w", "sfc.w", "sfc.b")}
    for k, g32 in grads["f32"].items():
        g16 = grads["bf16"][k]
        rel = float((g16 - g32).norm() / g32.norm())
        cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
        assert rel < 6e-2 and cos > 0.998, (k, rel, cos)


def test_panel_gemm_kernel_against_the_128_kernel_and_torch():
    """The 384-column panel GEMM (gemm_bf16_pn.hip: one 160-row panel x all 384 columns per workgroup) forced on: bit-identical
    to the 128x128 kernel (same products in the same k order) with every epilogue it fuses -- bias, relu, dropout, alpha, fp32
    residual, bf16 / fp32 store, column sums -- on M tails, several rounds of panels, Conv1d over time with 3 and 5 taps across
    utterance boundaries, and within bf16 rounding of fp32 torch math."""
    from a3t_amd import _lib, ops
    from a3t_amd._lib import ACT_NONE, ACT_RELU, BF16
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
    N = 384

    def both(fn):
        outs = []
        old8, oldp = lib.a3t_gemm_8p_mode(0), lib.a3t_gemm_pn_mode(0)
        try:
            for mode in (0, 1):
                lib.a3t_gemm_pn_mode(mode)
                outs.append(fn())
                outs.append(lib.a3t_gemm_last_kernel().decode())
        finally:
            lib.a3t_gemm_pn_mode(oldp)
            lib.a3t_gemm_8p_mode(old8)
        torch.cuda.synchronize()
        return outs

    rel = lambda a, b: float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-9))
    for (M, K, act, f32out, res, drop, cs) in [(160, 128, ACT_NONE, False, False, False, False), (1000, 384, ACT_RELU, False, False, False, True),
                                               (4099 * 8, 768, ACT_NONE, True, True, True, False), (45000, 256, ACT_NONE, False, False, True, True),
                                               (777, 1152, ACT_RELU, True, True, False, True), (8, 128, ACT_NONE, True, False, False, True)]:
        x, W, b = rn(M, K).bfloat16(), rn(N, K, sc=0.05).bfloat16(), rn(N)
        R = rn(M, N) if res else None

        def f():
            o = torch.empty(M, N, device=DEV, dtype=torch.float32 if f32out else torch.bfloat16)
            csum = torch.zeros(N, device=DEV) if cs else None
            ops.gemm(x, W, o, M, N, K, K, 1, K, 1, N, bias=b, R=R, alpha=0.7, act=act, compute=BF16, drop=(0.1, 99) if drop else None,
                     colsum=csum, colsum_scale=0.5)
            return (o, csum)
        (o0, c0), k0, (o1, c1), k1 = both(f)
        assert "pn_kernel" in k1 and "pn_kernel" not in k0, (k0, k1)
        assert torch.equal(o0, o1), (M, K, rel(o1, o0))
        if cs:
            assert rel(c1, c0) < 1e-5
        if not drop:
            ref = x.float() @ W.float().t() + b
            ref = (torch.relu(ref) if act == ACT_RELU else ref) * 0.7 + (R if res else 0)
            assert rel(o1, ref) < 1e-2
    for (B, T, cin, taps) in [(3, 200, 128, 3), (7, 333, 256, 5), (5, 7, 128, 3)]:
        M = B * T
        h = rn(M, cin).bfloat16()
        W2 = rn(N, taps, cin, sc=0.03).bfloat16()
        b2, xres = rn(N), rn(M, N)

        def f():
            o = torch.empty(M, N, device=DEV)
            ops.conv_fwd(h, W2, o, T, (taps - 1) // 2, bias=b2, R=xres, alpha=0.5, compute=BF16, drop=(0.1, 4242))
            o2 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            ops.conv_fwd(h, W2, o2, T, (taps - 1) // 2, compute=BF16)
            return (o, o2)
        (o0, p0), k0, (o1, p1), k1 = both(f)
        assert "pn_kernel<true>" in k1 and "pn_kernel" not in k0, (k0, k1)
        assert torch.equal(o0, o1) and torch.equal(p0, p1)
        ref = torch.nn.functional.conv1d(h.float().view(B, T, cin).transpose(1, 2), W2.float().permute(0, 2, 1), padding=(taps - 1) // 2)
        assert rel(p1, ref.transpose(1, 2).reshape(M, N)) < 1e-2
    # wider outputs walk 384-column chunks (N = 768 / 1152 / 1536, a ragged N, an odd number of K-tiles); the ReLU' mask tensor S
    for (M, Nw, K, taps, smask) in [(2000, 768, 384, 1, False), (3000, 1152, 192, 1, False), (1800, 1536, 384, 3, False),
                                    (1800, 1536, 384, 3, True), (1234, 1000, 256, 1, True), (41000, 768, 128, 1, False)]:
        cin = K // taps
        T = 100 if taps > 1 else 0
        a, W, b = rn(M, cin).bfloat16(), rn(Nw, taps, cin, sc=0.05).bfloat16(), rn(Nw)
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
        assert "pn_kernel" in k1 and "pn_kernel" not in k0, (k0, k1)
        assert torch.equal(o0, o1) and rel(c1, c0) < 1e-5, (M, Nw, K, rel(o1, o0), rel(c1, c0))
    # the randomised GEMM / conv sweep with the kernel forced on for every problem it accepts (ragged M, N < 384 and N tails,
    # dilation, K-tile counts from 1 up): against fp32 torch math
    import fuzz_gemm as mod
    old8, oldp = lib.a3t_gemm_8p_mode(0), lib.a3t_gemm_pn_mode(1)
    try:
        fails, seen = mod.run(seed=23, n_cases=80, verbose=False)
    finally:
        lib.a3t_gemm_pn_mode(oldp)
        lib.a3t_gemm_8p_mode(old8)
    assert fails == 0
    assert sum(v for k, v in seen.items() if k.startswith("gemm_bf16_pn_kernel")) >= 12, seen
    # what the cost model (the default mode) takes and what the kernel must leave alone
    old = lib.a3t_gemm_pn_mode(2)
    try:
        flags = ops.G8_BIAS_ACT | ops.G8_DROP | ops.G8_F32_OR_RES | ops.G8_COLSUM
        assert ops.gemm_pn_supported(35840, 384, 4608, 3, flags) and ops.gemm_pn_supported(35840, 1536, 1152, 3, ops.G8_SMASK | ops.G8_COLSUM)
        assert not ops.gemm_pn_supported(35840, 512, 2048) and not ops.gemm_pn_supported(1000, 384, 4608, 3, flags)
        assert not ops.gemm_pn_supported(35840, 384, 100)
        assert ops.gemm_pn_supported(35840, 1536, 1152, 3, ops.G8_KEEP_IN | ops.G8_COLSUM)      # (the row-major nibble image, below)
    finally:
        lib.a3t_gemm_pn_mode(old)


def test_row_major_keep_image_replaces_the_saved_activation_as_the_relu_mask():
    """Round 6 (a3t_gemm_desc::keep_layout = 1): the first FFN conv (multi_layer_conv.py:52-63; bias, ReLU, dropout) writes, from
    the 128-row kernel's vector epilogue, one nibble per four outputs = (stored value > 0); the data gradient of the second conv
    reads that image in the place of the saved activation -- on the panel kernel and on the 128-row kernel, bit-identical to the
    launch that reads the activation itself (S = h), column sums included.  M tails, utterance boundaries, both output types."""
    from a3t_amd import _lib, ops
    from a3t_amd._lib import ACT_RELU, BF16
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(5)
    rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
    for (B, T, cin, ff) in [(6, 300, 384, 1536), (3, 171, 128, 512), (32, 1120, 384, 1536)]:
        M = B * T
        x, W1, b1 = rn(M, cin).bfloat16(), rn(ff, 3, cin, sc=0.05).bfloat16(), rn(ff, sc=0.3)
        h0 = torch.empty(M, ff, device=DEV, dtype=torch.bfloat16)
        ops.conv_fwd(x, W1, h0, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.1, 77))
        bits = (h0.float() > 0).view(M, ff // 4, 4).to(torch.uint8)
        want = bits[..., 0] | (bits[..., 1] << 1) | (bits[..., 2] << 2) | (bits[..., 3] << 3)
        for mode in (0, 2):      # the writer is the 128-row kernel's vector epilogue whatever the panel kernel's mode
            h = torch.empty_like(h0)
            keep = torch.full((M * ff // 4,), 0xFF, device=DEV, dtype=torch.uint8)
            oldp = lib.a3t_gemm_pn_mode(mode)
            try:
                ops.conv_fwd(x, W1, h, T, 1, bias=b1, act=ACT_RELU, compute=BF16, drop=(0.1, 77), keep_out=keep, keep_layout=1)
                torch.cuda.synchronize()
            finally:
                lib.a3t_gemm_pn_mode(oldp)
            assert "gemm_bf16_glds_kernel" in lib.a3t_gemm_last_kernel().decode()
            assert torch.equal(h, h0)
            assert torch.equal(keep.view(M, ff // 4), want), mode
        assert 0.3 < float(bits.float().mean()) < 0.6
        ga, W2t = rn(M, cin).bfloat16(), rn(ff, 3, cin, sc=0.05).bfloat16()
        old8 = lib.a3t_gemm_8p_mode(0)
        try:
            for mode, dt in ((0, torch.bfloat16), (1, torch.bfloat16), (1, torch.float32)):
                oldp = lib.a3t_gemm_pn_mode(mode)
                try:
                    outs = []
                    for kw in (dict(S=h), dict(keep_in=keep, keep_layout=1)):
                        dh = torch.empty(M, ff, device=DEV, dtype=dt)
                        cs = torch.zeros(ff, device=DEV)
                        ops.conv_fwd(ga, W2t, dh, T, 1, alpha=0.625, compute=BF16, colsum=cs, **kw)
                        torch.cuda.synchronize()
                        outs.append((dh, cs, lib.a3t_gemm_last_kernel().decode()))
                finally:
                    lib.a3t_gemm_pn_mode(oldp)
                (d0, c0, k0), (d1, c1, k1) = outs
                assert ("pn_kernel" in k1) == (mode == 1) and ("pn_kernel" in k0) == (mode == 1), (mode, k0, k1)
                assert torch.equal(d0, d1), (M, mode)
                assert torch.allclose(c0, c1, rtol=1e-4, atol=1e-3 * float(c0.abs().max()))
                assert bool(((d1 == 0) | (h > 0)).all()) and float(d1.float().abs().max()) > 0
        finally:
            lib.a3t_gemm_8p_mode(old8)
    # the contract: no residual with keep_out, no S beside keep_in, bf16 operands
    with pytest.raises(Exception):
        ops.conv_fwd(x, W1, torch.empty(M, ff, device=DEV), T, 1, R=torch.zeros(M, ff, device=DEV), compute=BF16, keep_out=keep, keep_layout=1)
    with pytest.raises(Exception):
        ops.conv_fwd(ga, W2t, torch.empty(M, ff, device=DEV, dtype=torch.bfloat16), T, 1, compute=BF16, S=h, keep_in=keep, keep_layout=1)


def test_panel_gemm_inside_the_benchmark_step():
    """BASELINE configs[1] at full size (B=32, T=1120 -> 35840 tokens = 224 panels): the library's cost model sends the second FFN
    conv, the data gradient of the first (transposed w_1 shadow), linear_out / pointwise_conv2 and the Linear data gradients
    (transposed shadows) to the panel kernel; the same step with all of that switched off gives the same loss and gradients."""
    import os
    from a3t_amd import _lib, ops
    from a3t_amd.collate import synthetic_batch
    from a3t_amd.config import config_c2
    from a3t_amd.engine import MLMEngine
    from a3t_amd.init import xavier_init_
    from a3t_amd.params import ParamStore
    lib = _lib.load()
    c = config_c2()
    store = ParamStore(c, DEV)
    xavier_init_(store, seed=0, bn_gamma=1.0)
    batch = synthetic_batch(c, 32, 1000, 120, seed=4, device=DEV)
    M = 32 * 1120
    e_on = MLMEngine(c, store, compute="bf16", training=True, dropout=True)
    ops.PROFILE = []
    try:
        l_on = float(e_on.forward(batch)["loss"])
        store.zero_grad()
        e_on.backward()
        torch.cuda.synchronize()
        kernels = [r[0] for r in ops.PROFILE]
    finally:
        ops.PROFILE = None
    n_pn = sum("pn_kernel" in k for k in kernels)
    # per block: 2 FFN x (conv2 forward + conv1 data gradient) + linear_out, pw2 forward + 4 Linear data gradients = 10
    assert n_pn >= 10 * (c.enc_blocks + c.dec_blocks), (n_pn, sorted(set(kernels)))
    assert e_on._ffn_plan(M)[1] and e_on._ffn_plan(M)[2] and e_on._lin_plans, (e_on._ffn_plan(M), e_on._lin_plans)
    g_on = store.grad.clone()
    old = lib.a3t_gemm_pn_mode(0)
    os.environ["A3T_LIN_DGRAD_T"] = "0"
    try:
        e_off = MLMEngine(c, store, compute="bf16", training=True, dropout=True)
        e_off.step_seed = e_on.step_seed - 1
        l_off = float(e_off.forward(batch)["loss"])
        store.zero_grad()
        e_off.backward()
        torch.cuda.synchronize()
    finally:
        lib.a3t_gemm_pn_mode(old)
        del os.environ["A3T_LIN_DGRAD_T"]
    assert not any(e_off._ffn_plan(M)) and not any(e_off._lin_plans.values())
    rel = float((store.grad - g_on).norm() / g_on.norm())
    print(f"[configs[1] full size] panel GEMM path vs 128x128 path: loss {l_on:.6f} / {l_off:.6f}, gradient rel. diff {rel:.2e}")
    assert abs(l_off - l_on) <= 1e-6 * abs(l_on), (l_off, l_on)
    assert rel < 1e-5, rel
