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
