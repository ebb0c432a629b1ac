"""End-to-end GPU parity: the HIP training step (forward, loss, full backward, optimiser) against
the golden vectors produced by the reference and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import a3t_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def _engine(oc, seed, compute="f32", training=True):
    from a3t_amd.config import A3TConfig
    from a3t_amd.engine import MLMEngine
    from a3t_amd.params import ParamStore
    c = A3TConfig(**{k: getattr(oc, k) for k in ("idim", "odim", "vocab", "adim", "heads", "ff", "ff_kernel",
                                                  "enc_blocks", "dec_blocks", "enc_kernel", "dec_kernel",
                                                  "postnet_layers", "postnet_chans", "postnet_filts", "max_len",
                                                  "seg_table", "lsm_weight")})
    store = ParamStore(c, DEV)
    state = O.procedural_state(O.param_shapes(oc), seed)
    store.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    return MLMEngine(c, store, compute=compute, training=training), store


def _to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


def test_state_dict_roundtrip():
    oc = O.tiny_config()
    eng, store = _engine(oc, 1)
    sd = store.state_dict()
    ref = O.procedural_state(O.param_shapes(oc), 1)
    assert set(sd.keys()) == set(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(np.shape(v)), k
        np.testing.assert_array_equal(sd[k].cpu().numpy(), v, err_msg=k)


def test_e2e_tiny_against_reference_golden():
    g = np.load(os.path.join(G, "e2e_tiny.npz"))
    oc = O.tiny_config()
    batch = O.synthetic_batch(oc, B=2, T_mel=48, T_phn=8, seed=11, lengths=[48, 37], text_lengths=[8, 6])
    eng, store = _engine(oc, 1)
    out = eng.forward(_to_dev(batch))
    loss = float(out["loss"])
    assert abs(loss - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))   # 1e-4 relative on the summed loss
    np.testing.assert_allclose(out["before"].cpu().numpy(), g["before"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(out["after"].cpu().numpy(), g["after"], atol=2e-4, rtol=1e-4)
    store.zero_grad()
    eng.backward()
    grads = store.state_dict(grads=True)
    for k in g.files:
        if not k.startswith("grad."):
            continue
        n = k[5:]
        ref = g[k]
        got = grads[n].cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=5e-4 * max(1.0, float(np.abs(ref).max())), rtol=5e-3, err_msg=n)
    sd = store.state_dict()
    for k in g.files:
        if k.startswith("buf.") and "running" in k:
            np.testing.assert_allclose(sd[k[4:]].cpu().numpy(), g[k], atol=1e-5, rtol=1e-4, err_msg=k)
    # eval-mode BN (running statistics) forward
    eng_e, _ = _engine(oc, 1, training=False)
    le = float(eng_e.forward(_to_dev(batch), need_grad=False)["loss"])
    assert abs(le - float(g["loss_eval"])) < 1e-4 * abs(float(g["loss_eval"]))


def test_e2e_reference_yaml_config():
    g = np.load(os.path.join(G, "e2e_refyaml.npz"))
    oc = O.A3TConfig()
    batch = O.synthetic_batch(oc, B=2, T_mel=200, T_phn=30, seed=12, lengths=[200, 163], text_lengths=[30, 24])
    eng, store = _engine(oc, 3)
    assert store.n_params == int(g["n_params"])
    out = eng.forward(_to_dev(batch))
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    np.testing.assert_allclose(out["before"].cpu().numpy(), g["before"], atol=5e-4, rtol=1e-3)
    np.testing.assert_allclose(out["after"].cpu().numpy(), g["after"], atol=5e-4, rtol=1e-3)
    store.zero_grad()
    eng.backward()
    grads = store.state_dict(grads=True)
    for n, gn in zip(g["grad_names"], g["grad_norm"]):
        got = float(grads[str(n)].double().norm())
        assert abs(got - gn) <= 5e-3 * max(gn, 1e-2), (n, got, gn)


def test_e2e_bf16_compute_within_1e2():
    """bf16 MFMA compute path: loss within 1e-2 relative of the fp32 oracle (north_star tolerance)."""
    oc = O.A3TConfig(enc_blocks=2, dec_blocks=2)
    batch = O.synthetic_batch(oc, B=2, T_mel=200, T_phn=24, seed=5)
    p = O.to_torch_state(O.procedural_state(O.param_shapes(oc), 7))
    with torch.no_grad():
        ref, _, _ = O.forward_loss(p, batch, oc, True)
    eng, store = _engine(oc, 7, compute="bf16")
    out = eng.forward(_to_dev(batch))
    assert abs(float(out["loss"]) - float(ref)) < 1e-2 * abs(float(ref))
    store.zero_grad()
    eng.backward()
    assert torch.isfinite(store.grad).all()
    # every parameter gradient of the bf16 path (fused bias sums, bf16 operands) against the fp32 path
    eng32, store32 = _engine(oc, 7, compute="f32")
    eng32.forward(_to_dev(batch))
    store32.zero_grad()
    eng32.backward()
    g16, g32 = store.state_dict(grads=True), store32.state_dict(grads=True)
    bad = []
    for k in g32:
        a, b = g16[k].double().flatten(), g32[k].double().flatten()
        nb = float(b.norm())
        if nb < 1e-6 or k.endswith("depthwise_conv.bias"):   # a bias in front of BatchNorm: gradient is analytically 0
            continue
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        ratio = float(a.norm()) / nb
        if cos < 0.97 or not (0.9 < ratio < 1.1):
            bad.append((k, round(cos, 4), round(ratio, 4)))
    assert not bad, bad[:10]
