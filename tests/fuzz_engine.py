"""Randomised end-to-end parity: the fp32 HIP engine (forward, loss, full backward) against the CPU oracle on random
small architectures (width, heads, FFN size, block counts, conv-module kernels, postnet depth) and ragged batches
(padded utterances, padded phone sequences, random span masks).  The oracle is pinned to the reference by
tests/golden/*.npz (tests/test_oracle_golden.py)."""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import a3t_oracle as O  # noqa: E402

DEV = "cuda"


def one_case(rng, verbose=False):
    from a3t_amd.config import A3TConfig
    from a3t_amd.engine import MLMEngine
    from a3t_amd.params import ParamStore
    heads = rng.choice([1, 2, 4])
    adim = heads * rng.choice([8, 16, 24])
    oc = O.A3TConfig(adim=adim, heads=heads, ff=rng.choice([24, 48, 64]), enc_blocks=rng.choice([1, 2]),
                     dec_blocks=rng.choice([1, 2]), enc_kernel=rng.choice([3, 7, 15]), dec_kernel=rng.choice([7, 31]),
                     postnet_layers=rng.choice([2, 3, 5]), postnet_chans=rng.choice([16, 24]), vocab=rng.randrange(8, 40))
    B = rng.randrange(1, 4)
    T_mel, T_phn = rng.randrange(24, 90), rng.randrange(3, 12)
    lengths = [T_mel] + [rng.randrange(max(T_phn + 2, T_mel // 2), T_mel + 1) for _ in range(B - 1)]
    text_lengths = [T_phn] + [rng.randrange(2, T_phn + 1) for _ in range(B - 1)]
    seed = rng.randrange(1 << 20)
    batch = O.synthetic_batch(oc, B, T_mel, T_phn, seed=seed, lengths=lengths, text_lengths=text_lengths)
    state = O.procedural_state(O.param_shapes(oc), seed % 97)
    p = O.to_torch_state(state, requires_grad=True)
    loss, before, after = O.forward_loss(p, batch, oc, True)
    loss.backward()
    c = A3TConfig(**{k: getattr(oc, k) for k in ("idim", "odim", "vocab", "adim", "heads", "ff", "ff_kernel", "enc_blocks",
                                                  "dec_blocks", "enc_kernel", "dec_kernel", "postnet_layers",
                                                  "postnet_chans", "postnet_filts", "max_len", "seg_table", "lsm_weight")})
    store = ParamStore(c, DEV)
    store.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    eng = MLMEngine(c, store, compute="f32", training=True)
    out = eng.forward({k: v.to(DEV) for k, v in batch.items()})
    store.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    tag = (f"d{adim} H{heads} ff{oc.ff} blocks {oc.enc_blocks}+{oc.dec_blocks} K{oc.enc_kernel}/{oc.dec_kernel} "
           f"post{oc.postnet_layers}x{oc.postnet_chans} B{B} T{T_mel}+{T_phn} lens {lengths}/{text_lengths}")
    errs = []
    l_ref = float(loss.detach())
    if abs(float(out["loss"]) - l_ref) > 2e-4 * abs(l_ref):
        errs.append(f"loss {float(out['loss'])} vs {l_ref}")
    a_err = float((out["after"].cpu() - after.detach()).abs().max())
    if a_err > 5e-4 * max(1.0, float(after.abs().max())):
        errs.append(f"after max err {a_err:.2e}")
    grads = store.state_dict(grads=True)
    for k, t in p.items():
        if not t.requires_grad or t.grad is None:
            continue
        ref = t.grad
        got = grads[k].cpu()
        tol = 1e-3 * max(1.0, float(ref.abs().max()))
        if float((got - ref).abs().max()) > tol:
            errs.append(f"grad {k}: {float((got - ref).abs().max()):.2e} (max |ref| {float(ref.abs().max()):.2e})")
    if errs or verbose:
        print(("FAIL " if errs else "ok   ") + tag)
        for e in errs[:6]:
            print("      " + e)
    return len(errs)


def one_case_bf16(rng, verbose=False):
    """bf16 production schedule (two streams, fused epilogues, dropout ON) against the exact-fp32 engine with the same
    dropout masks, on random mid-size architectures and ragged, 8-aligned batches."""
    from a3t_amd.config import A3TConfig
    from a3t_amd.engine import MLMEngine
    from a3t_amd.init import xavier_init_
    from a3t_amd.params import ParamStore
    from a3t_amd.collate import synthetic_batch
    heads = rng.choice([1, 2, 4])
    adim = heads * rng.choice([16, 32, 64, 96])
    c = A3TConfig(adim=adim, heads=heads, ff=64 * rng.randrange(1, 9), enc_blocks=rng.choice([1, 2]),
                  dec_blocks=rng.choice([1, 2]), enc_kernel=rng.choice([3, 7, 15]), dec_kernel=rng.choice([7, 31]),
                  postnet_layers=rng.choice([2, 5]), postnet_chans=rng.choice([32, 64, 256]), vocab=rng.randrange(8, 60),
                  dropout_rate=0.2, positional_dropout_rate=0.2, attention_dropout_rate=0.2, postnet_dropout_rate=0.5)
    B = rng.randrange(1, 5)
    T_mel = 8 * rng.randrange(4, 40)
    T_phn = min(8 * rng.randrange(1, 6), T_mel - 8)          # (every phone needs at least one frame: case 27 of the seed-2026 sweep)
    store = ParamStore(c, DEV)
    xavier_init_(store, seed=rng.randrange(1000), bn_gamma=1.0)
    batch = synthetic_batch(c, B, T_mel, T_phn, seed=rng.randrange(1 << 20), device=DEV)
    if B > 1:                                  # pad the tails of the later utterances
        for b in range(1, B):
            L = rng.randrange(T_mel // 2, T_mel + 1)
            batch["speech_mask"][b, :, L:] = False
            batch["masked_position"][b, L:] = False
            batch["speech"][b, L:] = 0
    res = {}
    for compute in ("f32", "bf16"):
        eng = MLMEngine(c, store, compute=compute, training=True, dropout=True)
        eng.step_seed = 17
        eng.refresh_weights()
        store.zero_grad()
        res[compute] = float(eng.forward(batch)["loss"])
        eng.backward()
        torch.cuda.synchronize()
        res[compute + ".g"] = store.grad.clone()
    tag = (f"bf16 d{adim} H{heads} ff{c.ff} blocks {c.enc_blocks}+{c.dec_blocks} K{c.enc_kernel}/{c.dec_kernel} "
           f"post{c.postnet_layers}x{c.postnet_chans} B{B} T{T_mel}+{T_phn}")
    errs = []
    if abs(res["bf16"] - res["f32"]) > 1.5e-2 * abs(res["f32"]):
        errs.append(f"loss {res['bf16']} vs {res['f32']}")
    # The speech-embedding prologue (Linear -> LN -> ReLU) sits behind a kink: every masked frame carries the SAME input
    # row (mask_feature), so one pre-activation within bf16 rounding of zero flips the ReLU mask of a whole column in all
    # masked frames at once (seen: 1 of 64 columns at -0.0066 -> prologue gradients at cosine 0.90 while the gradient
    # entering the prologue agrees to 0.998).  Those five parameters are held to a looser bound, everything else to 0.98.
    import numpy as _np
    pro = torch.zeros_like(res["f32.g"], dtype=torch.bool)
    for name, (off, shp) in store.offsets.items():
        if name in ("emb.w", "emb.b", "emb.ln.g", "emb.ln.b", "mask_feature"):
            pro[off:off + int(_np.prod(shp))] = True
    cosf = lambda m: float(torch.nn.functional.cosine_similarity(res["bf16.g"][m], res["f32.g"][m], dim=0))
    cos, cos_pro = cosf(~pro), cosf(pro)
    if not (cos > 0.98) or not (cos_pro > 0.85) or not bool(torch.isfinite(res["bf16.g"]).all()):
        errs.append(f"grad cosine {cos} (speech-embedding prologue {cos_pro})")
    if errs or verbose:
        print(("FAIL " if errs else "ok   ") + tag + (f"  (loss {res['bf16']:.4f}/{res['f32']:.4f} cos {cos:.5f})"))
        for e in errs:
            print("      " + e)
        if errs:      # which parameters disagree (worst cosines first, with their share of the gradient norm)
            import numpy as _np
            rows = []
            tot = float(res["f32.g"].norm())
            for name, (off, shp) in store.offsets.items():
                n = int(_np.prod(shp))
                a_, b_ = res["bf16.g"][off:off + n], res["f32.g"][off:off + n]
                if float(b_.norm()) > 0:
                    rows.append((float(torch.nn.functional.cosine_similarity(a_, b_, dim=0)), name, float(b_.norm()) / tot,
                                 float(a_.norm()) / float(b_.norm())))
            for cs, name, share, ratio in sorted(rows)[:10]:
                print(f"      {name:28s} cos {cs:8.5f}  norm share {share:6.3f}  |bf16|/|f32| {ratio:6.3f}")
    return len(errs)


def run_bf16(seed=0, n=10, verbose=True):
    rng = random.Random(seed)
    fails = 0
    for i in range(n):
        try:
            fails += 1 if one_case_bf16(rng, verbose) else 0
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"EXC bf16 case {i}: {type(e).__name__}: {e}")
    if verbose:
        print(f"{n} bf16 cases, {fails} failures")
    return fails


def run(seed=0, n=10, verbose=True):
    rng = random.Random(seed)
    fails = 0
    for i in range(n):
        try:
            fails += 1 if one_case(rng, verbose) else 0
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"EXC case {i}: {type(e).__name__}: {e}")
    if verbose:
        print(f"{n} cases, {fails} failures")
    return fails


if __name__ == "__main__":
    _seed, _n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 10)
    run(_seed, _n)
    run_bf16(_seed, _n)
