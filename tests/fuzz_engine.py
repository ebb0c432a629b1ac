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
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 10)
