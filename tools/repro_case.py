"""Replay tests/fuzz_engine.py's bf16 case stream for a seed until a given case index and dump diagnostics."""
import sys, os, random, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import fuzz_engine as F
from a3t_amd.config import A3TConfig
from a3t_amd.engine import MLMEngine
from a3t_amd.init import xavier_init_
from a3t_amd.params import ParamStore
from a3t_amd.collate import synthetic_batch
DEV = "cuda"
seed, target = int(sys.argv[1]), sys.argv[2]
rng = random.Random(seed)
# run() consumes the rng first in __main__? no: run() and run_bf16() build their own Random(seed)
for i in range(200):
    heads = rng.choice([1, 2, 4])
    adim = heads * rng.choice([16, 32, 64, 96])
    c = A3TConfig(adim=adim, heads=heads, ff=64 * rng.randrange(1, 9), enc_blocks=rng.choice([1, 2]),
                  dec_blocks=rng.choice([1, 2]), enc_kernel=rng.choice([3, 7, 15]), dec_kernel=rng.choice([7, 31]),
                  postnet_layers=rng.choice([2, 5]), postnet_chans=rng.choice([32, 64, 256]), vocab=rng.randrange(8, 60),
                  dropout_rate=0.2, positional_dropout_rate=0.2, attention_dropout_rate=0.2, postnet_dropout_rate=0.5)
    B = rng.randrange(1, 5)
    T_mel = 8 * rng.randrange(4, 40)
    T_phn = 8 * rng.randrange(1, 6)
    xs = rng.randrange(1000)
    bs = rng.randrange(1 << 20)
    Ls = [rng.randrange(T_mel // 2, T_mel + 1) for b in range(1, B)] if B > 1 else []
    tag = f"d{adim} H{heads} ff{c.ff} blocks {c.enc_blocks}+{c.dec_blocks} K{c.enc_kernel}/{c.dec_kernel} post{c.postnet_layers}x{c.postnet_chans} B{B} T{T_mel}+{T_phn}"
    if tag == target:
        break
else:
    raise SystemExit("case not found")
print("found", tag, "xavier seed", xs, "batch seed", bs)
variants = [dict()] + [dict(zip(("dropout_rate", "positional_dropout_rate", "attention_dropout_rate", "postnet_dropout_rate"), v))
                       for v in [(0, 0, 0, 0), (0.2, 0, 0.2, 0.5), (0.2, 0.2, 0, 0.5), (0, 0.2, 0, 0)]]
for ov in variants:
    cc = A3TConfig(**{**{k: getattr(c, k) for k in ("adim", "heads", "ff", "enc_blocks", "dec_blocks", "enc_kernel", "dec_kernel",
                                                      "postnet_layers", "postnet_chans", "vocab", "dropout_rate",
                                                      "positional_dropout_rate", "attention_dropout_rate", "postnet_dropout_rate")}, **ov})
    store = ParamStore(cc, DEV)
    xavier_init_(store, seed=xs, bn_gamma=1.0)
    batch = synthetic_batch(cc, B, T_mel, T_phn, seed=bs, device=DEV)
    res = {}
    for compute in ("f32", "bf16"):
        eng = MLMEngine(cc, store, compute=compute, training=True, dropout=True)
        eng.step_seed = 17
        eng.refresh_weights()
        store.zero_grad()
        res[compute] = float(eng.forward(batch)["loss"])
        eng.backward()
        torch.cuda.synchronize()
        res[compute + ".g"] = store.grad.clone()
        res[compute + ".gx"] = eng.ws.get("grad.x", (B * (T_mel + T_phn), cc.adim)).clone()
        res[compute + ".e"] = eng.sv["embed"][1].float().clone()          # LayerNorm output feeding the ReLU of the speech embedding
        res[compute + ".masked"] = eng.sv["embed"][5].clone()
    cos = float(torch.nn.functional.cosine_similarity(res["bf16.g"], res["f32.g"], dim=0))
    gx_a, gx_b = res["bf16.gx"], res["f32.gx"]
    cos_x = float(torch.nn.functional.cosine_similarity(gx_a.flatten(), gx_b.flatten(), dim=0))
    rowcos = torch.nn.functional.cosine_similarity(gx_a, gx_b, dim=1)
    print(f"dropout {ov or 'as in the case'}: loss {res['bf16']:.4f}/{res['f32']:.4f} flat-grad cos {cos:.5f}; "
          f"encoder-input gradient cos {cos_x:.5f}, per-token cos min {float(rowcos.min()):.4f} "
          f"median {float(rowcos.median()):.4f}, |bf16|/|f32| {float(gx_a.norm() / gx_b.norm()):.4f}")
    ea, eb = res["bf16.e"], res["f32.e"]
    flips = ((ea > 0) != (eb > 0))
    mrow = int(torch.nonzero(res["f32.masked"].view(-1))[0])
    print(f"   ReLU mask of the speech embedding: {int(flips.sum())} of {flips.numel()} elements differ between the modes; in the "
          f"(shared) masked-frame row {int(flips[mrow].sum())} of {flips.shape[1]} columns flip, |e| there: "
          f"{[round(float(v), 4) for v in eb[mrow][flips[mrow]].tolist()]}")
    worst = torch.argsort(rowcos)[:8].tolist()
    print("   worst tokens:", [(t, round(float(rowcos[t]), 3), round(float(gx_b[t].norm()), 4)) for t in worst])
