"""How much of the bf16 `after` mel error is the postnet's own arithmetic, and how much is the postnet amplifying the error
that the bf16 decoder left in `before`?  For each fixture: the engine's bf16 `before` pushed through the ORACLE's fp32
postnet (test infrastructure; this is a tool, not the product) = the floor no postnet precision trick can beat."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_parity_r2 as T2
from oracle import a3t_oracle as O
from a3t_amd.espnet_model import ESPnetMLMEncAsDecoderModel

for tag in ("c1", "c4s", "refyaml"):
    oc, seed, batch = T2._extra_case(tag)
    pb = ESPnetMLMEncAsDecoderModel._pad_to_dma_granule(dict(batch))
    p = O.to_torch_state(O.procedural_state(O.param_shapes(oc), seed))
    with torch.no_grad():
        rl, rb, ra = O.forward_loss(p, pb, oc, True)
    eng, store = T2._engine(oc, seed, compute="bf16")
    out = eng.forward(T2._to_dev(pb))
    b16 = out["before"].float().cpu()
    with torch.no_grad():
        y = b16.transpose(1, 2)
        for l in range(oc.postnet_layers):
            pre = f"postnet.postnet.{l}."
            y = F.conv1d(y, p[pre + "0.weight"], None, padding=(oc.postnet_filts - 1) // 2)
            y = O._batch_norm(y, p, pre + "1", True, None)
            if l != oc.postnet_layers - 1:
                y = torch.tanh(y)
        floor = b16 + y.transpose(1, 2)
    e_b = T2._mel_err(b16.numpy(), rb.numpy())
    e_a = T2._mel_err(out["after"].float().cpu().numpy(), ra.numpy())
    e_f = T2._mel_err(floor.numpy(), ra.numpy())
    print(f"[{tag}] before: max {e_b[0]:.2e} rms {e_b[1]:.2e} | after (engine): max {e_a[0]:.2e} rms {e_a[1]:.2e} | "
          f"after with an EXACT fp32 postnet on the engine's before: max {e_f[0]:.2e} rms {e_f[1]:.2e}")
