"""Parameter initialisation = espnet2.torch_utils.initialize.initialize(model, "xavier_uniform")
(espnet2/torch_utils/initialize.py:63-88): xavier-uniform for every >1-d tensor, zeros for every
1-d tensor, then Embedding / LayerNorm modules reset to their torch defaults (N(0,1) with the
padding row zeroed; gamma=1, beta=0).  Host RNG, done once at start-up."""
import math

import torch

from .params import ParamStore


def _xavier(shape_ref, gen):
    """torch.nn.init.xavier_uniform_ on a tensor of the REFERENCE shape (fan_in/out from dims 1/0
    times the receptive field)."""
    rf = 1
    for s in shape_ref[2:]:
        rf *= s
    fan_in, fan_out = shape_ref[1] * rf, shape_ref[0] * rf
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape_ref, generator=gen) * 2 - 1) * a


def xavier_init_(store: ParamStore, seed: int = 0, bn_gamma: float = 0.0):
    """bn_gamma=0.0 reproduces the recipe exactly (the 1-d zeroing also hits BatchNorm.weight,
    SURVEY appendix A); benchmarks pass 1.0 so no GEMM runs on an all-zero operand."""
    c = store.cfg
    gen = torch.Generator().manual_seed(seed)
    sd = store.state_dict()
    new = {}
    for key, t in sd.items():
        shp = tuple(t.shape)
        if key.endswith("num_batches_tracked"):
            new[key] = torch.tensor(0)
        elif key.endswith("running_mean"):
            new[key] = torch.zeros(shp)
        elif key.endswith("running_var"):
            new[key] = torch.ones(shp)
        elif key in ("encoder.segment_emb.weight", "encoder.text_embed.0.weight"):
            w = torch.randn(shp, generator=gen)
            w[-1].zero_()                       # padding_idx=-1
            new[key] = w
        elif len(shp) > 1:
            new[key] = _xavier(shp, gen)
        elif ("norm" in key and "conv_module.norm" not in key) or key.startswith("encoder.speech_embed.2"):
            new[key] = torch.ones(shp) if key.endswith("weight") else torch.zeros(shp)   # LayerNorm reset
        elif key.endswith(("conv_module.norm.weight", ".1.weight")):
            new[key] = torch.full(shp, float(bn_gamma))
        else:
            new[key] = torch.zeros(shp)
    store.load_state_dict(new)
    return store
