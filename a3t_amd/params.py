"""Flat parameter store + reference state_dict mapping.

All trainable tensors live in ONE flat fp32 HBM buffer (and one flat gradient buffer, plus Adam
moments) so the optimiser, the gradient all-reduce (RCCL) and the grad-norm are single passes over
contiguous memory.  Kernel-friendly layouts differ from torch's in three places (Conv1d weights
are [out][tap][in]; q/k/v projections are one fused [3d][d] matrix; 1x1 convs are plain matrices);
``state_dict``/``load_state_dict`` convert to/from the reference's key names and shapes
 so reference checkpoints load unchanged.
"""
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch

from .config import A3TConfig

ALIGN = 64  # elements (256 B): every parameter view is 16-byte aligned for vector loads


def _block_layout(c: A3TConfig, K: int) -> List[Tuple[str, tuple]]:
    d, ff, kf = c.adim, c.ff, c.ff_kernel
    out = []
    for f in ("ffm", "ff"):
        out += [(f"{f}.ln.g", (d,)), (f"{f}.ln.b", (d,)), (f"{f}.w1", (ff, kf, d)), (f"{f}.b1", (ff,)),
                (f"{f}.w2", (d, kf, ff)), (f"{f}.b2", (d,))]
    out += [("mha.ln.g", (d,)), ("mha.ln.b", (d,)), ("mha.wqkv", (3 * d, d)), ("mha.bqkv", (3 * d,)),
            ("mha.wpos", (d, d)), ("mha.u", (d,)), ("mha.v", (d,)), ("mha.wo", (d, d)), ("mha.bo", (d,))]
    out += [("cnv.ln.g", (d,)), ("cnv.ln.b", (d,)), ("cnv.pw1", (2 * d, d)), ("cnv.pb1", (2 * d,)),
            ("cnv.dw", (d, K)), ("cnv.db", (d,)), ("cnv.bn.g", (d,)), ("cnv.bn.b", (d,)), ("cnv.pw2", (d, d)),
            ("cnv.pb2", (d,))]
    out += [("fin.ln.g", (d,)), ("fin.ln.b", (d,))]
    return out


def param_layout(c: A3TConfig) -> "OrderedDict[str, tuple]":
    d = c.adim
    lay = OrderedDict()
    lay["seg"] = (c.seg_table, d)
    lay["mask_feature"] = (c.idim,)
    lay["emb.w"] = (d, c.idim)
    lay["emb.b"] = (d,)
    lay["emb.ln.g"] = (d,)
    lay["emb.ln.b"] = (d,)
    lay["temb"] = (c.vocab, d)
    if c.spk_embed_dim > 0:
        lay["spk.w"] = (d, c.spk_embed_dim)
        lay["spk.b"] = (d,)
    for i in range(c.enc_blocks):
        for n, s in _block_layout(c, c.enc_kernel):
            lay[f"enc.{i}.{n}"] = s
    lay["enc.after.g"] = (d,)
    lay["enc.after.b"] = (d,)
    for i in range(c.dec_blocks):
        for n, s in _block_layout(c, c.dec_kernel):
            lay[f"dec.{i}.{n}"] = s
    lay["dec.after.g"] = (d,)
    lay["dec.after.b"] = (d,)
    lay["sfc.w"] = (c.odim, d)
    lay["sfc.b"] = (c.odim,)
    for l in range(c.postnet_layers):
        ic = c.odim if l == 0 else c.postnet_chans
        oc = c.odim if l == c.postnet_layers - 1 else c.postnet_chans
        lay[f"post.{l}.w"] = (oc, c.postnet_filts, ic)
        lay[f"post.{l}.bn.g"] = (oc,)
        lay[f"post.{l}.bn.b"] = (oc,)
    return lay


def buffer_layout(c: A3TConfig) -> "OrderedDict[str, tuple]":
    """BatchNorm running statistics (not trained, part of the checkpoint)."""
    lay = OrderedDict()
    for st, n in (("enc", c.enc_blocks), ("dec", c.dec_blocks)):
        for i in range(n):
            lay[f"{st}.{i}.cnv.bn.rm"] = (c.adim,)
            lay[f"{st}.{i}.cnv.bn.rv"] = (c.adim,)
    for l in range(c.postnet_layers):
        oc = c.odim if l == c.postnet_layers - 1 else c.postnet_chans
        lay[f"post.{l}.bn.rm"] = (oc,)
        lay[f"post.{l}.bn.rv"] = (oc,)
    return lay


def _ref_block_map(ref: str, mine: str, c: A3TConfig):
    """(reference key, store name, row slice | None, kind) for one EncoderLayer."""
    d = c.adim
    m = []
    a = ref + "self_attn."
    m += [(a + "pos_bias_u", mine + "mha.u", None, "reshape"), (a + "pos_bias_v", mine + "mha.v", None, "reshape")]
    for j, n in enumerate(("q", "k", "v")):
        m += [(a + f"linear_{n}.weight", mine + "mha.wqkv", (j * d, (j + 1) * d), "rows"),
              (a + f"linear_{n}.bias", mine + "mha.bqkv", (j * d, (j + 1) * d), "rows")]
    m += [(a + "linear_out.weight", mine + "mha.wo", None, "reshape"),
          (a + "linear_out.bias", mine + "mha.bo", None, "reshape"),
          (a + "linear_pos.weight", mine + "mha.wpos", None, "reshape")]
    for rf, mf in (("feed_forward", "ff"), ("feed_forward_macaron", "ffm")):
        m += [(ref + rf + ".w_1.weight", mine + mf + ".w1", None, "conv"),
              (ref + rf + ".w_1.bias", mine + mf + ".b1", None, "reshape"),
              (ref + rf + ".w_2.weight", mine + mf + ".w2", None, "conv"),
              (ref + rf + ".w_2.bias", mine + mf + ".b2", None, "reshape")]
    cm = ref + "conv_module."
    m += [(cm + "pointwise_conv1.weight", mine + "cnv.pw1", None, "reshape"),
          (cm + "pointwise_conv1.bias", mine + "cnv.pb1", None, "reshape"),
          (cm + "depthwise_conv.weight", mine + "cnv.dw", None, "reshape"),
          (cm + "depthwise_conv.bias", mine + "cnv.db", None, "reshape"),
          (cm + "norm.weight", mine + "cnv.bn.g", None, "reshape"),
          (cm + "norm.bias", mine + "cnv.bn.b", None, "reshape"),
          (cm + "norm.running_mean", mine + "cnv.bn.rm", None, "buffer"),
          (cm + "norm.running_var", mine + "cnv.bn.rv", None, "buffer"),
          (cm + "norm.num_batches_tracked", mine + "cnv.bn", None, "nbt"),
          (cm + "pointwise_conv2.weight", mine + "cnv.pw2", None, "reshape"),
          (cm + "pointwise_conv2.bias", mine + "cnv.pb2", None, "reshape")]
    for rn, mn in (("norm_ff", "ff.ln"), ("norm_mha", "mha.ln"), ("norm_ff_macaron", "ffm.ln"),
                   ("norm_conv", "cnv.ln"), ("norm_final", "fin.ln")):
        m += [(ref + rn + ".weight", mine + mn + ".g", None, "reshape"),
              (ref + rn + ".bias", mine + mn + ".b", None, "reshape")]
    return m


def reference_key_map(c: A3TConfig):
    m = [("encoder.segment_emb.weight", "seg", None, "reshape"),
         ("encoder.speech_embed.0.mask_feature", "mask_feature", None, "reshape"),
         ("encoder.speech_embed.1.weight", "emb.w", None, "reshape"),
         ("encoder.speech_embed.1.bias", "emb.b", None, "reshape"),
         ("encoder.speech_embed.2.weight", "emb.ln.g", None, "reshape"),
         ("encoder.speech_embed.2.bias", "emb.ln.b", None, "reshape"),
         ("encoder.text_embed.0.weight", "temb", None, "reshape")]
    if c.spk_embed_dim > 0:    # extension keys (no reference counterpart): optional when loading a reference checkpoint
        m += [("spk_proj.weight", "spk.w", None, "optional"), ("spk_proj.bias", "spk.b", None, "optional")]
    for i in range(c.enc_blocks):
        m += _ref_block_map(f"encoder.encoders.{i}.", f"enc.{i}.", c)
    m += [("encoder.after_norm.weight", "enc.after.g", None, "reshape"),
          ("encoder.after_norm.bias", "enc.after.b", None, "reshape")]
    for i in range(c.dec_blocks):
        m += _ref_block_map(f"decoder.encoders.{i}.", f"dec.{i}.", c)
    m += [("decoder.after_norm.weight", "dec.after.g", None, "reshape"),
          ("decoder.after_norm.bias", "dec.after.b", None, "reshape"),
          ("sfc.weight", "sfc.w", None, "reshape"), ("sfc.bias", "sfc.b", None, "reshape")]
    for l in range(c.postnet_layers):
        p = f"postnet.postnet.{l}."
        m += [(p + "0.weight", f"post.{l}.w", None, "conv"), (p + "1.weight", f"post.{l}.bn.g", None, "reshape"),
              (p + "1.bias", f"post.{l}.bn.b", None, "reshape"),
              (p + "1.running_mean", f"post.{l}.bn.rm", None, "buffer"),
              (p + "1.running_var", f"post.{l}.bn.rv", None, "buffer"),
              (p + "1.num_batches_tracked", f"post.{l}.bn", None, "nbt")]
    return m


def reference_shape(kind: str, store_shape: tuple, key: str) -> tuple:
    if kind == "conv":
        o, t, i = store_shape
        return (o, i, t)
    return store_shape


class ParamStore:
    """Flat fp32 parameter / gradient / Adam-moment buffers with named views."""

    def __init__(self, cfg: A3TConfig, device):
        self.cfg = cfg
        self.device = torch.device(device)
        self.layout = param_layout(cfg)
        self.offsets: Dict[str, Tuple[int, tuple]] = {}
        off = 0
        for name, shp in self.layout.items():
            n = int(np.prod(shp))
            self.offsets[name] = (off, shp)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.n_params = sum(int(np.prod(s)) for s in self.layout.values())
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.p = {k: self.flat[o:o + int(np.prod(s))].view(s) for k, (o, s) in self.offsets.items()}
        self.g = {k: self.grad[o:o + int(np.prod(s))].view(s) for k, (o, s) in self.offsets.items()}
        self.buf = {}
        for k, s in buffer_layout(cfg).items():
            self.buf[k] = (torch.ones if k.endswith(".rv") else torch.zeros)(s, dtype=torch.float32,
                                                                             device=self.device)
        self.nbt = {}  # num_batches_tracked counters (host ints)
        self.keymap = reference_key_map(cfg)

    # ---- reference checkpoint format ---------------------------------------------------------
    def _view(self, name, rows, kind, grads=False):
        if kind == "buffer":
            return self.buf[name]
        t = (self.g if grads else self.p)[name]
        if rows is not None:
            t = t[rows[0]:rows[1]]
        return t

    def state_dict(self, grads: bool = False) -> "OrderedDict[str, torch.Tensor]":
        """Reference-format state_dict; grads=True returns the gradients under the same keys."""
        sd = OrderedDict()
        for key, name, rows, kind in self.keymap:
            if kind == "nbt":
                if not grads:
                    sd[key] = torch.tensor(self.nbt.get(name, 0), dtype=torch.long)
                continue
            if grads and kind == "buffer":
                continue
            t = self._view(name, rows, kind, grads).detach()
            if kind == "conv":
                t = t.permute(0, 2, 1)
            elif key.endswith("mask_feature"):
                t = t.view(1, 1, -1)
            elif "pos_bias" in key:
                t = t.view(self.cfg.heads, self.cfg.dk)
            elif key.endswith(("pointwise_conv1.weight", "pointwise_conv2.weight")):
                t = t.unsqueeze(-1)
            elif key.endswith("depthwise_conv.weight"):
                t = t.unsqueeze(1)
            sd[key] = t.contiguous().clone()
        return sd

    def load_state_dict(self, sd, strict: bool = True):
        sd = dict(sd)
        # MLMTask.build_model_from_file renames encoder.embed* -> encoder.speech_embed* (tasks/mlm.py:489-494)
        for k in list(sd.keys()):
            if "encoder.embed" in k:
                sd[k.replace("encoder.embed", "encoder.speech_embed")] = sd.pop(k)
        missing = []
        for key, name, rows, kind in self.keymap:
            if key not in sd:
                if kind != "optional":
                    missing.append(key)
                continue
            src = sd.pop(key)
            if not torch.is_tensor(src):
                src = torch.as_tensor(np.asarray(src))
            if kind == "nbt":
                self.nbt[name] = int(src)
                continue
            dst = self._view(name, rows, kind)
            if kind == "conv":
                src = src.permute(0, 2, 1)
            src = src.reshape(dst.shape).to(device=self.device, dtype=torch.float32)
            dst.copy_(src)
        if strict and (missing or sd):
            raise RuntimeError(f"load_state_dict: missing keys {missing[:5]}..., unexpected {list(sd)[:5]}...")
        return missing, list(sd)

    def zero_grad(self):
        self.grad.zero_()
