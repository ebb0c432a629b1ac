"""ESPnet2 model-plugin surface of the A3T masked-mel model, backed by the HIP engine.

Mirror of ``ESPnetMLMEncAsDecoderModel`` / ``ESPnetMLMModel`` (espnet2/tts/sedit/sedit_model.py:
forward :155-187, _forward :350-375, inference :239-284, collect_feats :125-128): same call
signature, same ``(loss, stats, weight)`` return contract (AbsESPnetModel,
espnet2/train/abs_espnet_model.py:9-42), same ``state_dict`` keys/shapes, so
``Trainer.train_one_epoch`` (espnet2/train/trainer.py:545,610) can drive it unchanged:
``loss.backward()`` runs the hand-written backward schedule and leaves gradients in ``p.grad``.

There is no CPU execution path: calling forward without the HIP library / a GPU raises.
"""
from collections import OrderedDict
from typing import Dict, List, Tuple, Union

import torch

from .config import A3TConfig
from .params import ParamStore


class _Namespace:
    pass


class _TrainStep(torch.autograd.Function):
    """One autograd node for the whole model: forward = engine.forward, backward = engine.backward.
    The node's inputs are all parameters, so autograd (and a DistributedDataParallel wrapper's
    reduction hooks, trainer.py:250-265) see an ordinary gradient per parameter."""

    @staticmethod
    def forward(ctx, model, batch, *params):
        out = model._engine().forward(batch, need_grad=True)
        ctx.model = model
        model._last = out
        return out["loss"].clone()

    @staticmethod
    def backward(ctx, gloss):
        from . import ops
        m = ctx.model
        st = m.store
        st.grad.zero_()
        m._engine().backward()
        ops.scale_dev(st.grad, st.grad, gloss.reshape(-1)[:1].contiguous().float())   # d(total)/d(loss), no host sync
        return (None, None) + tuple(st.g[n] for n in m._params)


class ESPnetMLMEncAsDecoderModel(torch.nn.Module):
    def __init__(self, token_list: Union[Tuple[str, ...], List[str]], odim: int, feats_extract, normalize,
                 config: A3TConfig, device="cpu", compute: str = "f32", **model_conf):
        super().__init__()
        self.token_list = list(token_list)
        self.odim = odim
        self.feats_extract = feats_extract
        self.normalize = normalize            # stored, never applied (sedit_model.py:79)
        self.cfg = config
        self.compute = compute
        # torch.nn.Dropout semantics: the recipe's dropout sites are active in train() mode, off in eval() mode
        # (dropout=False builds a deterministic training engine: parity tests, finite-difference checks)
        self.dropout = bool(model_conf.get("dropout", True))
        self.mlm_prob = model_conf.get("mlm_prob", config.mlm_prob)
        self.mean_phn_span = model_conf.get("mean_phn_span", config.mean_phn_span)
        self.masking_schema = model_conf.get("masking_schema", "phn_span")
        self.ignore_id = -1
        self.encoder = _Namespace()           # attributes read by callers
        self.encoder._output_size = config.adim
        self.encoder.segment_emb = True
        self.encoder.pre_speech_layer = 0
        self.decoder = _Namespace()
        self._eng = {}
        self._last = None
        self._build_store(device)

    # ------------------------------------------------------------------ parameters
    def _build_store(self, device, state=None):
        self.store = ParamStore(self.cfg, device)
        if state is not None:
            self.store.load_state_dict(state)
        self._params = OrderedDict()
        for old in [k for k in self._parameters]:
            del self._parameters[old]
        for name, view in self.store.p.items():
            p = torch.nn.Parameter(view, requires_grad=True)
            self._params[name] = p
            self.register_parameter(name.replace(".", "__"), p)
        self._eng = {}

    def _apply(self, fn, *a, **k):
        """model.to(device) / .cuda(): migrate the flat store instead of per-tensor copies."""
        probe = fn(torch.zeros(1, device=self.store.device))
        if probe.device != self.store.device:
            self._build_store(probe.device, self.store.state_dict())
        if probe.dtype != torch.float32:
            raise NotImplementedError("master weights are fp32; pick bf16 compute with compute='bf16'")
        return self

    def _engine(self):
        from .engine import MLMEngine
        key = self.training
        if key not in self._eng:
            self._eng[key] = MLMEngine(self.cfg, self.store, compute=self.compute, training=self.training,
                                       dropout=self.dropout and self.training)
            if (not key) in self._eng:       # share activation buffers between train / eval schedules
                self._eng[key].ws = self._eng[not key].ws
        return self._eng[key]

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = self.store.state_dict()
        out = destination if destination is not None else OrderedDict()
        for k, v in sd.items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True):
        missing, unexpected = self.store.load_state_dict(state_dict, strict=strict)
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ------------------------------------------------------------------ plugin surface
    def collect_feats(self, speech, speech_lengths, text, text_lengths, masked_position, speech_mask, text_mask,
                      speech_segment_pos, text_segment_pos, y_masks=None) -> Dict[str, torch.Tensor]:
        return {"feats": speech, "feats_lengths": speech_lengths}

    def _batch(self, speech, text, masked_position, speech_mask, text_mask, speech_segment_pos, text_segment_pos,
               spembs=None):
        dev = self.store.device
        b = dict(speech=speech.to(dev, torch.float32), text=text.to(dev), masked_position=masked_position.to(dev),
                 speech_mask=speech_mask.to(dev), text_mask=text_mask.to(dev),
                 speech_segment_pos=speech_segment_pos.to(dev), text_segment_pos=text_segment_pos.to(dev))
        if spembs is not None and self.cfg.spk_embed_dim > 0:
            b["spembs"] = spembs.to(dev, torch.float32)
        if self.compute == "bf16":
            b = self._pad_to_dma_granule(b)
        return b

    @staticmethod
    def _pad_to_dma_granule(b):
        """The bf16 kernels move 16-byte granules, so T_mel and T_mel + T_phn must be multiples of 8.  Real batches are
        padded to the longest utterance anyway (collate_fn.py:197-214); here the padding is simply extended: extra speech
        frames are 0 with speech_mask / masked_position False and segment id 0, extra phones are the pad id 0 with
        text_mask False -- exactly what the reference's collate produces for a batch whose longest utterance is a few
        frames longer.  (As in the reference, padded positions still enter the BatchNorm statistics.)"""
        Tm, Tp = b["speech"].shape[1], b["text"].shape[1]
        pm = (-Tm) % 8
        pp = (-(Tm + pm + Tp)) % 8
        if pm == 0 and pp == 0:
            return b
        F = torch.nn.functional
        out = dict(b)
        if pm:
            out["speech"] = F.pad(b["speech"], (0, 0, 0, pm))
            out["masked_position"] = F.pad(b["masked_position"], (0, pm), value=False)
            out["speech_mask"] = F.pad(b["speech_mask"], (0, pm), value=False)
            out["speech_segment_pos"] = F.pad(b["speech_segment_pos"], (0, pm), value=0)
        if pp:
            out["text"] = F.pad(b["text"], (0, pp), value=0)
            out["text_mask"] = F.pad(b["text_mask"], (0, pp), value=False)
            out["text_segment_pos"] = F.pad(b["text_segment_pos"], (0, pp), value=0)
        return out

    def forward(self, speech, text, masked_position, speech_mask, text_mask, speech_segment_pos, text_segment_pos,
                y_masks=None, speech_lengths=None, text_lengths=None, spembs=None):
        """spembs (B, spk_embed_dim): speaker x-vectors; used only by a model built with spk_embed_dim > 0 (an extension:
        the reference model ignores them, sedit_model.py:246)."""
        batch_size = speech.shape[0]
        batch = self._batch(speech, text, masked_position, speech_mask, text_mask, speech_segment_pos,
                            text_segment_pos, spembs)
        if torch.is_grad_enabled() and self.training:
            loss = _TrainStep.apply(self, batch, *self._params.values())
        else:
            loss = self._engine().forward(batch, need_grad=False)["loss"].clone()
            self._last = None
        stats = dict(loss=loss.detach(), loss_mlm=loss.detach(), loss_copy=None)
        weight = torch.tensor([batch_size], dtype=torch.long, device=loss.device)
        return loss.reshape(1), {k: (v.reshape(1) if v is not None else None) for k, v in stats.items()}, weight

    def inference(self, speech, text, masked_position, speech_mask, text_mask, speech_segment_pos, text_segment_pos,
                  span_boundary, y_masks=None, speech_lengths=None, text_lengths=None, feats=None, spembs=None,
                  sids=None, lids=None, threshold=0.5, minlenratio=0.0, maxlenratio=10.0,
                  use_teacher_forcing: bool = False) -> Dict[str, torch.Tensor]:
        """Teacher-forcing branch only (the reference's step-by-step branch is dead code,
        sedit_model.py:285-317): one forward, splice the predicted span into the input mels."""
        if not use_teacher_forcing:
            raise NotImplementedError("only use_teacher_forcing=True is functional in the reference")
        batch = self._batch(speech, text, masked_position, speech_mask, text_mask, speech_segment_pos,
                            text_segment_pos, spembs)
        out = self._engine().forward(batch, need_grad=False)
        zs = out["after"]
        s, e = int(span_boundary[0]), int(span_boundary[1])
        sp = batch["speech"][:, :speech.shape[1]]          # (bf16 mode may have extended the padding)
        return dict(feat_gen=[sp[:, :s], zs[0][s:e].clone(), sp[:, e:]])
