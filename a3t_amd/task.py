"""Task-plugin surface: mirror of ``MLMTask`` (espnet2/tasks/mlm.py:107-496) for the A3T recipe.

Same classmethod names, argument meaning and side effects (``build_model`` mutates ``args`` exactly
like the reference: legacy rel-pos fallback strings, token_list materialisation) so a maintainer can
register this class as the ``mlm`` task and keep ``egs2/vctk/sedit`` unchanged.  What is *not*
mirrored is the generic AbsTask control plane (samplers, scp readers, reporters): see DESIGN.md.
"""
import argparse
import logging
from pathlib import Path
from typing import Callable, Tuple, Union

import torch
import yaml

from .collate import MLMCollateFn
from .config import A3TConfig
from .espnet_model import ESPnetMLMEncAsDecoderModel
from .features import LogMelFbank
from .init import xavier_init_


class MLMTask:
    num_optimizers: int = 1
    trainer = None            # the reference points at espnet2.train.trainer.Trainer; see a3t_amd.trainer

    # ---- arguments the recipe passes (mlm.sh:526-546,665-687; tasks/mlm.py:127-260) ----
    @classmethod
    def add_task_arguments(cls, parser: argparse.ArgumentParser):
        g = parser.add_argument_group(description="Task related")
        g.add_argument("--token_list", default=None)
        g.add_argument("--odim", type=int, default=None)
        g.add_argument("--input_size", type=int, default=None)
        g.add_argument("--init", default=None)
        g.add_argument("--feats_extract", default="fbank")
        g.add_argument("--feats_extract_conf", type=yaml.safe_load, default={})
        g.add_argument("--normalize", default=None)
        g.add_argument("--normalize_conf", type=yaml.safe_load, default={})
        g.add_argument("--encoder", default="conformer")
        g.add_argument("--encoder_conf", type=yaml.safe_load, default={})
        g.add_argument("--decoder", default="conformer")
        g.add_argument("--decoder_conf", type=yaml.safe_load, default={})
        g.add_argument("--model_conf", type=yaml.safe_load, default={})
        g.add_argument("--use_scaled_pos_enc", type=lambda s: str(s).lower() == "true", default=False)
        g.add_argument("--use_preprocessor", type=lambda s: str(s).lower() == "true", default=False)
        for k in ("token_type", "bpemodel", "non_linguistic_symbols", "cleaner", "g2p"):
            g.add_argument("--" + k, default=None)
        return parser

    @classmethod
    def required_data_names(cls, train: bool = True, inference: bool = False) -> Tuple[str, ...]:
        return ("speech",)

    @classmethod
    def optional_data_names(cls, train: bool = True, inference: bool = False) -> Tuple[str, ...]:
        return ("text", "align_start", "align_end")

    @classmethod
    def build_preprocess_fn(cls, args, train: bool):
        if getattr(args, "use_preprocessor", False):
            raise NotImplementedError("text front-end (CommonPreprocessor / g2p) is outside the hot path")
        return None

    @classmethod
    def _feats(cls, args, device):
        conf = dict(args.feats_extract_conf or {})
        return LogMelFbank(device=device, **conf)

    @classmethod
    def build_collate_fn(cls, args: argparse.Namespace, train: bool, epoch: int = -1, device="cuda",
                         device_out: bool = False) -> Callable:
        """tasks/mlm.py:262-291.  device_out=True (extension): the batch dict comes back as device tensors, built on the GPU."""
        feats = cls._feats(args, device)
        sega = args.encoder_conf.get("input_layer") == "sega_mlm"
        if args.encoder_conf.get("selfattention_layer_type") == "longformer":
            raise NotImplementedError("longformer attention is not part of the A3T recipe")
        factor = 1 if epoch == -1 else 0.8
        dur = args.model_conf.get("duration_predictor_layers", 0) > 0
        return MLMCollateFn(feats, float_pad_value=0.0, int_pad_value=0, mlm_prob=args.model_conf["mlm_prob"] * factor,
                            mean_phn_span=args.model_conf["mean_phn_span"], attention_window=0, pad_speech=False,
                            sega_emb=sega, duration_collect=dur, device_out=device_out)

    @classmethod
    def build_model(cls, args: argparse.Namespace, device="cpu", compute: str = "f32") -> ESPnetMLMEncAsDecoderModel:
        """tasks/mlm.py:328-443."""
        # Same side effects on `args` as the reference (tasks/mlm.py:331-355), because the mutated namespace is what gets
        # dumped to config.yaml: a token file path is replaced by the list it holds, and a model fed pre-computed features
        # (odim given) has its feats_extract entries cleared.
        tl = args.token_list
        if isinstance(tl, str):
            with open(tl, encoding="utf-8") as fh:
                tl = [ln.rstrip() for ln in fh]
            args.token_list = list(tl)
        elif not isinstance(tl, (tuple, list)):
            raise RuntimeError("token_list must be str or list")
        token_list = list(tl)
        vocab_size = len(token_list)
        logging.info("Vocabulary size: %d", vocab_size)
        feats_extract, odim = None, args.odim
        if odim is None:            # the model owns the feature extractor and takes its output size
            feats_extract = cls._feats(args, device)
            odim = feats_extract.output_size()
        else:
            args.feats_extract = args.feats_extract_conf = None
        if getattr(args, "normalize", None) is not None:
            logging.warning("normalize is built but never applied by the reference model (SURVEY §0); ignored")
        if args.encoder != "conformer" or args.decoder not in ("conformer",):
            raise NotImplementedError("only encoder=conformer + decoder=conformer (the A3T recipe) is implemented")
        # the reference force-selects the legacy rel-pos implementation and writes it back into args
        for conf in (args.encoder_conf, args.decoder_conf):
            if conf.get("pos_enc_layer_type", "rel_pos") == "rel_pos":
                conf["pos_enc_layer_type"] = "legacy_rel_pos"
            if conf.get("selfattention_layer_type", "rel_selfattn") == "rel_selfattn":
                conf["selfattention_layer_type"] = "legacy_rel_selfattn"
            if conf["selfattention_layer_type"] != "legacy_rel_selfattn":
                raise NotImplementedError("only legacy_rel_selfattn is implemented")
        if args.model_conf.get("duration_predictor_layers", 0) > 0:
            raise NotImplementedError("ESPnetMLMTTSModel (duration predictor variant) is listed as 'next' (SURVEY §8f)")
        cfg = A3TConfig.from_espnet(args.encoder_conf, args.decoder_conf, args.model_conf, args.input_size, odim,
                                    vocab_size, getattr(args, "feats_extract_conf", None))
        model = ESPnetMLMEncAsDecoderModel(token_list=token_list, odim=odim, feats_extract=feats_extract,
                                           normalize=None, config=cfg, device=device, compute=compute,
                                           **args.model_conf)
        if getattr(args, "init", None) is not None:
            if args.init != "xavier_uniform":
                raise NotImplementedError(f"init={args.init}")
            xavier_init_(model.store, seed=getattr(args, "seed", 0))
        else:
            # init=None: the reference keeps torch's default module initialisation; the flat store starts at zero (a dead
            # model: LayerNorm / BatchNorm gamma = 0).  Give norm scales their torch default (1) so that a checkpoint-less
            # model is at least alive, and say so -- weights are expected to come from load_state_dict (the
            # build_model_from_file path) or from `init: xavier_uniform` (the recipe).
            n = 0
            for k, v in model.store.p.items():
                if k.endswith(".g"):
                    v.fill_(1.0)
                    n += 1
            logging.warning("MLMTask.build_model(init=None): weights are zero until load_state_dict(); %d norm scales set to 1 "
                            "(the reference would apply torch's default init here). BatchNorm running statistics are plain "
                            "per-rank buffers of the flat store: a DDP wrapper does not broadcast them", n)
        return model

    @classmethod
    def build_model_from_file(cls, config_file: Union[Path, str] = None, model_file: Union[Path, str] = None,
                              device: str = "cpu", compute: str = "f32"):
        """tasks/mlm.py:446-496 (incl. the encoder.embed* -> encoder.speech_embed* key rename)."""
        if config_file is None:
            assert model_file is not None
            config_file = Path(model_file).parent / "config.yaml"
        with Path(config_file).open("r", encoding="utf-8") as f:
            args = yaml.safe_load(f)
        args["model_conf"].pop("ctc_weight", None)
        args = argparse.Namespace(**args)
        args.init = None
        model = cls.build_model(args, device=device, compute=compute)
        if model_file is not None:
            state = torch.load(model_file, map_location="cpu")
            model.load_state_dict(state)
        return model, args
