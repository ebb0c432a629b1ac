"""Hyper-parameters of the A3T masked-mel model (host-side mirror of what
MLMTask.build_model reads from the recipe yaml: espnet2/tasks/mlm.py:328-443,
egs2/vctk/sedit/conf/fsp2_conformer.yaml:27-75)."""
from dataclasses import dataclass
from typing import Any, Dict


@dataclass
class A3TConfig:
    idim: int = 80
    odim: int = 80
    vocab: int = 73
    adim: int = 384
    heads: int = 2
    ff: int = 1536
    ff_kernel: int = 3
    enc_blocks: int = 4
    dec_blocks: int = 4
    enc_kernel: int = 7
    dec_kernel: int = 31
    postnet_layers: int = 5
    postnet_chans: int = 256
    postnet_filts: int = 5
    max_len: int = 5000
    seg_table: int = 500
    lsm_weight: float = 0.1
    dropout_rate: float = 0.2
    positional_dropout_rate: float = 0.2
    attention_dropout_rate: float = 0.2
    postnet_dropout_rate: float = 0.5
    # speaker conditioning (BASELINE configs[3] "+ x-vector cond"): spembs (B, spk_embed_dim) -> Linear -> added to every token
    # after the embedding prologue.  The reference accepts `spembs` and ignores it (sedit_model.py:246): this is an
    # extension with no reference behaviour (parity unpinned); 0 = off, the reference's behaviour.
    spk_embed_dim: int = 0
    # feature extraction
    fs: int = 24000
    n_fft: int = 2048
    win_length: int = 1200
    hop_length: int = 300
    n_mels: int = 80
    fmin: float = 80.0
    fmax: float = 7600.0
    # masking
    mlm_prob: float = 0.8
    mean_phn_span: int = 8

    @property
    def dk(self) -> int:
        return self.adim // self.heads

    @staticmethod
    def from_espnet(encoder_conf: Dict[str, Any], decoder_conf: Dict[str, Any], model_conf: Dict[str, Any],
                    input_size: int, odim: int, vocab: int, feats_conf: Dict[str, Any] = None) -> "A3TConfig":
        """Translate the recipe's encoder_conf/decoder_conf/model_conf dictionaries."""
        e, d, m = encoder_conf, decoder_conf, model_conf
        if e.get("input_layer", "sega_mlm") != "sega_mlm":
            raise NotImplementedError("only input_layer='sega_mlm' (the A3T recipe) is implemented")
        if e.get("positionwise_layer_type", "conv1d") != "conv1d" or not e.get("macaron_style", True) \
                or not e.get("use_cnn_module", True):
            raise NotImplementedError("only the recipe's conformer layout (conv1d FFN, macaron, cnn module)")
        c = A3TConfig(
            idim=input_size, odim=odim, vocab=vocab, adim=e["attention_dim"], heads=e["attention_heads"],
            ff=e["linear_units"], ff_kernel=e.get("positionwise_conv_kernel_size", 3), enc_blocks=e["num_blocks"],
            dec_blocks=d["num_blocks"], enc_kernel=e.get("cnn_module_kernel", 31),
            dec_kernel=d.get("cnn_module_kernel", 31), postnet_layers=m.get("postnet_layers", 0),
            postnet_chans=m.get("postnet_chans", 0), postnet_filts=m.get("postnet_filts", 0),
            lsm_weight=m.get("lsm_weight", 0.0), dropout_rate=e.get("dropout_rate", 0.1),
            positional_dropout_rate=e.get("positional_dropout_rate", 0.1),
            attention_dropout_rate=e.get("attention_dropout_rate", 0.0), mlm_prob=m.get("mlm_prob", 0.25),
            mean_phn_span=m.get("mean_phn_span", 3))
        if feats_conf:
            for k in ("fs", "n_fft", "win_length", "hop_length", "n_mels", "fmin", "fmax"):
                if k in feats_conf and feats_conf[k] is not None:
                    setattr(c, k, feats_conf[k])
        return c


def config_c4(**kw) -> A3TConfig:
    """BASELINE.json configs[3]: LibriTTS multi-speaker, 6 + 6 blocks, d=512, H=4 (d_k=128), ff=2048, x-vector (512-d)
    conditioning."""
    c = A3TConfig(adim=512, heads=4, ff=2048, enc_blocks=6, dec_blocks=6, spk_embed_dim=512)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def config_c2(**kw) -> A3TConfig:
    """BASELINE.json configs[1]: '12-layer' = 6 encoder + 6 decoder blocks, d=384."""
    c = A3TConfig(enc_blocks=6, dec_blocks=6)
    for k, v in kw.items():
        setattr(c, k, v)
    return c
