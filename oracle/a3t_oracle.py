"""CPU oracle for the A3T masked-mel hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy / plain torch-CPU functional ops, the
algorithm of the reference (richardbaihe/a3t) for the one path this repo
accelerates.  It is *the checker*: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
(``a3t_amd``) never imports it and never falls back to it.

Pinning status
--------------
* Everything except the mel filter matrix is pinned against the reference's own
  Python implementation, imported in the build container by
  ``tests/golden/make_golden.py``; the resulting vectors live in ``tests/golden``
  and ``tests/test_oracle_golden.py`` checks this file against them.  The same
  script's ``--sweep`` mode compared this file with the imported reference on
  1000 random index cases (bit-exact) and 60 random architectures (loss and all
  gradients); the record is ``tests/golden/sweep_report.json``.
* ``slaney_mel`` restates librosa>=0.8 ``filters.mel`` (un-vendored, un-pinned
  dependency of the reference: ``setup.py:32``, call site
  ``espnet2/layers/log_mel.py:49``).  librosa is not installed and there is no
  network, so the matrix itself is **parity unpinned**; everything downstream
  of it (STFT, clamp, log10, padding) is pinned with this matrix injected into
  the reference.
* The vocoder restates the vendored ``ParallelWaveGANGenerator``
  (``espnet2/gan_tts/parallel_wavegan``); the pip ``parallel_wavegan`` package
  and its pretrained weights are unavailable => pretrained-weight parity unpinned.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).
"""
import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
@dataclass
class A3TConfig:
    """Hyper-parameters of ESPnetMLMEncAsDecoderModel as built by MLMTask.build_model
    (espnet2/tasks/mlm.py:328-443) from egs2/vctk/sedit/conf/fsp2_conformer.yaml."""

    idim: int = 80            # input_size
    odim: int = 80
    vocab: int = 73
    adim: int = 384           # attention_dim
    heads: int = 2
    ff: int = 1536            # linear_units
    ff_kernel: int = 3        # positionwise_conv_kernel_size
    enc_blocks: int = 4
    dec_blocks: int = 4
    enc_kernel: int = 7       # cnn_module_kernel (encoder)
    dec_kernel: int = 31      # cnn_module_kernel (decoder)
    postnet_layers: int = 5
    postnet_chans: int = 256
    postnet_filts: int = 5
    max_len: int = 5000       # PositionalEncoding max_len
    seg_table: int = 500      # segment_emb rows
    lsm_weight: float = 0.1   # >50 selects MSE (sedit_model.py:105-108)
    # feature extraction (egs2/vctk/sedit/run.sh:10-13, mlm.sh:63-65)
    fs: int = 24000
    n_fft: int = 2048
    win_length: int = 1200
    hop_length: int = 300
    n_mels: int = 80
    fmin: float = 80.0
    fmax: float = 7600.0
    # masking (fsp2_conformer.yaml:66-72)
    mlm_prob: float = 0.8
    mean_phn_span: int = 8

    @property
    def dk(self):
        return self.adim // self.heads


def tiny_config(**kw):
    c = A3TConfig(adim=32, heads=2, ff=64, enc_blocks=1, dec_blocks=1, postnet_layers=2,
                  postnet_chans=16, vocab=11)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


# ----------------------------------------------------------------------------
# parameter naming (= the reference state_dict; SURVEY §8b, probe-dumped)
# ----------------------------------------------------------------------------
def _block_shapes(prefix: str, c: A3TConfig, K: int) -> Dict[str, tuple]:
    d, ff, kf, H, dk = c.adim, c.ff, c.ff_kernel, c.heads, c.dk
    s = {}
    a = prefix + "self_attn."
    s[a + "pos_bias_u"] = (H, dk)
    s[a + "pos_bias_v"] = (H, dk)
    for n in ("q", "k", "v", "out"):
        s[a + f"linear_{n}.weight"] = (d, d)
        s[a + f"linear_{n}.bias"] = (d,)
    s[a + "linear_pos.weight"] = (d, d)
    for f in ("feed_forward", "feed_forward_macaron"):
        s[prefix + f + ".w_1.weight"] = (ff, d, kf)
        s[prefix + f + ".w_1.bias"] = (ff,)
        s[prefix + f + ".w_2.weight"] = (d, ff, kf)
        s[prefix + f + ".w_2.bias"] = (d,)
    m = prefix + "conv_module."
    s[m + "pointwise_conv1.weight"] = (2 * d, d, 1)
    s[m + "pointwise_conv1.bias"] = (2 * d,)
    s[m + "depthwise_conv.weight"] = (d, 1, K)
    s[m + "depthwise_conv.bias"] = (d,)
    s[m + "norm.weight"] = (d,)
    s[m + "norm.bias"] = (d,)
    s[m + "norm.running_mean"] = (d,)
    s[m + "norm.running_var"] = (d,)
    s[m + "norm.num_batches_tracked"] = ()
    s[m + "pointwise_conv2.weight"] = (d, d, 1)
    s[m + "pointwise_conv2.bias"] = (d,)
    for n in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final"):
        s[prefix + n + ".weight"] = (d,)
        s[prefix + n + ".bias"] = (d,)
    return s


def param_shapes(c: A3TConfig) -> Dict[str, tuple]:
    """All state_dict entries (parameters + buffers) of the reference model."""
    d = c.adim
    s = {}
    s["encoder.segment_emb.weight"] = (c.seg_table, d)
    s["encoder.speech_embed.0.mask_feature"] = (1, 1, c.idim)
    s["encoder.speech_embed.1.weight"] = (d, c.idim)
    s["encoder.speech_embed.1.bias"] = (d,)
    s["encoder.speech_embed.2.weight"] = (d,)
    s["encoder.speech_embed.2.bias"] = (d,)
    s["encoder.text_embed.0.weight"] = (c.vocab, d)
    for i in range(c.enc_blocks):
        s.update(_block_shapes(f"encoder.encoders.{i}.", c, c.enc_kernel))
    s["encoder.after_norm.weight"] = (d,)
    s["encoder.after_norm.bias"] = (d,)
    for i in range(c.dec_blocks):
        s.update(_block_shapes(f"decoder.encoders.{i}.", c, c.dec_kernel))
    s["decoder.after_norm.weight"] = (d,)
    s["decoder.after_norm.bias"] = (d,)
    s["sfc.weight"] = (c.odim, d)
    s["sfc.bias"] = (c.odim,)
    for l in range(c.postnet_layers):
        ic = c.odim if l == 0 else c.postnet_chans
        oc = c.odim if l == c.postnet_layers - 1 else c.postnet_chans
        p = f"postnet.postnet.{l}."
        s[p + "0.weight"] = (oc, ic, c.postnet_filts)
        s[p + "1.weight"] = (oc,)
        s[p + "1.bias"] = (oc,)
        s[p + "1.running_mean"] = (oc,)
        s[p + "1.running_var"] = (oc,)
        s[p + "1.num_batches_tracked"] = ()
    return s


def procedural_state(shapes: Dict[str, tuple], seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic non-degenerate weights keyed by entry name (numpy legacy
    RandomState is frozen across numpy versions, so this reproduces everywhere).
    Non-zero biases / BN gammas are mandatory: the recipe's xavier init zeroes
    every 1-d parameter (espnet2/torch_utils/initialize.py:63-88), which would
    leave the conv module and postnet dead in a fixture."""
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if name.endswith("num_batches_tracked"):
            out[name] = np.array(3, dtype=np.int64)
        elif name.endswith("running_var"):
            out[name] = rs.uniform(0.5, 1.5, shp).astype(np.float32)
        elif name.endswith("running_mean"):
            out[name] = rs.uniform(-0.2, 0.2, shp).astype(np.float32)
        elif len(shp) == 1 and ("norm" in name or ".1.weight" in name or "speech_embed.2" in name) \
                and name.endswith("weight"):
            out[name] = (1.0 + rs.uniform(-0.2, 0.2, shp)).astype(np.float32)
        elif len(shp) == 1:
            out[name] = rs.uniform(-0.1, 0.1, shp).astype(np.float32)
        elif "emb" in name and name.endswith("weight") and len(shp) == 2:
            out[name] = rs.uniform(-0.1, 0.1, shp).astype(np.float32)
        elif "mask_feature" in name or "pos_bias" in name:
            out[name] = rs.uniform(-0.5, 0.5, shp).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = math.sqrt(3.0 / fan_in)
            out[name] = rs.uniform(-a, a, shp).astype(np.float32)
    return out


# ----------------------------------------------------------------------------
# H2-H4: log-mel features
# ----------------------------------------------------------------------------
def slaney_mel(fs: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney')
    restated (espnet2/layers/log_mel.py:37-51).  PARITY UNPINNED (see header).
    Returns (n_mels, 1 + n_fft//2) float32."""
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        m = f / f_sp
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, m)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f = f_sp * m
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)

    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, fs / 2.0, n_bins)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def logmel_fbank(wav: torch.Tensor, ilens: torch.Tensor, c: A3TConfig,
                 melmat: Optional[np.ndarray] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """LogMelFbank.forward (espnet2/tts/feats_extract/log_mel_fbank.py:88-106):
    Stft.forward (espnet2/layers/stft.py:56-124: center=True reflect pad n_fft//2,
    periodic hann of win_length zero-padded to n_fft, olens, zero-fill),
    amplitude sqrt(clamp(re^2+im^2,1e-10)), LogMel.forward
    (espnet2/layers/log_mel.py:56-83: matmul, clamp 1e-10, log10, pad->0).
    wav (B,N) f32 zero-padded, ilens (B,) -> (B,F,n_mels), olens."""
    B, N = wav.shape
    n_fft, win, hop = c.n_fft, c.win_length, c.hop_length
    pad = n_fft // 2
    x = F.pad(wav[:, None, :], (pad, pad), mode="reflect")[:, 0]
    nfr = 1 + (x.shape[1] - n_fft) // hop
    window = torch.hann_window(win, periodic=True, dtype=wav.dtype)
    lpad = (n_fft - win) // 2
    wfull = torch.zeros(n_fft, dtype=wav.dtype)
    wfull[lpad:lpad + win] = window
    frames = x.unfold(1, n_fft, hop)[:, :nfr] * wfull           # (B,F,n_fft)
    spec = torch.fft.rfft(frames, dim=-1)
    power = spec.real ** 2 + spec.imag ** 2
    olens = (ilens + 2 * (win // 2) - win) // hop + 1           # stft.py:116-121
    fidx = torch.arange(nfr)[None, :]
    padmask = fidx >= olens[:, None]
    power = power.masked_fill(padmask[..., None], 0.0)
    amp = torch.sqrt(torch.clamp(power, min=1.0e-10))
    if melmat is None:
        melmat = slaney_mel(c.fs, c.n_fft, c.n_mels, c.fmin, c.fmax)
    mel = torch.matmul(amp, torch.from_numpy(melmat.T.copy()).to(amp.dtype))
    mel = torch.clamp(mel, min=1e-10).log10()
    mel = mel.masked_fill(padmask[..., None], 0.0)
    return mel, olens


# ----------------------------------------------------------------------------
# H5-H8: alignment -> frames, span masks, segment ids
# ----------------------------------------------------------------------------
def align_to_frames(align_sec: torch.Tensor, fs: int, hop: int) -> torch.Tensor:
    """collate_fn.py:236-237: floor(fs*align/hop).int() in the tensor's own floating dtype (float32 from the dataset,
    float64 when sedit_inference.py:603-604 hands over np.array(list of Python floats))."""
    if align_sec.dtype != torch.float64:
        align_sec = align_sec.to(torch.float32)
    return torch.floor(fs * align_sec / hop).int()


def random_spans_noise_mask(length: int, mlm_prob: float, mean_span: float) -> np.ndarray:
    """collate_fn.py:387-446 (T5 random_spans_helper); consumes np.random global
    state via two np.random.shuffle calls, noise segmentation first."""
    n_noise = int(np.round(length * mlm_prob))
    n_noise = min(max(n_noise, 1), length - 1)
    n_spans = int(np.round(n_noise / mean_span))
    n_spans = max(n_spans, 1)
    n_nonnoise = length - n_noise

    def seg(num_items, num_segments):
        ind = np.arange(num_items - 1) < (num_segments - 1)
        np.random.shuffle(ind)
        first = np.concatenate([[0], ind.astype(np.int64)])
        seg_id = np.cumsum(first)
        _, lens = np.unique(seg_id, return_counts=True)
        return lens

    noise = seg(n_noise, n_spans)
    nonnoise = seg(n_nonnoise, n_spans)
    inter = np.reshape(np.stack([nonnoise, noise], axis=1), [n_spans * 2])
    starts = np.cumsum(inter)[:-1]
    ind = np.zeros((length,), dtype=np.int8)
    ind[starts] = 1
    span_num = np.cumsum(ind)
    return np.equal(span_num % 2, 1)[:length]


def phones_masking(T_mel: int, speech_nonpad: np.ndarray, align_start: np.ndarray,
                   align_end: np.ndarray, align_lens: Sequence[int], mlm_prob: float,
                   mean_phn_span: int, span_boundary=None) -> np.ndarray:
    """collate_fn.py:346-385.  speech_nonpad (B,T_mel) bool; align_* (B,P) int
    frame indices.  Returns bool (B,T_mel)."""
    B = speech_nonpad.shape[0]
    mp = np.zeros((B, T_mel))
    if mlm_prob == 1.0:
        mp += 1
    elif mean_phn_span == 0:
        length = T_mel
        span = min(length * mlm_prob // 3, 50)
        idx = random_spans_noise_mask(length, mlm_prob, span).nonzero()
        mp[:, idx] = 1
    else:
        for b in range(B):
            if span_boundary is not None:
                sb = [int(v) for v in span_boundary[b]]
                for s, e in zip(sb[::2], sb[1::2]):
                    mp[b, s:e] = 1
            else:
                L = int(align_lens[b])
                if L < 2:
                    continue
                idx = random_spans_noise_mask(L, mlm_prob, mean_phn_span).nonzero()[0]
                for p in idx:
                    mp[b, int(align_start[b][p]):int(align_end[b][p])] = 1
    mp = mp * speech_nonpad.astype(np.float64)
    return mp.astype(bool)


def get_segment_pos(T_mel: int, T_phn: int, align_start, align_end, align_lens, sega_emb=True):
    """collate_fn.py:330-343."""
    B = len(align_lens)
    sp = np.zeros((B, T_mel), dtype=np.int64)
    tp = np.zeros((B, T_phn), dtype=np.int64)
    if not sega_emb:
        return sp, tp
    for b in range(B):
        for j in range(int(align_lens[b])):
            s, e = int(align_start[b][j]), int(align_end[b][j])
            sp[b, s:e] = j + 1
            tp[b, j] = j + 1
    return sp, tp


def _pad_list(arrs: List[np.ndarray], pad_value):
    """nets_utils.py:34-61."""
    n = len(arrs)
    m = max(a.shape[0] for a in arrs)
    out = np.full((n, m) + arrs[0].shape[1:], pad_value, dtype=arrs[0].dtype)
    for i, a in enumerate(arrs):
        out[i, : a.shape[0]] = a
    return out


def collate(data: List[Tuple[str, Dict[str, np.ndarray]]], c: A3TConfig,
            melmat: Optional[np.ndarray] = None, sega_emb: bool = True):
    """mlm_collate_fn (collate_fn.py:158-287) for the text+alignment case."""
    uids = [u for u, _ in data]
    ds = [d for _, d in data]
    speech = _pad_list([d["speech"] for d in ds], 0.0)
    slen = np.array([d["speech"].shape[0] for d in ds], dtype=np.int64)
    text = _pad_list([d["text"] for d in ds], 0)
    tlen = np.array([d["text"].shape[0] for d in ds], dtype=np.int64)
    a_s = _pad_list([d["align_start"] for d in ds], 0.0)
    a_e = _pad_list([d["align_end"] for d in ds], 0.0)
    alen = np.array([d["align_start"].shape[0] for d in ds], dtype=np.int64)
    feats, flen = logmel_fbank(torch.from_numpy(speech), torch.from_numpy(slen), c, melmat)
    fs_ = align_to_frames(torch.from_numpy(a_s), c.fs, c.hop_length).numpy()
    fe_ = align_to_frames(torch.from_numpy(a_e), c.fs, c.hop_length).numpy()
    max_slen = int(flen.max())
    speech_pad = feats[:, :max_slen]
    T_phn = text.shape[1]
    text_mask = (np.arange(T_phn)[None, :] < tlen[:, None])
    speech_mask = (np.arange(max_slen)[None, :] < flen.numpy()[:, None])
    span_boundary = None
    if "span_boundary" in ds[0]:
        span_boundary = _pad_list([d["span_boundary"] for d in ds], 0)
    masked = phones_masking(max_slen, speech_mask, fs_, fe_, alen, c.mlm_prob,
                            c.mean_phn_span, span_boundary)
    sp, tp = get_segment_pos(max_slen, T_phn, fs_, fe_, alen, sega_emb)
    out = dict(
        speech=speech_pad,
        text=torch.from_numpy(text),
        masked_position=torch.from_numpy(masked),
        speech_mask=torch.from_numpy(speech_mask)[:, None, :],
        text_mask=torch.from_numpy(text_mask)[:, None, :],
        speech_segment_pos=torch.from_numpy(sp),
        text_segment_pos=torch.from_numpy(tp),
        speech_lengths=torch.from_numpy(slen),
        text_lengths=torch.from_numpy(tlen),
    )
    return uids, out


# ----------------------------------------------------------------------------
# device-side model, restated with plain functional torch (CPU fp32/fp64)
# ----------------------------------------------------------------------------
def legacy_pe(c: A3TConfig, T: int, dtype=torch.float32) -> torch.Tensor:
    """LegacyRelPositionalEncoding table rows 0..T-1: pe[t] = PE(max_len-1-t)
    (transformer/embedding.py:59-80 with reverse=True, :147-170)."""
    d = c.adim
    position = torch.arange(c.max_len - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(c.max_len, d)
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe[:T].to(dtype)


def rel_shift_legacy(x: torch.Tensor) -> torch.Tensor:
    """LegacyRelPositionMultiHeadedAttention.rel_shift (transformer/attention.py:145-165)."""
    B, H, T1, T2 = x.shape
    xp = torch.cat([x.new_zeros(B, H, T1, 1), x], dim=-1)
    xp = xp.view(B, H, T2 + 1, T1)
    return xp[:, :, 1:].view_as(x)


def _ln(x, p, pre, eps):
    return F.layer_norm(x, (x.shape[-1],), p[pre + ".weight"], p[pre + ".bias"], eps)


def attention(x, pos, mask, p, pre, c: A3TConfig, return_probs=False):
    """LegacyRelPositionMultiHeadedAttention.forward (attention.py:167-209) +
    forward_qkv/forward_attention (:40-96).  x (B,T,d), pos (1,T,d), mask (B,1,T) bool."""
    B, T, d = x.shape
    H, dk = c.heads, c.dk
    q = F.linear(x, p[pre + "linear_q.weight"], p[pre + "linear_q.bias"]).view(B, T, H, dk)
    k = F.linear(x, p[pre + "linear_k.weight"], p[pre + "linear_k.bias"]).view(B, T, H, dk).transpose(1, 2)
    v = F.linear(x, p[pre + "linear_v.weight"], p[pre + "linear_v.bias"]).view(B, T, H, dk).transpose(1, 2)
    pp = F.linear(pos, p[pre + "linear_pos.weight"]).view(1, -1, H, dk).transpose(1, 2)
    qu = (q + p[pre + "pos_bias_u"]).transpose(1, 2)
    qv = (q + p[pre + "pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(qu, k.transpose(-2, -1))
    bd = rel_shift_legacy(torch.matmul(qv, pp.transpose(-2, -1)))
    scores = (ac + bd) / math.sqrt(dk)
    if scores.dtype == torch.bfloat16:
        # only under torch.autocast("cpu", bfloat16) (the bf16 yardstick of the tests): bf16 logits, fp32 softmax -- the same
        # promotion tests/golden/make_golden.py::gen_bf16ref applies to the reference, whose attention.py:81 cannot take bf16
        scores = scores.float()
    m = mask.unsqueeze(1).eq(0)
    min_value = float(np.finfo(np.float32).min) if scores.dtype == torch.float32 else float(np.finfo(np.float64).min)
    scores = scores.masked_fill(m, min_value)
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    ctx = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, T, d)
    out = F.linear(ctx, p[pre + "linear_out.weight"], p[pre + "linear_out.bias"])
    if return_probs:
        return out, attn
    return out


def ffn_conv(x, p, pre, c: A3TConfig):
    """MultiLayeredConv1d.forward (transformer/multi_layer_conv.py:52-63), dropout p=0."""
    pad = (c.ff_kernel - 1) // 2
    h = torch.relu(F.conv1d(x.transpose(1, 2), p[pre + "w_1.weight"], p[pre + "w_1.bias"], padding=pad))
    return F.conv1d(h, p[pre + "w_2.weight"], p[pre + "w_2.bias"], padding=pad).transpose(1, 2)


def _batch_norm(x, p, pre, train: bool, stats_out: Optional[dict] = None):
    """torch.nn.BatchNorm1d over (B,C,T): train = batch statistics incl. padded
    frames (biased var for normalisation; running stats momentum 0.1 / unbiased)."""
    w, b = p[pre + ".weight"], p[pre + ".bias"]
    if train:
        mean = x.mean(dim=(0, 2))
        var = x.var(dim=(0, 2), unbiased=False)
        if stats_out is not None:
            n = x.shape[0] * x.shape[2]
            stats_out[pre] = (mean.detach(), (var * n / max(n - 1, 1)).detach())
    else:
        mean, var = p[pre + ".running_mean"], p[pre + ".running_var"]
    return (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5) * w[None, :, None] + b[None, :, None]


def conv_module(x, p, pre, K: int, train_bn: bool, stats_out=None):
    """ConvolutionModule.forward (conformer/convolution.py:56-79) with Swish
    (conformer/swish.py:16-18)."""
    y = x.transpose(1, 2)
    y = F.conv1d(y, p[pre + "pointwise_conv1.weight"], p[pre + "pointwise_conv1.bias"])
    y = F.glu(y, dim=1)
    y = F.conv1d(y, p[pre + "depthwise_conv.weight"], p[pre + "depthwise_conv.bias"],
                 padding=(K - 1) // 2, groups=y.shape[1])
    y = _batch_norm(y, p, pre + "norm", train_bn, stats_out)
    y = y * torch.sigmoid(y)
    y = F.conv1d(y, p[pre + "pointwise_conv2.weight"], p[pre + "pointwise_conv2.bias"])
    return y.transpose(1, 2)


def conformer_block(x, pos, mask, p, pre, c: A3TConfig, K: int, train_bn: bool, stats_out=None):
    """EncoderLayer.forward (conformer/encoder_layer.py:80-180): macaron, pre-norm,
    dropout p=0, stochastic depth 0."""
    x = x + 0.5 * ffn_conv(_ln(x, p, pre + "norm_ff_macaron", 1e-12), p, pre + "feed_forward_macaron.", c)
    x = x + attention(_ln(x, p, pre + "norm_mha", 1e-12), pos, mask, p, pre + "self_attn.", c)
    x = x + conv_module(_ln(x, p, pre + "norm_conv", 1e-12), p, pre + "conv_module.", K, train_bn, stats_out)
    x = x + 0.5 * ffn_conv(_ln(x, p, pre + "norm_ff", 1e-12), p, pre + "feed_forward.", c)
    return _ln(x, p, pre + "norm_final", 1e-12)


def encoder_embed(batch, p, c: A3TConfig):
    """MLMEncoder.forward prologue, input_layer='sega_mlm' (conformer/encoder.py:522-553;
    NewMaskInputLayer espnet2/asr/encoder/mlm_encoder.py:57-70)."""
    speech = batch["speech"]
    dtype = speech.dtype
    mpos = batch["masked_position"].bool()[..., None]
    x = speech.masked_fill(mpos, 0) + p["encoder.speech_embed.0.mask_feature"].expand_as(speech).masked_fill(~mpos, 0)
    x = F.linear(x, p["encoder.speech_embed.1.weight"], p["encoder.speech_embed.1.bias"])
    x = F.layer_norm(x, (c.adim,), p["encoder.speech_embed.2.weight"], p["encoder.speech_embed.2.bias"], 1e-5)
    x = torch.relu(x) * math.sqrt(c.adim)
    Tm = speech.shape[1]
    text = batch["text"]
    Tp = text.shape[1]
    V = p["encoder.text_embed.0.weight"].shape[0]
    xt = F.embedding(text, p["encoder.text_embed.0.weight"], padding_idx=V - 1) * math.sqrt(c.adim)
    seg = p["encoder.segment_emb.weight"]
    x = x + F.embedding(batch["speech_segment_pos"], seg, padding_idx=c.seg_table - 1)
    xt = xt + F.embedding(batch["text_segment_pos"], seg, padding_idx=c.seg_table - 1)
    xs = torch.cat([x, xt], dim=1)
    pos = torch.cat([legacy_pe(c, Tm, dtype), legacy_pe(c, Tp, dtype)], dim=0)[None]
    masks = torch.cat([batch["speech_mask"], batch["text_mask"]], dim=-1)
    return xs, pos, masks


def model_forward(p: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], c: A3TConfig,
                  train_bn: bool = True, stats_out=None):
    """ESPnetMLMEncAsDecoderModel._forward (espnet2/tts/sedit/sedit_model.py:350-375):
    encoder -> decoder -> slice -> sfc -> postnet.  Returns before, after (B,T_mel,odim)."""
    xs, pos, masks = encoder_embed(batch, p, c)
    for i in range(c.enc_blocks):
        xs = conformer_block(xs, pos, masks, p, f"encoder.encoders.{i}.", c, c.enc_kernel, train_bn, stats_out)
    xs = _ln(xs, p, "encoder.after_norm", 1e-12)
    # MLMDecoder.forward (conformer/encoder.py:568-614), embed = LegacyRelPositionalEncoding
    T = xs.shape[1]
    xs = xs * math.sqrt(c.adim)
    pos = legacy_pe(c, T, xs.dtype)[None]
    for i in range(c.dec_blocks):
        xs = conformer_block(xs, pos, masks, p, f"decoder.encoders.{i}.", c, c.dec_kernel, train_bn, stats_out)
    xs = _ln(xs, p, "decoder.after_norm", 1e-12)
    Tm = batch["speech"].shape[1]
    hs = xs[:, :Tm].contiguous()
    before = F.linear(hs, p["sfc.weight"], p["sfc.bias"])
    if c.postnet_layers == 0:     # sedit_model.py:111-122, :369-374: no Postnet module -> after_outs is None
        return before, None
    # Postnet (tacotron2/decoder.py:150-267), dropout p=0
    y = before.transpose(1, 2)
    for l in range(c.postnet_layers):
        pre = f"postnet.postnet.{l}."
        y = F.conv1d(y, p[pre + "0.weight"], None, padding=(c.postnet_filts - 1) // 2)
        y = _batch_norm(y, p, pre + "1", train_bn, stats_out)
        if l != c.postnet_layers - 1:
            y = torch.tanh(y)
    after = before + y.transpose(1, 2)
    return before, after


def mlm_loss(before, after, target, masked_position, c: A3TConfig):
    """ESPnetMLMModel._calc_mlm_loss (sedit_model.py:320-340); the after-postnet term exists only when there is a
    postnet (:331-333)."""
    sq = c.lsm_weight > 50
    l = ((before - target) ** 2).sum(-1) if sq else (before - target).abs().sum(-1)
    if after is not None:
        l = l + (((after - target) ** 2).sum(-1) if sq else (after - target).abs().sum(-1))
    m = masked_position.to(l.dtype)
    return (l * m).sum() / (m.sum() + 1e-10)


def forward_loss(p, batch, c: A3TConfig, train_bn=True, stats_out=None):
    before, after = model_forward(p, batch, c, train_bn, stats_out)
    loss = mlm_loss(before, after, batch["speech"], batch["masked_position"], c)
    return loss, before, after


def inference_splice(p, batch, c: A3TConfig, span: Tuple[int, int]):
    """ESPnetMLMModel.inference teacher-forcing branch (sedit_model.py:274-284);
    model in eval mode (BN running stats)."""
    before, after = model_forward(p, batch, c, train_bn=False)
    s, e = span
    return torch.cat([batch["speech"][0, :s], after[0, s:e], batch["speech"][0, e:]], dim=0)


# ----------------------------------------------------------------------------
# trainer step restatement (D23): clip + Adam + NoamLR
# ----------------------------------------------------------------------------
def noam_lr(step_num: int, base_lr: float, model_size: int, warmup: int) -> float:
    """NoamLR.get_lr (espnet2/schedulers/noam_lr.py:58-65), step_num = last_epoch+1."""
    return base_lr * model_size ** -0.5 * min(step_num ** -0.5, step_num * warmup ** -1.5)


def clip_adam_step(params, grads, m, v, step: int, lr: float, clip: float = 1.0,
                   betas=(0.9, 0.999), eps=1e-8):
    """trainer.py:631-679 with torch.optim.Adam defaults: clip_grad_norm_(max_norm=clip,
    L2): coef = clip/(norm+1e-6) clamped to 1; skip the update when norm is non-finite.
    Operates in place on lists of tensors; returns grad_norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    if not torch.isfinite(total):
        return total
    coef = torch.clamp(clip / (total + 1e-6), max=1.0)
    b1, b2 = betas
    for p_, g, m_, v_ in zip(params, grads, m, v):
        g = g * coef
        m_.mul_(b1).add_(g, alpha=1 - b1)
        v_.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
        p_.addcdiv_(m_, denom, value=-lr / bc1)
    return total


# ----------------------------------------------------------------------------
# ParallelWaveGAN generator (vendored twin) V1-V6
# ----------------------------------------------------------------------------
@dataclass
class PWGConfig:
    """ParallelWaveGANGenerator defaults (espnet2/gan_tts/parallel_wavegan/parallel_wavegan.py:29-47)
    with the recipe's hop 300 = 4*5*3*5."""
    layers: int = 30
    stacks: int = 3
    res_ch: int = 64
    gate_ch: int = 128
    skip_ch: int = 64
    aux_ch: int = 80
    aux_context_window: int = 2
    kernel_size: int = 3
    upsample_scales: Tuple[int, ...] = (4, 5, 3, 5)


def pwg_param_shapes(c: PWGConfig) -> Dict[str, tuple]:
    """state_dict of the generator after remove_weight_norm()."""
    s = {}
    s["first_conv.weight"] = (c.res_ch, 1, 1)
    s["first_conv.bias"] = (c.res_ch,)
    s["upsample_net.conv_in.weight"] = (c.aux_ch, c.aux_ch, 2 * c.aux_context_window + 1)
    for i, sc in enumerate(c.upsample_scales):
        s[f"upsample_net.upsample.up_layers.{2 * i + 1}.weight"] = (1, 1, 1, 2 * sc + 1)
    for l in range(c.layers):
        pre = f"conv_layers.{l}."
        s[pre + "conv.weight"] = (c.gate_ch, c.res_ch, c.kernel_size)
        s[pre + "conv.bias"] = (c.gate_ch,)
        s[pre + "conv1x1_aux.weight"] = (c.gate_ch, c.aux_ch, 1)
        s[pre + "conv1x1_out.weight"] = (c.res_ch + c.skip_ch, c.gate_ch // 2, 1)
        s[pre + "conv1x1_out.bias"] = (c.res_ch + c.skip_ch,)
    s["last_conv_layers.1.weight"] = (c.skip_ch, c.skip_ch, 1)
    s["last_conv_layers.1.bias"] = (c.skip_ch,)
    s["last_conv_layers.3.weight"] = (1, c.skip_ch, 1)
    s["last_conv_layers.3.bias"] = (1,)
    return s


def pwg_forward(p: Dict[str, torch.Tensor], c_feats: torch.Tensor, z: torch.Tensor, cfg: PWGConfig,
                taps: Optional[dict] = None) -> torch.Tensor:
    """ParallelWaveGANGenerator.forward (parallel_wavegan.py:136-177), ResidualBlock.forward
    (wavenet/residual_block.py:114-169), ConvInUpsampleNetwork/UpsampleNetwork/Stretch2d
    (parallel_wavegan/upsample.py:22-189).  c_feats (B,80,T_feats), z (B,1,T_wav)."""
    w = cfg.aux_context_window
    c = F.pad(c_feats, (w, w), mode="replicate")
    c = F.conv1d(c, p["upsample_net.conv_in.weight"])
    c = c.unsqueeze(1)
    for i, sc in enumerate(cfg.upsample_scales):
        c = F.interpolate(c, scale_factor=(1, sc), mode="nearest")
        c = F.conv2d(c, p[f"upsample_net.upsample.up_layers.{2 * i + 1}.weight"], padding=(0, sc))
    c = c.squeeze(1)
    x = F.conv1d(z, p["first_conv.weight"], p["first_conv.bias"])
    skips = 0
    lps = cfg.layers // cfg.stacks
    for l in range(cfg.layers):
        pre = f"conv_layers.{l}."
        dil = 2 ** (l % lps)
        res = x
        y = F.conv1d(x, p[pre + "conv.weight"], p[pre + "conv.bias"],
                     padding=(cfg.kernel_size - 1) // 2 * dil, dilation=dil)
        xa, xb = y.split(y.shape[1] // 2, dim=1)
        ca, cb = F.conv1d(c, p[pre + "conv1x1_aux.weight"]).split(y.shape[1] // 2, dim=1)
        g = torch.tanh(xa + ca) * torch.sigmoid(xb + cb)
        o = F.conv1d(g, p[pre + "conv1x1_out.weight"], p[pre + "conv1x1_out.bias"])
        r, s = o.split([cfg.res_ch, cfg.skip_ch], dim=1)
        x = (r + res) * math.sqrt(0.5)
        skips = skips + s
        if taps is not None and l < 2:
            taps[f"x{l}"] = x
            taps[f"skip{l}"] = s
    skips = skips * math.sqrt(1.0 / cfg.layers)
    x = torch.relu(skips)
    x = torch.relu(F.conv1d(x, p["last_conv_layers.1.weight"], p["last_conv_layers.1.bias"]))
    return F.conv1d(x, p["last_conv_layers.3.weight"], p["last_conv_layers.3.bias"])


# ----------------------------------------------------------------------------
# helpers shared by tests / bench
# ----------------------------------------------------------------------------
def to_torch_state(state: Dict[str, np.ndarray], dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in state.items():
        t = torch.from_numpy(np.array(v))
        if t.is_floating_point():
            t = t.to(dtype)
            if requires_grad and not k.endswith(("running_mean", "running_var")):
                t.requires_grad_(True)
        out[k] = t
    return out


def synthetic_batch(c: A3TConfig, B: int, T_mel: int, T_phn: int, seed: int,
                    lengths: Optional[Sequence[int]] = None, text_lengths: Optional[Sequence[int]] = None):
    """SURVEY §8(d) synthetic training inputs: log-mel-like frames, phones tiling
    the utterance, phoneme-span mask from phones_masking under np.random.seed(seed)."""
    rs = np.random.RandomState(seed)
    speech = np.clip(rs.standard_normal((B, T_mel, c.idim)) * 1.5 - 4.0, -10, 2).astype(np.float32)
    text = rs.randint(2, c.vocab - 1, size=(B, T_phn)).astype(np.int64)
    lengths = [T_mel] * B if lengths is None else list(lengths)
    text_lengths = [T_phn] * B if text_lengths is None else list(text_lengths)
    a_s = np.zeros((B, T_phn), dtype=np.int32)
    a_e = np.zeros((B, T_phn), dtype=np.int32)
    for b in range(B):
        P, L = text_lengths[b], lengths[b]
        cuts = np.sort(rs.choice(np.arange(1, L), size=P - 1, replace=False)) if P > 1 else np.array([], dtype=np.int64)
        bounds = np.concatenate([[0], cuts, [L]])
        a_s[b, :P] = bounds[:-1]
        a_e[b, :P] = bounds[1:]
        speech[b, L:] = 0.0
        text[b, P:] = 0
    smask = np.arange(T_mel)[None, :] < np.array(lengths)[:, None]
    tmask = np.arange(T_phn)[None, :] < np.array(text_lengths)[:, None]
    np.random.seed(seed)
    masked = phones_masking(T_mel, smask, a_s, a_e, text_lengths, c.mlm_prob, c.mean_phn_span)
    sp, tp = get_segment_pos(T_mel, T_phn, a_s, a_e, text_lengths, True)
    return dict(
        speech=torch.from_numpy(speech), text=torch.from_numpy(text),
        masked_position=torch.from_numpy(masked),
        speech_mask=torch.from_numpy(smask)[:, None, :], text_mask=torch.from_numpy(tmask)[:, None, :],
        speech_segment_pos=torch.from_numpy(sp), text_segment_pos=torch.from_numpy(tp),
    )


# ----------------------------------------------------------------------------
# speech-editing inference driver: host-side span arithmetic (espnet2/bin/sedit_inference.py)
# The forced aligner (HTK), the phoneme dictionary tool and the FastSpeech2 duration predictor are external programs /
# checkpoints of the reference; their OUTPUTS are the inputs here.
# ----------------------------------------------------------------------------
def sedit_masked_mel_boundary(mfa_start, mfa_end, fs: int, hop: int, span):
    """sedit_inference.py:426-435 -- phone span [a, b) -> mel-frame span; an empty span past the end pins to the end."""
    a_s = torch.floor(fs * torch.tensor(mfa_start) / hop).int().tolist()
    a_e = torch.floor(fs * torch.tensor(mfa_end) / hop).int().tolist()
    if span[0] >= len(mfa_start):
        return [a_e[-1], a_e[-1]]
    return [a_s[span[0]], a_e[span[1] - 1]]


def sedit_phone_spans(times2, word2phns: Dict[str, str], new_phns: List[str], new_word2phns: Dict[str, List[str]],
                      old_str: str, new_str: str):
    """sedit_inference.py:437-504 (get_phns_and_spans) with the aligner output (times2 = [[phn, start, end], ...],
    word2phns = {"i_WORD": "PH PH"}) and the phonemiser output (new_phns, new_word2phns = {"i_WORD": [PH, ...]})
    passed in.  Word-level common prefix / suffix -> phone spans to be replaced (old) and added (new)."""
    append_new = (old_str == new_str[:len(old_str)])
    old_phns = [t[0] for t in times2]
    mfa_start = [float(t[1]) for t in times2]
    mfa_end = [float(t[2]) for t in times2]
    rep = [0, len(old_phns) - 1]
    add = [0, len(new_phns) - 1]
    left_index, left, sp = 0, [], 0
    for key in word2phns.keys():
        idx, wrd = key.split("_")
        if wrd == "sp":
            sp += 1
            left.append("sp")
        else:
            k2 = str(int(idx) - sp) + "_" + wrd
            if k2 in new_word2phns:
                left_index += len(new_word2phns[k2])
                left.extend(word2phns[key].split())
            else:
                rep[0] = len(left)
                add[0] = len(left)
                break
    right_index, right, sp = 0, [], 0
    wmax = int(list(word2phns.keys())[-1].split("_")[0])
    nmax = int(list(new_word2phns.keys())[-1].split("_")[0])
    middle = []
    if append_new:
        middle = new_phns[left_index:]
        rep[0] = len(left)
        add[0] = len(left)
        add[1] = len(left) + len(middle)
        rep[1] = len(old_phns) - len(right)
    else:
        for key in list(word2phns.keys())[::-1]:
            idx, wrd = key.split("_")
            if wrd == "sp":
                sp += 1
                right = ["sp"] + right
            else:
                k2 = str(nmax - (wmax - int(idx) - sp)) + "_" + wrd
                if k2 in new_word2phns:
                    right_index -= len(new_word2phns[k2])
                    right = word2phns[key].split() + right
                else:
                    rep[1] = len(old_phns) - len(right)
                    middle = new_phns[left_index:right_index]
                    add[1] = len(left) + len(middle)
                    if len(middle) == 0:
                        add[1] = min(add[1] + 1, len(new_phns))
                        add[0] = max(0, add[0] - 1)
                        rep[0] = max(0, rep[0] - 1)
                        rep[1] = min(rep[1] + 1, len(old_phns))
                    break
    return mfa_start, mfa_end, old_phns, left + middle + right, rep, add


def sedit_duration_adjust_factor(original_dur, pred_dur, phns) -> float:
    """sedit_inference.py:506-524 -- trimmed mean (2 lowest / 2 highest dropped) of aligned/predicted duration ratios."""
    f = []
    for ori, pred, phn in zip(original_dur, pred_dur, phns):
        if pred == 0 or phn == "sp":
            continue
        f.append(ori / pred)
    f = np.array(f)
    f.sort()
    if len(f) < 5:
        return 1
    return np.average(f[2:-2])


def sedit_plan_edit(wav_org: np.ndarray, fs: int, hop: int, mfa_start, mfa_end, old_phns, new_phns, rep, add,
                    duration_fn, new_str: str, mask_reconstruct=False, duration_adjust=True, start_end_sp=False):
    """sedit_inference.py:526-594 (prepare_features_with_duration) after get_phns_and_spans; duration_fn(phns) stands
    for the FastSpeech2 duration predictor (:398-424, seconds per phone).  Returns (new_wav, new_phns, new_mfa_start,
    new_mfa_end, old_span_boundary, new_span_boundary)."""
    mfa_start, mfa_end, new_phns = list(mfa_start), list(mfa_end), list(new_phns)
    rep, add = list(rep), list(add)
    if start_end_sp and new_phns[-1] != "sp":
        new_phns = new_phns + ["sp"]
    if "[MASK]" in new_str and mask_reconstruct:
        ob = sedit_masked_mel_boundary(mfa_start, mfa_end, fs, hop, rep)
        return wav_org, old_phns, mfa_start, mfa_end, ob, ob
    old_dur = duration_fn(old_phns)
    orig = [e - s for e, s in zip(mfa_end, mfa_start)]
    if "[MASK]" in new_str:
        new_phns = old_phns
        add = rep
        fl = sedit_duration_adjust_factor(orig[:rep[0]], old_dur[:rep[0]], old_phns[:rep[0]])
        fr = sedit_duration_adjust_factor(orig[rep[1]:], old_dur[rep[1]:], old_phns[rep[1]:])
        d = (fl + fr) / 2
        new_dur = [d * i for i in old_dur]
    else:
        d = sedit_duration_adjust_factor(orig, old_dur, old_phns) if duration_adjust else 1
        new_dur = [d * i for i in duration_fn(new_phns)]
        if rep[0] < len(old_phns) and old_phns[rep[0]] == new_phns[add[0]]:
            new_dur[add[0]] = orig[rep[0]]
        if rep[1] < len(old_phns) and add[1] < len(new_phns):
            if old_phns[rep[1]] == new_phns[add[1]]:
                new_dur[add[1]] = orig[rep[1]]
    new_sum = sum(new_dur[add[0]:add[1]])
    old_sum = sum(orig[rep[0]:rep[1]])
    off = new_sum - old_sum
    ns, ne = mfa_start[:rep[0]], mfa_end[:rep[0]]
    for i in new_dur[add[0]:add[1]]:
        if len(ne) == 0:
            ns.append(0)
            ne.append(i)
        else:
            ns.append(ne[-1])
            ne.append(ne[-1] + i)
    ns += [i + off for i in mfa_start[rep[1]:]]
    ne += [i + off for i in mfa_end[rep[1]:]]
    if rep[0] >= len(mfa_start):
        li = ri = len(wav_org)
    else:
        li = int(np.floor(mfa_start[rep[0]] * fs))
        ri = int(np.ceil(mfa_end[rep[1] - 1] * fs))
    blank = np.zeros((int(np.ceil(new_sum * fs)),), dtype=wav_org.dtype)
    new_wav = np.concatenate([wav_org[:li], blank, wav_org[ri:]])
    ob = sedit_masked_mel_boundary(mfa_start, mfa_end, fs, hop, rep)
    nb = sedit_masked_mel_boundary(ns, ne, fs, hop, add)
    return new_wav, new_phns, ns, ne, ob, nb


def sedit_splice_feat_gen(output: List[torch.Tensor]) -> torch.Tensor:
    """sedit_inference.py:622-629 -- [left (1,s,80) | generated (e-s,80) | right (1,T-e,80)] -> (T,80), empty ends dropped."""
    if 0 in output[0].shape and 0 not in output[-1].shape:
        return torch.cat(output[1:-1] + [output[-1].squeeze()], dim=0)
    if 0 not in output[0].shape and 0 in output[-1].shape:
        return torch.cat([output[0].squeeze()] + output[1:-1], dim=0)
    if 0 in output[0].shape and 0 in output[-1].shape:
        return torch.cat(output[1:-1], dim=0)
    return torch.cat([output[0].squeeze(0)] + output[1:-1] + [output[-1].squeeze(0)], dim=0)


def sedit_replace_waveform(wav_org: np.ndarray, replaced_wav: np.ndarray, hop: int, old_span, new_span) -> np.ndarray:
    """sedit_inference.py:80-84 -- original audio with the vocoded new span spliced in (sample = frame * hop)."""
    return np.concatenate([wav_org[:hop * old_span[0]], replaced_wav[hop * new_span[0]:hop * new_span[1]],
                           wav_org[hop * old_span[1]:]])
