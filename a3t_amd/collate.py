"""Host side of batch construction: the mirror of MLMCollateFn
(espnet2/train/collate_fn.py:106-287) with the same call signature and output dictionary.

The index work (alignment seconds -> frame indices, T5-style phoneme-span masking driven by the
numpy global RNG, segment ids) is inherently host-side in the reference too (DataLoader worker);
it is restated here in vectorised numpy.  Feature extraction (STFT -> mel -> log10) runs on the
GPU through liba3t_hip (a3t_amd/features.py) when a device is given.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .config import A3TConfig


def random_spans_noise_mask(length: int, mlm_prob: float, mean_span: float) -> np.ndarray:
    """collate_fn.py:387-446.  Consumes the numpy global RNG exactly like the reference:
    two np.random.shuffle calls (noise spans first, then non-noise spans)."""
    n_noise = min(max(int(np.round(length * mlm_prob)), 1), length - 1)
    n_spans = max(int(np.round(n_noise / mean_span)), 1)

    def segment_lengths(num_items, num_segments):
        first = np.arange(num_items - 1) < (num_segments - 1)
        np.random.shuffle(first)
        starts = np.flatnonzero(np.concatenate([[True], first]))
        return np.diff(np.concatenate([starts, [num_items]]))

    noise = segment_lengths(n_noise, n_spans)
    nonnoise = segment_lengths(length - n_noise, n_spans)
    inter = np.stack([nonnoise, noise], axis=1).reshape(-1)
    ends = np.cumsum(inter)
    is_noise = np.zeros(length, dtype=bool)
    for s, e in zip(ends[0::2], ends[1::2]):          # odd-numbered spans are noise
        is_noise[s:e] = True
    return is_noise


def align_to_frames(align_sec: np.ndarray, fs: int, hop: int) -> np.ndarray:
    """collate_fn.py:236-237 -- `floor(fs * t / hop)` in the dtype of the alignment array and the reference's order
    (fs*t first, then /hop): float32 for dataset arrays, float64 for the arrays sedit_inference.py:603-604 builds from
    Python floats (boundaries a hair below a frame edge land one frame apart in the two precisions)."""
    a = np.asarray(align_sec)
    if a.dtype != np.float64:
        a = a.astype(np.float32)
    ft = a.dtype.type
    return np.floor((ft(fs) * a) / ft(hop)).astype(np.int32)


def phones_masking(T_mel: int, speech_nonpad: np.ndarray, align_start: np.ndarray, align_end: np.ndarray,
                   align_lens: Sequence[int], mlm_prob: float, mean_phn_span: int,
                   span_boundary: Optional[np.ndarray] = None) -> np.ndarray:
    """collate_fn.py:346-385 -> bool (B, T_mel)."""
    B = speech_nonpad.shape[0]
    mp = np.zeros((B, T_mel), dtype=bool)
    if mlm_prob == 1.0:
        mp[:] = True
    elif mean_phn_span == 0:
        span = min(T_mel * mlm_prob // 3, 50)
        mp[:, random_spans_noise_mask(T_mel, mlm_prob, span)] = True
    else:
        for b in range(B):
            if span_boundary is not None:
                sb = np.asarray(span_boundary[b]).astype(np.int64)
                for s, e in zip(sb[0::2], sb[1::2]):
                    mp[b, s:e] = True
                continue
            L = int(align_lens[b])
            if L < 2:
                continue
            for ph in np.flatnonzero(random_spans_noise_mask(L, mlm_prob, mean_phn_span)):
                mp[b, int(align_start[b][ph]):int(align_end[b][ph])] = True
    return mp & speech_nonpad.astype(bool)


def get_segment_pos(T_mel: int, T_phn: int, align_start, align_end, align_lens, sega_emb: bool = True):
    """collate_fn.py:330-343 (later phones overwrite earlier ones where spans overlap)."""
    B = len(align_lens)
    sp = np.zeros((B, T_mel), dtype=np.int64)
    tp = np.zeros((B, T_phn), dtype=np.int64)
    if not sega_emb:
        return sp, tp
    for b in range(B):
        L = int(align_lens[b])
        for j in range(L):
            sp[b, int(align_start[b][j]):int(align_end[b][j])] = j + 1
        tp[b, :L] = np.arange(1, L + 1)
    return sp, tp


def pad_list(arrs: List[np.ndarray], pad_value) -> np.ndarray:
    """nets_utils.py:34-61."""
    m = max(a.shape[0] for a in arrs)
    out = np.full((len(arrs), m) + arrs[0].shape[1:], pad_value, dtype=arrs[0].dtype)
    for i, a in enumerate(arrs):
        out[i, : a.shape[0]] = a
    return out


class MLMCollateFn:
    """Same constructor / call contract as the reference functor (collate_fn.py:106-157):
    __call__(List[(uid, {speech, text, align_start, align_end[, span_boundary]})]) ->
    (uids, dict(speech, text, masked_position, speech_mask, text_mask, speech_segment_pos,
    text_segment_pos, speech_lengths, text_lengths))."""

    def __init__(self, feats_extract, float_pad_value=0.0, int_pad_value=0, not_sequence=(), mlm_prob=0.8,
                 mean_phn_span=8, attention_window=0, pad_speech=False, sega_emb=False, duration_collect=False):
        if attention_window or duration_collect:
            raise NotImplementedError("longformer window / duration collect are outside the A3T recipe path")
        self.feats_extract = feats_extract
        self.float_pad_value = float_pad_value
        self.int_pad_value = int_pad_value
        self.not_sequence = set(not_sequence)
        self.mlm_prob = mlm_prob
        self.mean_phn_span = mean_phn_span
        self.sega_emb = sega_emb

    def __call__(self, data):
        uids = [u for u, _ in data]
        ds = [d for _, d in data]
        assert all(set(ds[0]) == set(d) for d in ds), "dict-keys mismatching"
        for k in ("text", "align_start", "align_end"):
            if k not in ds[0]:
                raise NotImplementedError("speech-only batches are outside the A3T recipe path")
        speech = pad_list([d["speech"] for d in ds], self.float_pad_value)
        slen = np.array([d["speech"].shape[0] for d in ds], dtype=np.int64)
        text = pad_list([d["text"] for d in ds], self.int_pad_value)
        tlen = np.array([d["text"].shape[0] for d in ds], dtype=np.int64)
        a_s = pad_list([d["align_start"] for d in ds], self.float_pad_value)
        a_e = pad_list([d["align_end"] for d in ds], self.float_pad_value)
        alen = np.array([d["align_start"].shape[0] for d in ds], dtype=np.int64)
        feats, flen = self.feats_extract(torch.from_numpy(speech), torch.from_numpy(slen))
        feats = feats.cpu()
        flen_np = flen.cpu().numpy()
        fs_ = align_to_frames(a_s, self.feats_extract.fs, self.feats_extract.hop_length)
        fe_ = align_to_frames(a_e, self.feats_extract.fs, self.feats_extract.hop_length)
        max_slen = int(flen_np.max())
        speech_pad = feats[:, :max_slen]
        T_phn = text.shape[1]
        text_mask = np.arange(T_phn)[None, :] < tlen[:, None]
        speech_mask = np.arange(max_slen)[None, :] < flen_np[:, None]
        sb = pad_list([d["span_boundary"] for d in ds], 0) if "span_boundary" in ds[0] else None
        masked = phones_masking(max_slen, speech_mask, fs_, fe_, alen, self.mlm_prob, self.mean_phn_span, sb)
        sp, tp = get_segment_pos(max_slen, T_phn, fs_, fe_, alen, self.sega_emb)
        out = dict(speech=speech_pad, text=torch.from_numpy(text), masked_position=torch.from_numpy(masked),
                   speech_mask=torch.from_numpy(speech_mask)[:, None, :],
                   text_mask=torch.from_numpy(text_mask)[:, None, :], speech_segment_pos=torch.from_numpy(sp),
                   text_segment_pos=torch.from_numpy(tp), speech_lengths=torch.from_numpy(slen),
                   text_lengths=torch.from_numpy(tlen))
        return uids, out


def synthetic_batch(c: A3TConfig, B: int, T_mel: int, T_phn: int, seed: int, device="cpu") -> Dict[str, torch.Tensor]:
    """SURVEY §8(d) synthetic training batch: log-mel-like frames, P contiguous phones tiling the
    utterance, no padding, phoneme-span mask from phones_masking under np.random.seed(seed)."""
    rs = np.random.RandomState(seed)
    speech = np.clip(rs.standard_normal((B, T_mel, c.idim)) * 1.5 - 4.0, -10, 2).astype(np.float32)
    text = rs.randint(2, c.vocab - 1, size=(B, T_phn)).astype(np.int64)
    a_s = np.zeros((B, T_phn), dtype=np.int32)
    a_e = np.zeros((B, T_phn), dtype=np.int32)
    for b in range(B):
        cuts = np.sort(rs.choice(np.arange(1, T_mel), size=T_phn - 1, replace=False))
        bounds = np.concatenate([[0], cuts, [T_mel]])
        a_s[b], a_e[b] = bounds[:-1], bounds[1:]
    ones_s = np.ones((B, T_mel), dtype=bool)
    np.random.seed(seed)
    masked = phones_masking(T_mel, ones_s, a_s, a_e, [T_phn] * B, c.mlm_prob, c.mean_phn_span)
    sp, tp = get_segment_pos(T_mel, T_phn, a_s, a_e, [T_phn] * B, True)
    out = dict(speech=torch.from_numpy(speech), text=torch.from_numpy(text),
               masked_position=torch.from_numpy(masked), speech_mask=torch.from_numpy(ones_s)[:, None, :],
               text_mask=torch.ones(B, 1, T_phn, dtype=torch.bool), speech_segment_pos=torch.from_numpy(sp),
               text_segment_pos=torch.from_numpy(tp))
    if c.spk_embed_dim > 0:      # unit-norm speaker vectors
        xv = rs.standard_normal((B, c.spk_embed_dim)).astype(np.float32)
        out["spembs"] = torch.from_numpy(xv / np.linalg.norm(xv, axis=1, keepdims=True))
    return {k: v.to(device) for k, v in out.items()}
