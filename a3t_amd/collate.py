"""Host side of batch construction: the mirror of MLMCollateFn
(espnet2/train/collate_fn.py:106-287) with the same call signature and output dictionary.

The index work (alignment seconds -> frame indices, T5-style phoneme-span masking driven by the
numpy global RNG, segment ids) is inherently host-side in the reference too (DataLoader worker);
it is restated here in vectorised numpy.  Feature extraction (STFT -> mel -> log10) runs on the
GPU through liba3t_hip (a3t_amd/features.py) when a device is given.
"""
import os
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


def masking_plan(T_mel: int, align_lens: Sequence[int], mlm_prob: float, mean_phn_span: int, P: int,
                 span_boundary: Optional[np.ndarray] = None):
    """The host half of phones_masking for the on-device collate: consumes the numpy global RNG exactly like phones_masking
    (same calls in the same order) and returns what it decided as integers -- sel [B][P] uint8 (phone j of utterance b is
    masked) and explicit frame spans mspan [B][S][2] int32 / nms [B] (span_boundary, mean_phn_span == 0, mlm_prob == 1)."""
    B = len(align_lens)
    sel = np.zeros((B, max(P, 1)), dtype=np.uint8)
    spans = [[] for _ in range(B)]
    if mlm_prob == 1.0:
        for b in range(B):
            spans[b].append((0, T_mel))
    elif mean_phn_span == 0:
        span = min(T_mel * mlm_prob // 3, 50)
        noise = random_spans_noise_mask(T_mel, mlm_prob, span)
        edges = np.flatnonzero(np.diff(np.concatenate([[0], noise.astype(np.int8), [0]])))
        runs = list(zip(edges[0::2], edges[1::2]))
        for b in range(B):
            spans[b].extend(runs)
    else:
        for b in range(B):
            if span_boundary is not None:
                sb = np.asarray(span_boundary[b]).astype(np.int64)
                spans[b].extend(zip(sb[0::2], sb[1::2]))
                continue
            L = int(align_lens[b])
            if L < 2:
                continue
            sel[b, :L] = random_spans_noise_mask(L, mlm_prob, mean_phn_span)
    S = max(1, max(len(x) for x in spans))
    mspan = np.zeros((B, S, 2), dtype=np.int32)
    nms = np.zeros(B, dtype=np.int32)
    for b in range(B):
        nms[b] = len(spans[b])
        for q, (s0, e0) in enumerate(spans[b]):
            # python slice semantics of mp[b, s:e]: negative indices count from the end, out-of-range ones clip
            s0, e0, _ = slice(int(s0), int(e0)).indices(T_mel)
            mspan[b, q] = (s0, e0)
    return sel, mspan, nms


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
                 mean_phn_span=8, attention_window=0, pad_speech=False, sega_emb=False, duration_collect=False,
                 device_out=False):
        if attention_window or duration_collect:
            raise NotImplementedError("longformer window / duration collect are outside the A3T recipe path")
        self.feats_extract = feats_extract
        self.float_pad_value = float_pad_value
        self.int_pad_value = int_pad_value
        self.not_sequence = set(not_sequence)
        self.mlm_prob = mlm_prob
        self.mean_phn_span = mean_phn_span
        self.sega_emb = sega_emb
        # device_out (extension): with a GPU feature extractor the whole batch dict is built and RETURNED
        # on the device -- features never visit the host, masks / segment ids are painted by a3t_collate_paint from the
        # integer span lists; only the numpy-RNG draws run on the host.  Same values, bit for bit, as the host path.
        self.device_out = bool(device_out)

    def __call__(self, data):
        uids = [u for u, _ in data]
        ds = [d for _, d in data]
        assert all(set(ds[0]) == set(d) for d in ds), "dict-keys mismatching"
        for k in ("text", "align_start", "align_end"):
            if k not in ds[0]:
                raise NotImplementedError("speech-only batches are outside the A3T recipe path")
        slen = np.array([d["speech"].shape[0] for d in ds], dtype=np.int64)
        if self.device_out and getattr(self.feats_extract, "device", None) is not None and \
                torch.device(self.feats_extract.device).type == "cuda" and ds[0]["speech"].ndim == 1:
            return uids, self._device_pipeline(ds, slen)
        speech = pad_list([d["speech"] for d in ds], self.float_pad_value)
        text = pad_list([d["text"] for d in ds], self.int_pad_value)
        tlen = np.array([d["text"].shape[0] for d in ds], dtype=np.int64)
        a_s = pad_list([d["align_start"] for d in ds], self.float_pad_value)
        a_e = pad_list([d["align_end"] for d in ds], self.float_pad_value)
        alen = np.array([d["align_start"].shape[0] for d in ds], dtype=np.int64)
        feats, flen = self.feats_extract(torch.from_numpy(speech), torch.from_numpy(slen))
        if self.device_out and feats.is_cuda:
            return uids, self._finish_on_device(ds, feats, flen, text, tlen, a_s, a_e, alen, slen)
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


def _collate_finish_on_device(self, ds, feats, flen, text, tlen, a_s, a_e, alen, slen):
    from . import ops
    dev = feats.device
    # (the feature extractor derives the frame counts from the host-side sample counts it is handed, so `flen` is a host
    #  tensor here and nothing waits for the device; a caller that passes device lengths pays one sync for them)
    flen_np = flen.cpu().numpy() if flen.is_cuda else flen.numpy()
    fe_x = self.feats_extract
    fs_ = align_to_frames(a_s, fe_x.fs, fe_x.hop_length)
    fe_ = align_to_frames(a_e, fe_x.fs, fe_x.hop_length)
    max_slen = int(flen_np.max())
    B, T_phn, P = text.shape[0], text.shape[1], fs_.shape[1]
    sb = pad_list([d["span_boundary"] for d in ds], 0) if "span_boundary" in ds[0] else None
    sel, mspan, nms = masking_plan(max_slen, alen, self.mlm_prob, self.mean_phn_span, P, sb)
    # one small H2D copy for all the integer side inputs
    ints = np.concatenate([fs_.reshape(-1), fe_.reshape(-1), alen.astype(np.int32), mspan.reshape(-1), nms,
                           flen_np.astype(np.int32), tlen.astype(np.int32)]).astype(np.int32)
    di = torch.from_numpy(ints).to(dev, non_blocking=True)
    o = 0
    def take(n, shape):
        nonlocal o
        v = di[o:o + n].view(shape)
        o += n
        return v
    d_fs, d_fe = take(B * P, (B, P)), take(B * P, (B, P))
    d_alen = take(B, (B,))
    d_ms = take(mspan.size, mspan.shape)
    d_nms, d_flen, d_tlen = take(B, (B,)), take(B, (B,)), take(B, (B,))
    d_sel = torch.from_numpy(sel).to(dev, non_blocking=True)
    masked = torch.empty(B, max_slen, dtype=torch.uint8, device=dev)
    smask = torch.empty(B, max_slen, dtype=torch.uint8, device=dev)
    tmask = torch.empty(B, T_phn, dtype=torch.uint8, device=dev)
    sp = torch.empty(B, max_slen, dtype=torch.int64, device=dev)
    tp = torch.empty(B, T_phn, dtype=torch.int64, device=dev)
    ops.collate_paint(d_fs, d_fe, d_alen, d_sel, d_ms, d_nms, d_flen, d_tlen, masked, smask, tmask, sp, tp, self.sega_emb)
    return dict(speech=feats[:, :max_slen], text=torch.from_numpy(text).to(dev, non_blocking=True),
                masked_position=masked.view(torch.bool), speech_mask=smask.view(torch.bool)[:, None, :],
                text_mask=tmask.view(torch.bool)[:, None, :], speech_segment_pos=sp, text_segment_pos=tp,
                speech_lengths=torch.from_numpy(slen).to(dev, non_blocking=True),
                text_lengths=torch.from_numpy(tlen).to(dev, non_blocking=True))


def _collate_device_pipeline(self, ds, slen):
    """device_out with raw waveforms: the utterances are padded straight into a pinned staging buffer, go to the device in ONE
    asynchronous copy and through STFT -> mel -> log10 there.  (Chunked copies that overlap the host-side padding with PCIe and
    the GPU were measured in round 4 -- 1 chunk 6.1 ms, 2: 6.6, 4: 6.9, 8: 8.4 per batch: the padding memcpy is 1.3 ms, a chunk's
    launches cost more than they hide.)"""
    fe_x = self.feats_extract
    dev = torch.device(fe_x.device)
    B, N = len(ds), int(slen.max())
    # pinned staging buffer: flat and grow-only (a hipHostMalloc per batch shape would cost more than the copy it serves)
    if getattr(self, "_pin", None) is None or self._pin.numel() < B * N:
        self._pin = torch.empty(max(B * N, int(1.25 * getattr(self, "_pin_cap", 0))), dtype=torch.float32).pin_memory()
        self._pin_cap = self._pin.numel()
    pin = self._pin[:B * N].view(B, N)
    pnp = pin.numpy()
    ev = getattr(self, "_pin_ev", None)
    if ev is not None:
        ev.synchronize()          # the previous call's copy out of the staging buffer is done
    for i in range(B):
        n = int(slen[i])
        pnp[i, :n] = ds[i]["speech"]
        if n < N:
            pnp[i, n:] = self.float_pad_value
    xd = pin.to(dev, non_blocking=True)
    feats, flen = fe_x(xd, torch.from_numpy(slen))
    self._pin_ev = torch.cuda.Event()
    self._pin_ev.record()
    text = pad_list([d["text"] for d in ds], self.int_pad_value)
    tlen = np.array([d["text"].shape[0] for d in ds], dtype=np.int64)
    a_s = pad_list([d["align_start"] for d in ds], self.float_pad_value)
    a_e = pad_list([d["align_end"] for d in ds], self.float_pad_value)
    alen = np.array([d["align_start"].shape[0] for d in ds], dtype=np.int64)
    return self._finish_on_device(ds, feats, flen, text, tlen, a_s, a_e, alen, slen)


MLMCollateFn._finish_on_device = _collate_finish_on_device
MLMCollateFn._device_pipeline = _collate_device_pipeline


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
