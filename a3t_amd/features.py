"""Log-mel feature extraction on the GPU: mirror of ``LogMelFbank``
(espnet2/tts/feats_extract/log_mel_fbank.py:14-106) = Stft (espnet2/layers/stft.py:56-124:
center=True reflect padding, periodic hann(win_length) centred in n_fft, hop, onesided) ->
amplitude sqrt(clamp(re^2+im^2, 1e-10)) -> LogMel (espnet2/layers/log_mel.py:56-83: Slaney mel
matrix, clamp 1e-10, log10, padded frames -> 0).

MI355X formulation: the STFT is ONE fp32 MFMA GEMM -- frames are overlapping rows of the padded
waveform (A row stride = hop, no framing copy) times the windowed DFT basis [2*(n_fft/2+1)][n_fft];
|.| and log10 are element-wise kernels, the mel projection is a second GEMM.  The mel matrix
restates librosa.filters.mel(htk=False, norm='slaney') (un-vendored dependency of the reference).
"""
import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from ._lib import F32


def slaney_mel_matrix(fs: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """librosa.filters.mel (Slaney scale, area-normalised triangles) -> (n_mels, n_fft//2+1) float32."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def h2m(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def m2h(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    nb = 1 + n_fft // 2
    freqs = np.linspace(0, fs / 2.0, nb)
    edges = m2h(np.linspace(h2m(fmin), h2m(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    lower = -ramps[:-2] / width[:-1, None]
    upper = ramps[2:] / width[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return w.astype(np.float32)


class LogMelFbank:
    """Callable with the reference's interface: (wav (B,N) float32, lengths (B,)) ->
    (feats (B,F,n_mels) float32 on `device`, feats_lengths (B,) int64)."""

    def __init__(self, fs=16000, n_fft=1024, win_length=None, hop_length=256, window="hann", center=True,
                 normalized=False, onesided=True, n_mels=80, fmin=80, fmax=7600, htk=False, device="cuda"):
        if window != "hann" or not center or normalized or not onesided or htk:
            raise NotImplementedError("only the recipe's STFT variant (hann, center, onesided, Slaney) is implemented")
        self.fs = int(fs)
        self.n_fft = n_fft
        self.win_length = win_length or n_fft
        self.hop_length = hop_length
        self.n_mels = n_mels
        self.fmin = 0 if fmin is None else fmin
        self.fmax = fs / 2 if fmax is None else fmax
        self.device = torch.device(device)
        self._dev_tables = None

    def output_size(self) -> int:
        return self.n_mels

    def get_parameters(self):
        return dict(fs=self.fs, n_fft=self.n_fft, hop_length=self.hop_length, window="hann", n_mels=self.n_mels,
                    win_length=self.win_length, center=True, normalized=False, fmin=self.fmin, fmax=self.fmax)

    def _tables(self):
        if self._dev_tables is None:
            n_fft, win = self.n_fft, self.win_length
            nb = n_fft // 2 + 1
            w = np.zeros(n_fft)
            lp = (n_fft - win) // 2
            w[lp:lp + win] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win) / win)     # periodic hann
            k = np.arange(n_fft)[None, :]
            f = np.arange(nb)[:, None]
            ang = 2 * np.pi * ((f * k) % n_fft) / n_fft
            basis = np.concatenate([np.cos(ang) * w[None], -np.sin(ang) * w[None]], 0).astype(np.float32)
            ld = (nb + 3) // 4 * 4
            mel = np.zeros((self.n_mels, ld), np.float32)
            mel[:, :nb] = slaney_mel_matrix(self.fs, n_fft, self.n_mels, self.fmin, self.fmax)
            self._dev_tables = (torch.from_numpy(basis).to(self.device), torch.from_numpy(mel).to(self.device), nb, ld)
        return self._dev_tables

    def __call__(self, wav: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        basis, melT, nb, ld = self._tables()
        dev = self.device
        x = wav.to(dev, torch.float32).contiguous()
        B, N = x.shape
        if lengths is None:
            lengths = torch.full((B,), N, dtype=torch.long)
        pad = self.n_fft // 2
        F = 1 + N // self.hop_length
        row = (N + 2 * pad + 3) // 4 * 4
        xp = torch.empty(B, row, device=dev)
        ops.reflect_pad(x, xp, pad)
        S = torch.empty(B * F, 2 * nb, device=dev)
        # frames are overlapping rows: A(m,k) = xp[b][m*hop + k]
        ops.gemm(xp, basis, S, F, 2 * nb, self.n_fft, self.hop_length, 1, self.n_fft, 1, 2 * nb, batch=B, batch_inner=1,
                 a_bs=(row, 0), c_bs=(F * 2 * nb, 0), compute=F32)
        amp = torch.empty(B * F, ld, device=dev)
        ops.stft_amp(S, amp, nb)
        mel = torch.empty(B * F, self.n_mels, device=dev)
        ops.gemm(amp, melT, mel, B * F, self.n_mels, ld, ld, 1, ld, 1, self.n_mels, compute=F32)
        olens = (lengths.to(torch.long) + 2 * (self.win_length // 2) - self.win_length) // self.hop_length + 1
        ops.logmel_finish(mel, olens.to(dev), B, F, self.n_mels)
        return mel.view(B, F, self.n_mels), olens
