"""Where the on-device collate spends its time (bench.py's collate leg)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from a3t_amd.collate import MLMCollateFn
from a3t_amd.features import LogMelFbank
B, Tm, Tp, hop, fs = 32, 1000, 120, 300, 24000
rs = np.random.RandomState(5)
data = []
for i in range(B):
    n = hop * (Tm - 1)
    cuts = np.sort(rs.choice(np.arange(1, Tm - 1), Tp - 1, replace=False))
    st = np.concatenate([[0], cuts]).astype(np.float32) * hop / fs + 1e-4
    en = np.concatenate([cuts, [Tm - 1]]).astype(np.float32) * hop / fs + 1e-4
    data.append((f"u{i}", dict(speech=(0.1 * rs.standard_normal(n)).astype(np.float32), text=rs.randint(2, 70, size=Tp).astype(np.int64),
                               align_start=st.astype(np.float32), align_end=en.astype(np.float32))))
fe = LogMelFbank(fs=fs, n_fft=2048, win_length=1200, hop_length=hop, n_mels=80, fmin=80, fmax=7600, device="cuda")
for nch in (1,):
    coll = MLMCollateFn(fe, mlm_prob=0.8, mean_phn_span=8, sega_emb=True, device_out=True)
    np.random.seed(1)
    coll(data); coll(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        coll(data)
    th = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    print(f"chunks {nch:2d}: {t * 1e3:.2f} ms per batch ({B * Tm / t / 1e6:.2f} M frames/s), host returns after {th * 1e3:.2f} ms")
pin = torch.empty(B, hop * (Tm - 1), dtype=torch.float32).pin_memory().numpy()
t0 = time.perf_counter()
for _ in range(5):
    for i in range(B):
        pin[i, :] = data[i][1]["speech"]
print(f"host memcpy into the pinned staging buffer alone: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
