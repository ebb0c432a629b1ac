"""ParallelWaveGAN generator inference on the GPU (mel -> waveform).

Mirror of the vocoder the reference calls in ``sedit_inference.py:77,339-348`` through
``ParallelWaveGANPretrainedVocoder`` (espnet2/tts/utils/parallel_wavegan_pretrained_vocoder.py:49-63);
the network restated is the vendored twin ``ParallelWaveGANGenerator``
(espnet2/gan_tts/parallel_wavegan/parallel_wavegan.py:136-229, wavenet/residual_block.py:114-169,
parallel_wavegan/upsample.py:22-189), weight-norm removed.  Layout is channels-last [T][C] so every
Conv1d (dilated k=3, 1x1) is the shared implicit-im2col MFMA GEMM; the gated activation, residual /
skip update and nearest-neighbour upsampling+smoothing are element-wise HIP kernels.
"""
import math
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import ops
from ._lib import ACT_RELU, F32


class ParallelWaveGANGeneratorHIP:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", layers=30, stacks=3, residual_channels=64,
                 gate_channels=128, skip_channels=64, aux_channels=80, aux_context_window=2,
                 upsample_scales: Sequence[int] = (4, 5, 3, 5), stats: Optional[Dict[str, np.ndarray]] = None,
                 fused: Optional[bool] = None):
        self.dev = torch.device(device)
        self.layers, self.stacks = layers, stacks
        self.R, self.G, self.S, self.A = residual_channels, gate_channels, skip_channels, aux_channels
        self.ctx = aux_context_window
        self.scales = tuple(upsample_scales)
        self.upsample_factor = int(np.prod(self.scales))
        self.stats = None
        if stats is not None:     # normalize_before of the pretrained wrapper (c - mean) / scale
            self.stats = (torch.as_tensor(stats["mean"], dtype=torch.float32, device=self.dev),
                          torch.as_tensor(stats["scale"], dtype=torch.float32, device=self.dev))

        def t(k):
            return torch.as_tensor(np.asarray(state_dict[k]), dtype=torch.float32).to(self.dev)

        def conv(k):              # (out, in, taps) -> [out][tap][in]
            return t(k).permute(0, 2, 1).contiguous()

        self.w_first = t("first_conv.weight").reshape(self.R, 1).contiguous()
        self.b_first = t("first_conv.bias")
        self.w_in = conv("upsample_net.conv_in.weight")
        self.w_up = [t(f"upsample_net.upsample.up_layers.{2 * i + 1}.weight").reshape(-1).contiguous()
                     for i in range(len(self.scales))]
        # fused residual-block kernels (pwg_fused.hip) need the v1 channel plan: 64 residual / 128 gate / 64 skip / 80 aux
        if fused is None:
            fused = os.environ.get("A3T_PWG_FUSED", "1") != "0"
        self.fused = bool(fused) and (self.R, self.G, self.S, self.A) == (64, 128, 64, 80)
        # stage-0 output column n' holds gate channel c = 32*(n'//64) + n'%32, tanh half for (n'//32)%2 == 0 else sigmoid
        npr = np.arange(128)
        perm = torch.as_tensor((npr // 64) * 32 + npr % 32 + 64 * ((npr // 32) % 2), device=self.dev)
        self.blocks = []
        for l in range(layers):
            p = f"conv_layers.{l}."
            blk = dict(w=conv(p + "conv.weight"), b=t(p + "conv.bias"),
                       aux=t(p + "conv1x1_aux.weight").reshape(self.G, self.A).contiguous(),
                       out=t(p + "conv1x1_out.weight").reshape(self.R + self.S, self.G // 2).contiguous(),
                       bout=t(p + "conv1x1_out.bias"))
            if self.fused:
                wk = blk["w"].reshape(self.G, 3 * self.R)                       # [out][tap*64 + in]
                w0 = torch.cat([wk, blk["aux"]], dim=1)[perm]                   # [n'][272]
                blk["wt0"] = w0.t().contiguous()                                # [272][128] k-major
                blk["b0"] = blk["b"][perm].contiguous()
                blk["wt1"] = blk["out"].t().contiguous()                        # [64][128]
            self.blocks.append(blk)
        self.w_l1 = t("last_conv_layers.1.weight").reshape(self.S, self.S).contiguous()
        self.b_l1 = t("last_conv_layers.1.bias")
        self.w_l3 = t("last_conv_layers.3.weight").reshape(1, self.S).contiguous()
        self.b_l3 = t("last_conv_layers.3.bias")

    @torch.no_grad()
    def inference(self, c: torch.Tensor, z: Optional[torch.Tensor] = None, normalize_before: bool = False):
        """c (T_feats, aux) [or (B, T_feats, aux)], z (T_wav, 1) noise -> (T_wav, 1) [or (B, T_wav, 1)]."""
        single = (c.dim() == 2)
        c = c.to(self.dev, torch.float32)
        if single:
            c = c[None]
        B, Tf, A = c.shape
        if normalize_before and self.stats is not None:
            c = (c - self.stats[0]) / self.stats[1]
        Tw = Tf * self.upsample_factor
        if z is None:
            z = torch.randn(B, Tw, 1, device=self.dev)
        z = z.to(self.dev, torch.float32).reshape(B * Tw, 1).contiguous()
        dev = self.dev
        # ---- ConvInUpsampleNetwork: replication pad, conv_in (k = 2*ctx+1, no bias), stretch+smooth per scale
        w = self.ctx
        Tp = Tf + 2 * w
        cp = torch.empty(B * Tp, A, device=dev)
        ops.replicate_pad(c.contiguous(), cp, w)
        ci = torch.empty(B * Tp, A, device=dev)
        ops.conv_fwd(cp, self.w_in, ci, Tp, w, compute=F32)
        cu = ci.view(B, Tp, A)[:, w:w + Tf].contiguous()
        T = Tf
        for sc, wk in zip(self.scales, self.w_up):
            out = torch.empty(B, T * sc, A, device=dev)
            ops.pwg_upsample(cu, wk, out, sc)
            cu, T = out, T * sc
        cu = cu.view(B * Tw, A)
        # ---- first conv (1 -> R), residual stack
        x = torch.empty(B * Tw, self.R, device=dev)
        ops.linear_fwd(z, self.w_first, x, bias=self.b_first, compute=F32)
        skips = torch.zeros(B * Tw, self.S, device=dev)
        y = torch.empty(B * Tw, self.G, device=dev)
        ca = torch.empty(B * Tw, self.G, device=dev)
        g = torch.empty(B * Tw, self.G // 2, device=dev)
        o = torch.empty(B * Tw, self.R + self.S, device=dev)
        lps = self.layers // self.stacks
        for l, blk in enumerate(self.blocks):
            dil = 2 ** (l % lps)
            if self.fused:
                ops.pwg_block(x, cu, blk["wt0"], blk["b0"], blk["wt1"], blk["bout"], g, skips, B, Tw, dil)
                continue
            ops.conv_fwd(x, blk["w"], y, Tw, 1, dil, bias=blk["b"], compute=F32)
            ops.linear_fwd(cu, blk["aux"], ca, compute=F32)
            ops.pwg_gate(y, ca, g)
            ops.linear_fwd(g, blk["out"], o, bias=blk["bout"], compute=F32)
            ops.pwg_res_skip(o, x, skips)
        ops.bias_act(skips, None, ACT_RELU, math.sqrt(1.0 / self.layers))
        h = torch.empty(B * Tw, self.S, device=dev)
        ops.linear_fwd(skips, self.w_l1, h, bias=self.b_l1, act=ACT_RELU, compute=F32)
        wav = torch.empty(B * Tw, 1, device=dev)
        ops.linear_fwd(h, self.w_l3, wav, bias=self.b_l3, compute=F32)
        wav = wav.view(B, Tw, 1)
        return wav[0] if single else wav

    __call__ = inference
