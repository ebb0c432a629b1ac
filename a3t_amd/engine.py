"""Forward / backward sequencing of the A3T masked-mel training step on one MI355X.

This is the host-side replacement for the autograd graph the reference builds out of
nn.Modules (ESPnetMLMEncAsDecoderModel._forward, espnet2/tts/sedit/sedit_model.py:350-375,
MLMEncoder/MLMDecoder conformer/encoder.py:522-614, EncoderLayer conformer/encoder_layer.py:80-180):
an explicit, allocation-free schedule of liba3t_hip launches on the current HIP stream, with a
hand-derived backward pass.  torch supplies device buffers only.

Two numeric modes:
  compute="f32"  : everything fp32, GEMMs on the exact-fp32 MFMA  (parity path, 1e-4)
  compute="bf16" : tensors that feed GEMMs are stored in bf16 (LayerNorm outputs, FFN hidden, q/k/v,
                   attention probabilities, ...), weights are cast to a bf16 shadow once per step,
                   GEMMs run on the bf16 MFMA with fp32 accumulation; the residual stream, all
                   normalisation statistics, softmax, loss, gradients of parameters and the
                   optimizer stay fp32  (throughput path, 1e-2)
"""
import math
import os
import zlib
from typing import Dict

import torch

from . import ops
from ._lib import ACC_ADD, ACC_ATOMIC, ACC_STORE, ACT_NONE, ACT_RELU, ACT_SWISH, ACT_TANH, BF16, F32
from .config import A3TConfig
from .params import ParamStore


def legacy_pe_table(c: A3TConfig) -> torch.Tensor:
    """LegacyRelPositionalEncoding table (transformer/embedding.py:59-80 with reverse=True): a
    constant computed once on the host with the same fp32 formula, row t = PE(max_len-1-t)."""
    d = c.adim
    position = torch.arange(c.max_len - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(c.max_len, d)
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe


_STREAMS = {}


def shared_stream(dev, kind, high=False):
    """One HIP stream per (device, role) for the whole process.  HIP maps streams onto a small pool of hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default) round-robin: a process that builds a second engine / trainer with streams of its own
    gets a main and a side stream on the SAME hardware queue sooner or later and the two-stream schedule silently runs as
    one (bench.py's configs[3] leg, the third trainer of its process: 84 ms per step instead of 66)."""
    idx = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    key = (idx, kind)
    if key not in _STREAMS:
        pr = torch.cuda.Stream.priority_range()[1] if high else 0
        _STREAMS[key] = torch.cuda.Stream(device=dev, priority=pr)
    return _STREAMS[key]


class _FastEvent:
    """Cross-stream hand-over event created with hipEventDisableTiming | hipEventDisableSystemFence: the marker packet of
    a default-flag event costs the queue it is recorded on ~7.3 us between two kernels, this one ~5.2 (tools/event_cost.py,
    rocprofv3 kernel trace); the engine records ~110 of them per step on the main stream.  Same interface as the two methods
    of torch.cuda.Event the engine uses (torch's own events are the fallback when libamdhip64.so cannot be loaded by that name)."""
    _hip = None
    _pools = {}        # device -> [events, next]: an event belongs to the device that was current when it was created

    @classmethod
    def create(cls):
        """A fresh event that belongs to its caller (never handed out again)."""
        import ctypes
        if cls._hip is None:
            cls._hip = ctypes.CDLL("libamdhip64.so")
        h = ctypes.c_void_p()
        if cls._hip.hipEventCreateWithFlags(ctypes.byref(h), ctypes.c_uint(0x2 | 0x20000000)) != 0:
            raise RuntimeError("hipEventCreateWithFlags failed")
        e = cls.__new__(cls)
        e.h = h
        return e

    @classmethod
    def get(cls):
        """An event from the per-device ring, for hand-overs that are recorded and waited for back to back (a wait captures
        the record that precedes it, so re-recording a ring event later is safe).  Anything a holder keeps across other
        hand-overs must NOT come from here (MLMEngine._owned_event): 1024 later get() calls would re-record it."""
        pool = cls._pools.setdefault(torch.cuda.current_device(), [[], 0])
        if len(pool[0]) < 1024:
            e = cls.create()
            pool[0].append(e)
            return e
        pool[1] = (pool[1] + 1) % len(pool[0])
        return pool[0][pool[1]]

    def record(self, stream=None):
        import ctypes
        st = stream if stream is not None else torch.cuda.current_stream()
        if self._hip.hipEventRecord(self.h, ctypes.c_void_p(st.cuda_stream)) != 0:
            raise RuntimeError("hipEventRecord failed")

    def wait_on(self, stream):
        import ctypes
        if self._hip.hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), self.h, 0) != 0:
            raise RuntimeError("hipStreamWaitEvent failed")


class _TorchEvent:
    def __init__(self):
        self.e = torch.cuda.Event()

    def record(self, stream=None):
        self.e.record(stream) if stream is not None else self.e.record()

    def wait_on(self, stream):
        stream.wait_event(self.e)


def _new_event(owned=False):
    global _FAST_EVENTS
    if _FAST_EVENTS:
        try:
            return _FastEvent.create() if owned else _FastEvent.get()
        except (OSError, RuntimeError):          # no libamdhip64 by that name: torch's events do the same job
            _FAST_EVENTS = False
    return _TorchEvent()


_FAST_EVENTS = True


class Workspace:
    """Named device buffers, allocated on first use and reused every step (static shapes)."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[tuple, torch.Tensor] = {}

    def get(self, name, shape, dtype=torch.float32, zero=False, zero_once=False):
        """zero: cleared on every call; zero_once: cleared when allocated only (buffers with entries nobody writes)."""
        key = (name, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = (torch.zeros if (zero or zero_once) else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
            self.bufs[key] = t
        elif zero:
            t.zero_()
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class MLMEngine:
    def __init__(self, cfg: A3TConfig, store: ParamStore, compute: str = "f32", training: bool = True,
                 dropout: bool = False):
        """dropout=True enables the recipe's Dropout sites (dropout_rate / positional / attention rates of
        the config, 0.5 in the postnet) in training mode; masks come from a counter RNG keyed by
        (step seed, site), regenerated in the backward pass."""
        self.c = cfg
        self.dropping = bool(dropout and training)
        self.step_seed = 0
        self.store = store
        self.dev = store.device
        self.ws = Workspace(self.dev)
        self.bf16 = (compute == "bf16")
        self.adt = torch.bfloat16 if self.bf16 else torch.float32   # storage of GEMM-operand activations
        self.cmp = BF16 if self.bf16 else F32
        self.training = training
        self.pe = legacy_pe_table(cfg).to(self.dev)
        self.sv = {}
        self.scratch64 = torch.zeros(4 * max(cfg.ff, 3 * cfg.adim, cfg.postnet_chans, 64), dtype=torch.float64,
                                     device=self.dev)
        self.bn_momentum = 0.1
        # Weight-gradient GEMMs feed nothing until the optimizer, so the backward schedule launches them on a second HIP
        # stream: they fill the partially empty last round of the data-gradient GEMMs' grids and overlap the HBM-bound
        # row kernels (LayerNorm / softmax backward).  Scratch tensors they read are double-buffered by sub-layer parity.
        self.side = self.side2 = None
        if self.dev.type == "cuda" and os.environ.get("A3T_SIDE_STREAM", "1") != "0":
            self.side = shared_stream(self.dev, "side")
            # results the main stream waits for inside the sub-layer (dV, dK of the attention backward) get a stream of their own:
            # queued behind the weight gradients they would hand the main stream the whole backlog to wait for
            self.side2 = (shared_stream(self.dev, "side2", high=True)
                          if os.environ.get("A3T_SIDE2", "1") != "0" else self.side)
        self._owned = {}       # events a holder keeps across other hand-overs: one per role, owned by this engine
        self._par = 0
        self._depth = max(2, int(os.environ.get("A3T_SIDE_DEPTH", "48")))   # scratch sets the main stream may run ahead by
        self._side_ev = [None] * self._depth
        # Linear weight gradients collected for one grouped launch (_lin_wgrad): 4 = one Conformer block's; needs the deep scratch ring
        self._wg_group = 4 if (self.bf16 and self.side is not None and self._depth >= 8 and
                               os.environ.get("A3T_WGRAD_GROUP", "1") != "0") else 1
        self._wg_pending = []
        self._wg_slots = set()
        self._gm_ready = None
        self._arena = {k: dict(buf=None, used=0, slots={}, dtype=dt) for k, dt in
                       (("fwd64", torch.float64), ("bwd64", torch.float64), ("bwd32", torch.float32))}
        self.colsum_slots = 16          # spread of the attention bias-gradient atomics (1 and 64 measured: slower / equal)
        self.fuse_ln_dropout = True      # LayerNorm backward emits the next sub-layer's masked gradient (bf16, d % 128 == 0)
        # bf16 mode: the first postnet conv reads `before` (log-mel scale, |x| ~ 4: one bf16 ulp = 0.03) whose rounding the
        # five BatchNorm'ed postnet layers amplify.  Its forward therefore runs on (hi, lo) = (bf16(x), bf16(x - hi)): two
        # K = 5*80 bf16 GEMMs carry `before` to ~2^-17 (max error of `after` 6.2e-2 -> 4.3e-2 of scale); gradients use hi only.
        self.post_f32_first = os.environ.get("A3T_POST_F32_FIRST", "1") != "0"
        self.sfc_f32 = True
        # the attention-dropout mask of the score gradients comes back from the counter RNG instead of being read off the dropped
        # probabilities (one T x T read less)
        self.attn_regen = True
        # dS / dBD straight from the saved probabilities in one launch (a3t_attn_bwd_ds) instead of the dprobs GEMM + softmax
        # backward: dprobs is never stored.  Default since round 5 at d_k >= 160, since round 6 (the kernel forms delta itself:
        # one launch and one [B][H][T] tensor less) at every supported d_k -- configs[3] (d_k = 128): 61.25 ms per step against
        # 62.33 / 62.30 with the materialised pair (profiles/r06_c4_ab.txt).  A3T_ATTN_BWD_DS=0: the materialised pair.
        self.attn_bwd_ds = os.environ.get("A3T_ATTN_BWD_DS", "1") != "0"
        # With attention dropout the fused training forward saves ONE score-sized tensor: exp(s - m_ref) with the dropout mask in its
        # sign bits (a3t_attn_fwd_train without probs_drop); a3t_attn_bwd_ds reads the mask there and the dV product clamps the
        # tagged elements to zero in its fragment registers (a3t_gemm_desc::a_signmask).  The dropped copy -- 161 MB written by
        # the forward per layer at configs[1], 415 MB at configs[3] -- is gone.  A3T_ATTN_SIGNED=0: two tensors (the A/B twin).
        self.attn_signed = os.environ.get("A3T_ATTN_SIGNED", "1") != "0"
        self.attn_dq_dual = os.environ.get("A3T_ATTN_DQ_DUAL", "1") != "0"
        self.attn_dbd_view = os.environ.get("A3T_ATTN_DBD_VIEW", "1") != "0"
        self.attn_dk_main = os.environ.get("A3T_ATTN_DK_MAIN", "1") != "0"
        # Fused legacy rel-pos attention forward (csrc/attn_fused.hip: scores, shifted position term, softmax, dropout and PV in one
        # launch, no logits in HBM).  A3T_FUSED_ATTN = auto (default) | fwd | 0:
        #   forward-only passes (need_grad=False) use a3t_attn_fwd when it launches >= 64 workgroups (fwd: always; below that its
        #   latency floor loses to the materialised forward);
        #   training steps under "auto" use a3t_attn_fwd_train, which also stores the un-normalised probabilities exp(s - m_ref), their
        #   dropped copy and 1 / row sum for the backward (A3T_FUSED_ATTN_TRAIN=0 switches that off, =2 lifts the 64-workgroup
        #   threshold so that the small-batch parity fixtures go through it).
        # The backward reads the saved probabilities (a3t_attn_bwd_ds or the materialised pair above).
        fa = os.environ.get("A3T_FUSED_ATTN", "auto")
        if fa not in ("auto", "fwd", "0"):
            raise ValueError(f"A3T_FUSED_ATTN={fa!r}: expected auto, fwd or 0 (the flash backward behind '1' left the library in "
                             "round 5, tools/experiments/attn_flash_backward.patch)")
        self.fused_attn_fwd_only = self.bf16 and fa == "fwd"
        self.fused_attn_auto = self.bf16 and fa == "auto"
        self.fused_attn_train = self.bf16 and fa == "auto" and os.environ.get("A3T_FUSED_ATTN_TRAIN", "1") != "0"
        self.fused_attn_train_min = 1 if os.environ.get("A3T_FUSED_ATTN_TRAIN", "1") == "2" else 64
        self._fused_now = False
        self._fused_train_now = False
        self._need_grad = training
        self._mode_tag = None
        if self.bf16:
            for n, v in (("adim", cfg.adim), ("ff", cfg.ff), ("idim", cfg.idim), ("odim", cfg.odim),
                         ("dk", cfg.dk), ("postnet_chans", cfg.postnet_chans or 8)):
                if v % 8:
                    raise ValueError(f"compute='bf16' needs {n} % 8 == 0 (16-byte DMA granules), got {v}")
            self.flat16 = torch.zeros(store.total, dtype=torch.bfloat16, device=self.dev)
            self.p16 = {k: self.flat16[o:o + math.prod(s)].view(s) for k, (o, s) in store.offsets.items()}
        self._ffn_plans = {}
        self._lin_plans = {}
        self._wt = {}        # transposed FFN weight shadows (8-phase data gradients), built on demand
        # (settled by the A/Bs of rounds 4-5; the switches left in round 6) weights cast late on the side stream, the
        # positional projections made ahead on it, the head's weight gradients and the attention's small kernels on the side streams
        self._late_cast = self._pos_ahead = True
        self._P_ahead, self._pos_ev = {}, None
        self._cast_ev = self._wt_ev = None
        dec = [o for k, (o, _) in store.offsets.items() if k.startswith("dec.")]
        self._cast_split = min(dec) if dec else 0        # flat offset of the first decoder parameter (a multiple of 64 elements)

    # ------------------------------------------------------------------ helpers
    def _drop(self, p, tag):
        """(p, key) of one dropout site for the current step, or None when dropout is off."""
        if not self.dropping or p <= 0.0:
            return None
        h = (zlib.crc32(tag.encode()) ^ ((self.step_seed * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF
        h = ((h ^ (h >> 16)) * 0x85EBCA6B) & 0xFFFFFFFF
        h = ((h ^ (h >> 13)) * 0xC2B2AE35) & 0xFFFFFFFF
        return (float(p), (h ^ (h >> 16)) & 0xFFFFFFFF)

    def _gm(self, g, tag, p, bias_grad, bias_scale):
        """Gradient entering a "residual + a*dropout(branch)" sub-layer: the GEMM operand (bf16 on the
        bf16 path) and, with dropout on, the masked gradient + the branch's output-bias gradient."""
        dr = self._drop(p, tag)
        if dr is None:
            return (self._g16(g) if self.bf16 else g)
        gm = self.ws.get(self._t("tmp.gm"), tuple(g.shape), self.adt)
        if self._gm_ready == tag:       # already produced by the LayerNorm backward that wrote g (fused epilogue)
            self._gm_ready = None
            return gm
        ops.dropout_bwd_cast(g, gm, dr[0], dr[1], colsum=bias_grad, colsum_scale=bias_scale)
        return gm

    def W(self, name):
        """GEMM-operand view of a parameter (bf16 shadow in bf16 mode)."""
        return self.p16[name] if self.bf16 else self.store.p[name]

    def refresh_weights(self):
        """bf16 shadows of the fp32 master weights, once per forward.  The flat buffer is laid out in forward order (prologue,
        encoder, decoder, head): only the part in front of the decoder is cast on the main stream; the rest, and the transposed
        shadows that only the backward reads, are cast on the (idle) side stream under the encoder's GEMMs -- forward() waits
        for the first before the decoder, backward() for the second (A3T_LATE_CAST=0: everything up front, 0.33 ms of HBM-bound
        kernels with nothing beside them)."""
        self._cast_ev = self._wt_ev = None
        if not self.bf16:
            return
        split = self._cast_split if (self.side is not None and self._late_cast) else 0
        flat, flat16 = self.store.flat, self.flat16
        wt = list(self._wt.values()) if self._need_grad else []      # transposed shadows feed data gradients only

        def transposed():
            for src_off, dst_off, _, tflat, shp in wt:
                ops.cast_bf16_conv_t(flat, tflat, src_off, dst_off, *shp)
        if split <= 0:
            ops.cast_bf16(flat, flat16)
            transposed()
            return
        ops.cast_bf16(flat[:split], flat16[:split])
        self._cast_ev = self._side(lambda: ops.cast_bf16(flat[split:], flat16[split:]), want_event="cast")
        if wt:
            self._wt_ev = self._side(transposed, want_event="wt")

    def _wait_cast(self, which):
        ev = getattr(self, which, None)
        if ev is not None:
            ev.wait_on(torch.cuda.current_stream())
            setattr(self, which, None)

    def _setup_wt(self, suf, shape):
        """Transposed, tap-reversed bf16 shadows of every weight `*.suf` of `shape` = [n][k][c] -> [c][k'][n] (Linear: k = 1, stored
        [n][c]): the data gradient of that conv / linear becomes a k-contiguous conv of the output gradient
        (a3t_cast_bf16_conv_t), which the 8-phase and the 384-column panel GEMMs can run.  One transposing cast per step for all
        of them; only built when one of those kernels would take the GEMM."""
        stored = (tuple(shape), (shape[0], shape[2])) if shape[1] == 1 else (tuple(shape),)
        names = [k for k in self.store.offsets if k.endswith("." + suf) and tuple(self.store.offsets[k][1]) in stored]
        if not names:
            return
        n = math.prod(shape)
        flat = torch.zeros(n * len(names), dtype=torch.bfloat16, device=self.dev)
        src = torch.tensor([self.store.offsets[k][0] for k in names], dtype=torch.int64, device=self.dev)
        dst = torch.arange(len(names), dtype=torch.int64, device=self.dev) * n
        views = {k: flat[i * n:(i + 1) * n].view(shape[2], shape[1], shape[0]) for i, k in enumerate(names)}
        self._wt[suf] = (src, dst, views, flat, tuple(shape))
        ops.cast_bf16_conv_t(self.store.flat, flat, src, dst, *shape)

    def _ffn_plan(self, M):
        """Which FFN GEMMs of an M-token batch go to the persistent 8-phase / panel kernels (the library's cost models decide,
        a3t_gemm_8p_supported / a3t_gemm_pn_supported): (keep, dgrad1, dgrad2) = the keep-bit protocol (forward conv 1 writes one bit per hidden activation,
        the data gradient of conv 2 reads it through the transposed w_2), and the data gradient of conv 1 through the
        transposed w_1.  A3T_FFN_8P=0 turns both off."""
        # keyed by what the probes depend on: the kernels' modes may change at run time (a3t_gemm_8p_mode / _pn_mode), and a
        # forward-only pass needs neither keep bits nor transposed weight shadows
        key = (M, self._need_grad, self._mode_tag)
        if key not in self._ffn_plans:
            c = self.c
            keep = d1 = d2 = keep4 = False
            if self._need_grad and self.bf16 and self.dev.type == "cuda" and os.environ.get("A3T_FFN_8P", "1") != "0":
                k = c.ff_kernel
                drop = ops.G8_DROP if (self.dropping and c.dropout_rate > 0) else 0
                keep = ops.gemm_8p_supported(M, c.ff, k * c.adim, k, ops.G8_BIAS_ACT | drop | ops.G8_KEEP_OUT) and \
                    ops.gemm_8p_supported(M, c.ff, k * c.adim, k, ops.G8_KEEP_IN | ops.G8_COLSUM)
                d1 = ops.gemm_8p_supported(M, c.adim, k * c.ff, k, 0) or ops.gemm_pn_supported(M, c.adim, k * c.ff, k, 0)
                # (no keep bits: the panel GEMM runs the data gradient of conv 2 on the transposed w_2 with the saved hidden
                #  activation as its ReLU' mask, 
                d2 = (not keep) and \
                    ops.gemm_pn_supported(M, c.ff, k * c.adim, k, ops.G8_SMASK | ops.G8_COLSUM)
                # ... or, round 6, with a row-major keep image the forward conv writes from the 128-row kernel's epilogue (4 bits
                # per byte: 14 MB read in the place of the 110-MB activation at configs[1]; A3T_FFN_KEEP4=0: the activation)
                keep4 = (not keep) and os.environ.get("A3T_FFN_KEEP4", "1") != "0" and c.ff % 8 == 0 and \
                    ops.gemm_pn_supported(M, c.ff, k * c.adim, k, ops.G8_KEEP_IN | ops.G8_COLSUM)
                d2 = d2 or keep4
                if (keep or d2) and "w2" not in self._wt:
                    self._setup_wt("w2", (c.adim, k, c.ff))
                if d1 and "w1" not in self._wt:
                    self._setup_wt("w1", (c.ff, k, c.adim))
                keep, d1, d2 = keep and "w2" in self._wt, d1 and "w1" in self._wt, d2 and "w2" in self._wt
                keep4 = keep4 and d2
            self._ffn_plans[key] = (keep, d1, d2, keep4)
        return self._ffn_plans[key]

    def _lin_dgrad(self, dy, name, dx):
        """dx = dy W for the Linear weight `name` ([out][in]).  When the 384-column panel GEMM would take the k-contiguous form
        (in = 384 columns, A3T_LIN_DGRAD_T=0 turns it off) it runs as dx = dy (W^T)^T on the transposed bf16 shadow of W."""
        suf = name.rsplit(".", 1)[1]
        M = dy.shape[0]
        key = (suf, M, dx.dtype, self._mode_tag)
        if key not in self._lin_plans:
            use = False
            if self.bf16 and self.dev.type == "cuda" and os.environ.get("A3T_LIN_DGRAD_T", "1") != "0":
                n_out, n_in = self.store.offsets[name][1]
                use = ops.gemm_pn_supported(M, n_in, n_out, 1, ops.G8_F32_OR_RES if dx.dtype == torch.float32 else 0)
                if use and suf not in self._wt:
                    self._setup_wt(suf, (n_out, 1, n_in))
                use = use and suf in self._wt and name in self._wt[suf][2]
            self._lin_plans[key] = use
        if self._lin_plans[key]:
            wt = self._wt[suf][2][name]
            ops.linear_fwd(dy, wt.view(wt.shape[0], wt.shape[2]), dx, compute=self.cmp)
        else:
            ops.linear_bwd_data(dy, self.W(name), dx, compute=self.cmp)

    def _act(self, name, shape):
        return self.ws.get(name, shape, self.adt)

    def _ln_fwd(self, tag, x, pre, eps=1e-12, out_dtype=None):
        p = self.store.p
        M, D = x.shape
        y = self.ws.get(tag + ".y", (M, D), out_dtype or self.adt)
        mean = self.ws.get(tag + ".mean", (M,))
        rstd = self.ws.get(tag + ".rstd", (M,))
        ops.layernorm_fwd(x, p[pre + ".g"], p[pre + ".b"], y, mean, rstd, eps)
        self.sv[tag] = (x, y, mean, rstd)
        return y

    def _ln_bwd(self, tag, dy, pre, dres, dx, dx16=None, nb=None, nxt=None):
        """nb = (bias-gradient view, scale) of the layer that produced the residual stream this LN read:
        its bias gradient is the column sum of dx; fused into the kernel on the bf16 path.
        nxt = dropout-site tag of that layer's output dropout (the next sub-layer of the backward schedule): with
        dropout on, the kernel also emits that sub-layer's masked bf16 gradient operand and masked bias gradient,
        which replaces the separate dropout_bwd_cast pass over g."""
        p, g = self.store.p, self.store.g
        x, _, mean, rstd = self.sv[tag]
        dropping = self.dropping and self.c.dropout_rate > 0
        if dropping and self.bf16 and self.fuse_ln_dropout and nb is not None and nxt is not None and x.shape[1] % 128 == 0:
            dr = self._drop(self.c.dropout_rate, nxt)
            nx = (self._par + 1) % self._depth
            gm = self.ws.get(f"tmp.gm.{nx}", tuple(dx.shape), self.adt)
            ev = self._side_ev[nx]                     # side-stream readers of that scratch set must have drained
            if ev is not None:
                ev.wait_on(torch.cuda.current_stream())
                self._side_ev[nx] = None
            ops.layernorm_bwd(dy, x, p[pre + ".g"], mean, rstd, dres, dx, g[pre + ".g"], g[pre + ".b"], dx16=gm,
                              dxsum=nb[0], dxsum_scale=nb[1], drop=dr)
            self._gm_ready = nxt
            return
        if dropping and nxt is not None:
            nb = None          # a layer with output dropout: its bias gradient comes from the masked gradient (_gm)
        # (nxt is None: the layer has no output dropout -- the speech-embedding Linear -- and its bias gradient is the plain
        #  column sum of dx also when dropout is on; round 3: until then emb.b got no gradient in dropout-on steps)
        fuse = self.bf16 and nb is not None
        ops.layernorm_bwd(dy, x, p[pre + ".g"], mean, rstd, dres, dx, g[pre + ".g"], g[pre + ".b"], dx16=dx16,
                          dxsum=nb[0] if fuse else None, dxsum_scale=nb[1] if fuse else 1.0)
        if nb is not None and not fuse:
            self._bias_grad(dx, nb[0], nb[1])

    def _bias_grad(self, dy, gb, scale=1.0):
        ops.bias_grad(dy, gb, self.scratch64, scale)

    def _g16(self, g):
        """bf16 companion of the residual-stream gradient (written by the LayerNorm backward)."""
        return self.ws.get("grad.x16", tuple(g.shape), torch.bfloat16) if self.bf16 else None

    # ---- small accumulators (BatchNorm sums, rel-pos projection gradient) -------------------------
    # They must start every pass at zero.  Instead of ~45 tiny fill launches per step they live in two arenas
    # (fp64 / fp32) that are cleared with ONE fill each at the start of forward and of backward; a slot keeps its
    # offset for the lifetime of the engine.
    def _arena_slot(self, kind, name, numel):
        ar = self._arena[kind]
        if name not in ar["slots"]:
            ar["slots"][name] = (ar["used"], numel)
            ar["used"] += (numel + 63) // 64 * 64
            if ar["buf"] is None or ar["used"] > ar["buf"].numel():   # grow (first step only): re-create and clear
                new = torch.zeros(max(2 * ar["used"], 1 << 16), dtype=ar["dtype"], device=self.dev)
                ar["buf"] = new
        o, n = ar["slots"][name]
        return ar["buf"][o:o + n]

    def _arena_clear(self, kind):
        ar = self._arena[kind]
        if ar["buf"] is not None and ar["used"]:
            ar["buf"][:ar["used"]].zero_()

    # ---- side-stream protocol -------------------------------------------------------------------
    def _t(self, name):
        """Scratch tensor name for the current sub-layer parity (tensors a side-stream GEMM reads)."""
        return f"{name}.{self._par}"

    def _sub_begin(self):
        """Start of a sub-layer backward: flip the parity; its scratch set was last read by the side-stream work
        issued two sub-layers ago, which must have drained before the main stream overwrites it."""
        self._par = (self._par + 1) % self._depth
        ev = self._side_ev[self._par]
        if ev is not None:
            ev.wait_on(torch.cuda.current_stream())
            self._side_ev[self._par] = None

    def _owned_event(self, key):
        """The engine's own event for one long-lived role (scratch-set slot, weight casts, positional projections): held from
        its record to a wait that may come a whole backward later, so it cannot be a ring event (_FastEvent.get)."""
        ev = self._owned.get(key)
        if ev is None:
            ev = self._owned[key] = _new_event(owned=True)
        return ev

    def _sub_end(self):
        if self.side is not None:
            ev = self._owned_event(("slot", self._par))
            ev.record(self.side)
            self._side_ev[self._par] = ev

    def _side(self, fn, want_event=False, urgent=False):
        """Run fn (work whose inputs are complete on the main stream NOW) on the side stream; with want_event the
        returned event marks its completion (for results the main stream consumes later).  urgent: the main stream joins
        this work before the sub-layer ends -- it goes to the second side stream, in front of no backlog."""
        if self.side is None:
            fn()
            return None
        ev = _new_event()
        ev.record()
        st = self.side2 if urgent else self.side
        with torch.cuda.stream(st):
            ev.wait_on(st)
            fn()
            if want_event:     # True: waited for inside the same sub-layer (ring event); a key: kept by the caller (owned event)
                done = _new_event() if want_event is True else self._owned_event(("side", want_event))
                done.record()
                return done
        return None

    def _lin_wgrad(self, dy, x, dW):
        """Weight gradient of a Linear (dW += dy^T x), off the main stream.  bf16 path: the Linear weight gradients of a block
        (linear_out, linear_q/k/v, pointwise_conv2, pointwise_conv1 -- all reductions over the same tokens with 384 input
        channels) are collected and handed over as ONE grouped launch (ops.linear_bwd_weight_group): they feed nothing until the
        optimizer, their operands live in the scratch ring / the saved activations, and four of them fill the chip where each
        alone has 16-20 K-tiles per workgroup.  A3T_WGRAD_GROUP=0: one launch each."""
        if self._wg_group > 1:
            self._wg_pending.append((dy, x, dW, 1.0))
            self._wg_slots.add(self._par)       # the scratch set its operands may live in (tmp.gm / tmp.dg / tmp.dqkv of this sub-layer)
            if len(self._wg_pending) >= self._wg_group:
                self._wg_flush()
            return
        self._side(lambda: ops.linear_bwd_weight(dy, x, dW, compute=self.cmp))

    def _wg_flush(self):
        items, self._wg_pending = self._wg_pending, []
        slots, self._wg_slots = self._wg_slots, set()
        if items:
            cmp = self.cmp
            self._side(lambda: ops.linear_bwd_weight_group(items, compute=cmp))
            # A collected gradient may be handed over AFTER its sub-layer ended: the slot event _sub_end recorded then does
            # not cover this launch.  Re-record the event of every scratch set the group reads, behind the group (ADVICE r5:
            # with A3T_SIDE_DEPTH=8 the main stream could otherwise reuse tmp.gm.{slot} under the group kernel).
            if self.side is not None:
                for sl in slots:
                    ev = self._owned_event(("slot", sl))
                    ev.record(self.side)
                    self._side_ev[sl] = ev

    def join_side(self):
        """Make the current stream wait for every weight gradient issued so far (for on_group_done hooks that read gradients)."""
        self._side_join()

    def _side_join(self):
        self._wg_flush()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
            if self.side2 is not self.side:
                torch.cuda.current_stream().wait_stream(self.side2)
            self._side_ev = [None] * self._depth

    def _pre_ln(self, ga, g, g16):
        """The sub-layer's closing LayerNorm backward rewrites g / g16 in place: if the side-stream GEMMs read the
        gradient from there (no dropout copy), they must finish first."""
        if ga is g or ga is g16:
            self._wg_flush()              # (collected Linear weight gradients read the gradient in place too)
        if self.side is not None and (ga is g or ga is g16):
            torch.cuda.current_stream().wait_stream(self.side)

    # ------------------------------------------------------------------ FFN (MultiLayeredConv1d)
    def _ffn_fwd(self, tag, pre, x, T):
        p, c = self.store.p, self.c
        M = x.shape[0]
        pad = (c.ff_kernel - 1) // 2
        y = self._ln_fwd(tag + ".ln", x, pre + ".ln")
        h = self._act(tag + ".h", (M, c.ff))
        keep, lay = None, 0
        if self._ffn_plan(M)[0]:  # one bit per element of h (value > 0 after relu / dropout) for the backward mask
            keep = self.ws.get(tag + ".keep", (ops.gemm_keep_bytes(M, c.ff),), torch.uint8)
        elif self._ffn_plan(M)[3]:      # the same bits as a row-major nibble image (128-row kernel -> panel kernel)
            keep, lay = self.ws.get(tag + ".keep4", (M * c.ff // 4,), torch.uint8), 1
        ops.conv_fwd(y, self.W(pre + ".w1"), h, T, pad, bias=p[pre + ".b1"], act=ACT_RELU, compute=self.cmp,
                     drop=self._drop(c.dropout_rate, tag + ".h"), keep_out=keep, keep_layout=lay)
        xo = self.ws.get(tag + ".xo", (M, c.adim))
        ops.conv_fwd(h, self.W(pre + ".w2"), xo, T, pad, bias=p[pre + ".b2"], R=x, alpha=0.5, compute=self.cmp,
                     drop=self._drop(c.dropout_rate, tag + ".o"))
        self.sv[tag] = (y, h, keep, lay)
        return xo

    def _ffn_bwd(self, tag, pre, g, T, nb=None, nxt=None):
        """g = grad wrt the sub-layer output (fp32 residual stream, updated in place to the grad wrt
        the sub-layer input); in bf16 mode grad.x16 holds the same values in bf16 on entry and exit."""
        p, gr, c = self.store.p, self.store.g, self.c
        y, h, keep, lay = self.sv[tag]
        M = g.shape[0]
        pad = (c.ff_kernel - 1) // 2
        self._sub_begin()
        g16 = self._g16(g)
        ga = self._gm(g, tag + ".o", c.dropout_rate, gr[pre + ".b2"], 0.5)
        self._side(lambda: ops.conv_bwd_weight(ga, h, gr[pre + ".w2"], T, pad, alpha=0.5, compute=self.cmp))
        dh = self._act(self._t("tmp.dh"), (M, c.ff))
        # (without dropout b2's gradient = 0.5*colsum(g) was accumulated by the LayerNorm backward that
        #  produced g; the dropout on h folds into the relu mask S=h>0 and the 1/(1-p) factor)
        hd = self._drop(c.dropout_rate, tag + ".h")
        a_dh = 0.5 / (1.0 - hd[0]) if hd else 0.5
        if keep is not None:     # k-contiguous conv of ga with the transposed weights, masked by the forward's keep bits
            ops.conv_fwd(ga, self._wt["w2"][2][pre + ".w2"], dh, T, c.ff_kernel - 1 - pad, alpha=a_dh, compute=self.cmp,
                         keep_in=keep, keep_layout=lay, colsum=gr[pre + ".b1"])
        elif self._ffn_plan(M)[2]:
            ops.conv_fwd(ga, self._wt["w2"][2][pre + ".w2"], dh, T, c.ff_kernel - 1 - pad, alpha=a_dh, compute=self.cmp,
                         S=h, colsum=gr[pre + ".b1"])
        else:
            ops.conv_bwd_data(ga, self.W(pre + ".w2"), dh, T, pad, S=h, alpha=a_dh,
                              compute=self.cmp, colsum=gr[pre + ".b1"] if self.bf16 else None)
        self._side(lambda: ops.conv_bwd_weight(dh, y, gr[pre + ".w1"], T, pad, compute=self.cmp))
        dy = self._act("tmp.dy", (M, c.adim))
        if self._ffn_plan(M)[1]:
            ops.conv_fwd(dh, self._wt["w1"][2][pre + ".w1"], dy, T, c.ff_kernel - 1 - pad, compute=self.cmp)
        else:
            ops.conv_bwd_data(dh, self.W(pre + ".w1"), dy, T, pad, compute=self.cmp)
        if not self.bf16:
            self._bias_grad(dh, gr[pre + ".b1"])
        self._pre_ln(ga, g, g16)
        self._ln_bwd(tag + ".ln", dy, pre + ".ln", g, g, g16, nb, nxt)
        self._sub_end()
        return g

    # ------------------------------------------------------------------ rel-pos self-attention
    def _mha_fwd(self, tag, pre, x, pos, keymask, B, T):
        p, c = self.store.p, self.c
        d, H, dk = c.adim, c.heads, c.dk
        M = B * T
        cmp = self.cmp
        y = self._ln_fwd(tag + ".ln", x, pre + ".ln")
        qkv = self._act(tag + ".qkv", (M, 3 * d))
        ops.linear_fwd(y, self.W(pre + ".wqkv"), qkv, bias=p[pre + ".bqkv"], compute=cmp)
        # q + pos_bias_u / q + pos_bias_v (attention.py:190-194): the fused kernels add the biases as they load their query
        # fragments (bit for bit what a3t_add_pos_bias stores); the two [M][d] tensors only exist for the materialised forward and,
        # in the backward, as operands of the dK / d linear_pos products (made there, off the main stream)
        fused = (self._fused_now or self._fused_train_now) and ops.attn_fused_supported(dk, T)
        pbias = (p[pre + ".u"], p[pre + ".v"])
        qu = qv = None
        if not fused:
            qu = self._act(tag + ".qu", (M, d))
            qv = self._act(tag + ".qv", (M, d))
            ops.add_pos_bias(qkv, pbias[0], pbias[1], qu, qv)
        P = getattr(self, "_P_ahead", {}).get(tag)
        if P is not None:       # projected ahead on the side stream (forward())
            ev = self._pos_ev.get(tag[:3]) if self._pos_ev else None
            if ev is not None:
                ev.wait_on(torch.cuda.current_stream())
                self._pos_ev[tag[:3]] = None
        else:
            P = self._act(tag + ".P", (T, d))
            ops.linear_fwd(pos, self.W(pre + ".wpos"), P, compute=cmp)
        if self._fused_now and ops.attn_fused_supported(dk, T):
            adr = self._drop(c.attention_dropout_rate, tag + ".att")
            ctx = self._act(tag + ".ctx", (M, d))
            lse = self.ws.get(tag + ".lse", (B, H, T))
            ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk), drop=adr or (0.0, 0),
                         pos_bias=pbias if qu is None else None)
            xo = self.ws.get(tag + ".xo", (M, d))
            ops.linear_fwd(ctx, self.W(pre + ".wo"), xo, bias=p[pre + ".bo"], R=x, compute=cmp,
                           drop=self._drop(c.dropout_rate, tag + ".o"))
            self.sv[tag] = None          # forward-only pass: nothing is kept for a backward
            self.sv[tag + ".fused"] = True
            return xo
        self.sv.pop(tag + ".fused", None)
        self.sv.pop(tag + ".rs", None)
        self.sv.pop(tag + ".signed", None)
        if self._fused_train_now and ops.attn_fused_supported(dk, T):
            # fused forward, materialised backward: probabilities stay un-normalised (1 / row sum in `rs`)
            adr = self._drop(c.attention_dropout_rate, tag + ".att")
            probs = self._act(tag + ".probs", (B, H, T, T))
            # attention dropout: ONE saved tensor -- the mask rides on the sign bits of probs (the score-gradient kernel and the dV
            # product read it there); the dropped copy only exists for the materialised score gradients (A3T_ATTN_BWD_DS=0)
            signed = bool(adr) and self.bf16 and self.attn_bwd_ds and self.attn_signed and T <= 2048     # (= the backward's ds_fused)
            pdrop = self._act(tag + ".pdrop", (B, H, T, T)) if adr and not signed else None
            ctx = self._act(tag + ".ctx", (M, d))
            lse = self.ws.get(tag + ".lse", (B, H, T))
            rs = self.ws.get(tag + ".rs", (B, H, T))
            ops.attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, pdrop, rs, B, H, T, 1.0 / math.sqrt(dk),
                               drop=adr or (0.0, 0), pos_bias=pbias if qu is None else None)
            self.sv[tag + ".signed"] = signed
            xo = self.ws.get(tag + ".xo", (M, d))
            ops.linear_fwd(ctx, self.W(pre + ".wo"), xo, bias=p[pre + ".bo"], R=x, compute=cmp,
                           drop=self._drop(c.dropout_rate, tag + ".o"))
            self.sv[tag] = (y, qkv, qu, qv, P, probs, ctx, pos, pdrop)
            self.sv[tag + ".rs"] = rs
            return xo
        # score-sized scratch: fp32 in fp32 mode; bf16 logits (fp32 softmax math) in bf16 mode
        sdt = torch.bfloat16 if self.bf16 else torch.float32
        ac = self.ws.get("tmp.ac", (B, H, T, T), sdt)
        bd = self.ws.get("tmp.bd", (B, H, T, T), sdt)
        kk = qkv.view(-1)[d:]
        vv = qkv.view(-1)[2 * d:]
        # ac[b,h] = (q+u) k^T ; bd[b,h] = (q+v) P_h^T   (attention.py:190-203)
        ops.gemm(qu, kk, ac, T, T, dk, d, 1, 3 * d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk),
                 b_bs=(T * 3 * d, dk), c_bs=(H * T * T, T * T), compute=cmp)
        ops.gemm(qv, P, bd, T, T, dk, d, 1, d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk), b_bs=(0, dk),
                 c_bs=(H * T * T, T * T), compute=cmp)
        probs = self._act(tag + ".probs", (B, H, T, T))
        adr = self._drop(c.attention_dropout_rate, tag + ".att")
        pdrop = self._act(tag + ".pdrop", (B, H, T, T)) if adr else None
        ops.relpos_softmax_fwd(ac, bd, keymask, probs, B, H, T, 1.0 / math.sqrt(dk), probs_drop=pdrop,
                               drop=adr or (0.0, 0))
        ctx = self._act(tag + ".ctx", (M, d))
        # ctx[b,:,h,:] = dropout(probs[b,h]) V[b,h]
        ops.gemm(pdrop if adr else probs, vv, ctx, T, dk, T, T, 1, 1, 3 * d, d, batch=B * H, batch_inner=H,
                 a_bs=(H * T * T, T * T), b_bs=(T * 3 * d, dk), c_bs=(T * d, dk), compute=cmp)
        xo = self.ws.get(tag + ".xo", (M, d))
        ops.linear_fwd(ctx, self.W(pre + ".wo"), xo, bias=p[pre + ".bo"], R=x, compute=cmp,
                       drop=self._drop(c.dropout_rate, tag + ".o"))
        self.sv[tag] = (y, qkv, qu, qv, P, probs, ctx, pos, pdrop)
        return xo

    def _mha_bwd(self, tag, pre, g, B, T, nb=None, nxt=None):
        p, gr, c = self.store.p, self.store.g, self.c
        d, H, dk = c.adim, c.heads, c.dk
        M = B * T
        cmp = self.cmp
        y, qkv, qu, qv, P, probs, ctx, pos, pdrop = self.sv[tag]
        scale = 1.0 / math.sqrt(dk)
        self._sub_begin()
        g16 = self._g16(g)
        ga = self._gm(g, tag + ".o", c.dropout_rate, gr[pre + ".bo"], 1.0)
        self._lin_wgrad(ga, ctx, gr[pre + ".wo"])
        dctx = self._act("tmp.dctx", (M, d))
        self._lin_dgrad(ga, pre + ".wo", dctx)
        kk = qkv.view(-1)[d:]
        vv = qkv.view(-1)[2 * d:]
        dqkv = self._act(self._t("tmp.dqkv"), (M, 3 * d))
        dkk = dqkv.view(-1)[d:]
        dvv = dqkv.view(-1)[2 * d:]
        sdt = torch.bfloat16 if self.bf16 else torch.float32
        rs = self.sv.get(tag + ".rs")      # fused training forward: probs / pdrop are exp(s - m_ref), rs = 1 / row sum
        dctx_v = dctx
        if rs is not None:                 # dV = pdrop^T (rs * dctx): fold the row normalisation into the small operand
            dctx_v = self._act(self._t("tmp.dctxs"), (M, d))      # (scaled on the side stream, in front of the one GEMM that reads it)
        signed = bool(self.sv.get(tag + ".signed"))      # the forward saved ONE tensor: |probs| = the probability, sign = dropped
        adr = self._drop(c.attention_dropout_rate, tag + ".att") if (pdrop is not None or signed) else None
        # bf16 path: the attention-dropout mask comes back from the counter RNG (same key and index as the forward) instead
        # of being read off the dropped probabilities: one T x T read less in the most HBM-bound kernel of the step
        regen = self.bf16 and self.attn_regen and adr is not None and T % 8 == 0 and T <= 2048
        # saved un-normalised probabilities + counter-RNG mask: dS / dBD in one launch, dprobs never stored (a3t_attn_bwd_ds)
        ds_fused = (rs is not None and self.bf16 and self.attn_bwd_ds and (adr is None or regen) and dk % 32 == 0 and dk <= 192
                    and dk != 160 and T % 8 == 0)
        zb = (H * T * T, T * T)
        assert ds_fused or not signed
        if not ds_fused:
            dpr = self.ws.get("tmp.ac", (B, H, T, T), sdt)      # reuse the score buffers
            # dprobs[b,h] = dctx[b,:,h,:] V[b,h]^T
            ops.gemm(dctx, vv, dpr, T, T, dk, d, 1, 3 * d, 1, T, batch=B * H, batch_inner=H, a_bs=(T * d, dk),
                     b_bs=(T * 3 * d, dk), c_bs=zb, compute=cmp)
        # dV[b,h] = probs[b,h]^T dctx[b,:,h,:]
        fz = self.bf16   # bias / pos-bias gradients ride on the GEMM epilogues as column sums
        gbq = gr[pre + ".bqkv"]
        # The four N = d_k GEMMs below run as ONE round of co-resident tiles that all finish together: 576 atomics per
        # bias column in one burst cost ~40 us per GEMM.  Their column sums are therefore spread over S accumulator
        # copies (slot = (row tile + utterance) % S; arena slot, zeroed once per backward) and folded by one small kernel.
        S = self.colsum_slots
        sl = self._arena_slot("bwd32", tag + ".bsl", S * 4 * d) if fz else None
        csk = dict(colsum_bs1=dk, colsum_slots=S, colsum_ss=4 * d)
        # (independent of the dprobs -> softmax-backward chain: runs beside it on the side stream)
        make_q = qu is None        # fused forward: (q+u), (q+v) were never stored -- the dK product (second side stream, below) and
        if make_q:                 # the gradient of linear_pos (first side stream) read them: made on the second side stream
            qu = self._act("tmp.qu", (M, d))                  # (read on that stream only, in order)
            qv = self._act(self._t("tmp.qv"), (M, d))         # (read by the other side stream, whenever it gets there)

        def dv_gemm():
            if rs is not None:
                ops.attn_scale_rows(dctx, rs, dctx_v, B, H, T)
            ops.gemm(pdrop if pdrop is not None else probs, dctx_v, dvv, T, dk, T, 1, T, 1, d, 3 * d,
                     batch=B * H, batch_inner=H, a_bs=zb, b_bs=(T * d, dk), c_bs=(T * 3 * d, dk),
                     compute=cmp, colsum=sl[3 * d:] if fz else None, a_signmask=signed,
                     alpha=1.0 / (1.0 - adr[0]) if signed else 1.0, **csk)
            if make_q:      # behind dV (which does not read them): dK cannot start before the score gradients exist anyway
                ops.add_pos_bias(qkv, p[pre + ".u"], p[pre + ".v"], qu, qv)
        qv_ready = self._side(dv_gemm, want_event=make_q, urgent=True)
        zbd = zb
        # dq = ds K + dbd P as one launch of the streaming kernel where it runs (below); and the compact dBD matrix is not stored at all: it
        # is the flat dS sequence shifted by T - 1 elements, read by its two consumers as a strided VIEW of dS (row stride T + 1,
        # 2-byte aligned rows: the LDS-DMA takes them) with T zeros in front of every (b, h) block for the entries of its first row
        # that never reach the scores (attention.py:157-165).  A3T_ATTN_DBD_VIEW=0: the stored matrix.
        dual = ds_fused and fz and self.attn_dq_dual and ops.gemm_tt_supported(T, dk, T, B * H)
        dview = ds_fused and fz and self.attn_dbd_view
        if dview:
            bsv = T + T * T
            flat = self.ws.get(self._t("tmp.dsv"), (B * H * bsv,), torch.bfloat16, zero_once=True)
            ds, dbd = flat[T:], flat[1:]
            zbd = (H * bsv, bsv)       # (batch strides of ds and of its dbd view; zb stays the probabilities')
        elif self.bf16:
            ds = self.ws.get("tmp.ds16", (B, H, T, T), torch.bfloat16)
            dbd = self.ws.get(self._t("tmp.dbd16"), (B, H, T, T), torch.bfloat16)
        else:
            ds = dpr
            dbd = self.ws.get(self._t("tmp.dbd"), (B, H, T, T), sdt)
        if ds_fused:
            ops.attn_bwd_ds(dctx, ctx, qkv, probs, rs, ds, None if dview else dbd, B, H, T, scale, drop=adr or (0.0, 0),
                            signed_probs=signed, ds_bs=zbd[1] if dview else 0)
        else:
            ops.relpos_softmax_bwd(probs, dpr, ds, dbd, B, H, T, scale, probs_drop=None if regen else pdrop,
                                   drop_p=adr[0] if adr else 0.0, drop_key=adr[1] if regen else 0, rowscale=rs)
        def pos_weight_grad():   # dP_h += sum_b dbd^T (q+v) -> d W_pos; only the side stream touches tmp.dP*
            if qv_ready is not None and self.side is not None:
                qv_ready.wait_on(self.side)
            dP = self._arena_slot("bwd32", tag + ".dP", T * d).view(T, d)   # cleared once per backward (main stream)
            ops.gemm(dbd, qv, dP, T, dk, T, 1, T + 1 if dview else T, 1, d, d, batch=B * H, batch_inner=H, a_bs=zbd, b_bs=(T * d, dk),
                     c_bs=(0, dk), acc=ACC_ATOMIC, compute=cmp, a_view=dview)
            if self.bf16:
                dP16 = self.ws.get("tmp.dP16", (T, d), torch.bfloat16)
                ops.cast_bf16(dP, dP16)
                dP = dP16
            ops.linear_bwd_weight(dP, pos, gr[pre + ".wpos"], compute=cmp)
        self._side(pos_weight_grad)
        # bf16 path: dq = d(q+u) + d(q+v) goes straight into the q third of dqkv -- the first product stores, the second adds (sum
        # in fp32, one rounding); pos_bias_u / pos_bias_v / linear_q.bias take their gradients from the products' own column sums.
        # fp32 path: both products are kept (their plain column sums are the bias gradients) and added by a3t_add_pos_bias_bwd.
        dq_acc = fz
        dqu = dqkv if dq_acc else self._act("tmp.dqu", (M, d))
        dqv = dqkv if dq_acc else self._act("tmp.dqv", (M, d))
        ldq, cbq = (3 * d, (T * 3 * d, dk)) if dq_acc else (d, (T * d, dk))
        # dqu[b,h] = ds K ; dK[b,h] = ds^T (q+u)
        # bf16 path on the streaming kernel: dq = ds K + dbd P as ONE launch (two K loops into one accumulator set, one rounding, no
        # read-modify-write of the q third; their column sums come apart at the hand-over) -- A3T_ATTN_DQ_DUAL=0: two launches
        if dual:
            ops.gemm(ds, kk, dqkv, T, dk, T, T, 1, 1, 3 * d, ldq, batch=B * H, batch_inner=H, a_bs=zbd,
                     b_bs=(T * 3 * d, dk), c_bs=cbq, compute=cmp, colsum=sl,
                     second=(dbd, P, d, (0, dk), sl[d:]) + ((T + 1,) if dview else ()), **csk)
        else:
            ops.gemm(ds, kk, dqu, T, dk, T, T, 1, 1, 3 * d, ldq, batch=B * H, batch_inner=H, a_bs=zbd,
                     b_bs=(T * 3 * d, dk), c_bs=cbq, compute=cmp, colsum=sl if fz else None, **csk)
        dk_fn = lambda: ops.gemm(ds, qu, dkk, T, dk, T, 1, T, 1, d, 3 * d, batch=B * H, batch_inner=H,
                                 a_bs=zbd, b_bs=(T * d, dk), c_bs=(T * 3 * d, dk), compute=cmp,
                                 colsum=sl[2 * d:] if fz else None, **csk)
        if self.attn_dk_main and dual and self.side is not None and make_q and qv_ready is not None:
            # dK on the MAIN queue behind dq where the products run on the streaming kernel (second session): they take turns on the
            # CUs whatever queue they sit on (128 KiB of LDS each), and a hand-over event behind the last of them cost the main queue
            # ~25 us of bubble per layer (profiles/r06_trace_attn_layer.txt): 39.52 / 39.56 -> 39.41 / 39.39 ms.  On the 128-row kernel
            # (configs[3]) dK and dq do run side by side: 58.2 -> 58.6 ms with dK on the main queue, so it stays on the side queue
            # there.  (q + u) and dV come from the second side queue: its event is long past here.
            qv_ready.wait_on(torch.cuda.current_stream())
            dk_fn()
            dk_done = None
        else:
            dk_done = self._side(dk_fn, want_event=True, urgent=True)
        # dqv[b,h] = dbd P_h ; dP_h += sum_b dbd^T (q+v)
        if not dual:
            ops.gemm(dbd, P, dqv, T, dk, T, T + 1 if dview else T, 1, 1, d, ldq, batch=B * H, batch_inner=H, a_bs=zbd, b_bs=(0, dk),
                     c_bs=cbq, acc=ACC_ADD if dq_acc else ACC_STORE, compute=cmp, colsum=sl[d:] if fz else None, a_view=dview, **csk)
        if not dq_acc:
            ops.add_pos_bias_bwd(dqu, dqv, dqkv)      # (the dq slice only)
        if dk_done is not None:      # the dV / dK slices of dqkv and their column sums come from the side stream
            dk_done.wait_on(torch.cuda.current_stream())
        if fz:   # d u, d v, d b_q = d u + d v, d b_k, d b_v from the slot sums (all four GEMMs have drained here); feeds nothing
            self._side(
                lambda: ops.attn_bias_fold(sl, S, d, gr[pre + ".u"], gr[pre + ".v"], gbq))       # but the optimizer
        else:
            self._bias_grad(dqu, gr[pre + ".u"])
            self._bias_grad(dqv, gr[pre + ".v"])
            self._bias_grad(dqkv, gbq)
        self._lin_wgrad(dqkv, y, gr[pre + ".wqkv"])
        dy = self._act("tmp.dy", (M, d))
        self._lin_dgrad(dqkv, pre + ".wqkv", dy)
        self._pre_ln(ga, g, g16)
        self._ln_bwd(tag + ".ln", dy, pre + ".ln", g, g, g16, nb, nxt)
        self._sub_end()
        return g

    # ------------------------------------------------------------------ convolution module
    def _bn_fwd(self, tag, z, pre, bufpre, act, out):
        p, b = self.store.p, self.store.buf
        M, C = z.shape
        stats = self._arena_slot("fwd64", tag + ".stats", 2 * C)      # cleared once per forward
        if self.training:
            ops.col_reduce(z, stats[:C], stats[C:], mode=1)
        mean = self.ws.get(tag + ".bnmean", (C,))
        rstd = self.ws.get(tag + ".bnrstd", (C,))
        ops.bn_act_fwd(z, stats, p[pre + ".g"], p[pre + ".b"], b[bufpre + ".rm"], b[bufpre + ".rv"], mean, rstd, out,
                       1e-5, self.bn_momentum if self.training else 0.0, self.training, act)
        if self.training:
            self.store.nbt[bufpre] = self.store.nbt.get(bufpre, 0) + 1
        self.sv[tag + ".bn"] = (z, mean, rstd)

    def _bn_bwd(self, tag, dy, pre, act, dz):
        p, gr = self.store.p, self.store.g
        z, mean, rstd = self.sv[tag + ".bn"]
        M, C = z.shape
        sums = self._arena_slot("bwd64", tag + ".bnsums", 2 * C)      # cleared once per backward
        ops.bn_act_bwd(dy, z, mean, rstd, p[pre + ".g"], p[pre + ".b"], sums, dz, gr[pre + ".g"], gr[pre + ".b"],
                       self.training, act, zero=False)

    def _conv_fwd(self, tag, pre, x, T):
        p, c = self.store.p, self.c
        M, d = x.shape
        cmp = self.cmp
        y = self._ln_fwd(tag + ".ln", x, pre + ".ln")
        g2 = self._act(tag + ".g", (M, 2 * d))
        ops.linear_fwd(y, self.W(pre + ".pw1"), g2, bias=p[pre + ".pb1"], compute=cmp)
        glu = self._act(tag + ".glu", (M, d))
        z = self.ws.get(tag + ".z", (M, d))
        ops.glu_dwconv_fwd(g2, p[pre + ".dw"], p[pre + ".db"], glu, z, T)
        s = self._act(tag + ".s", (M, d))
        self._bn_fwd(tag, z, pre + ".bn", pre + ".bn", ACT_SWISH, s)
        xo = self.ws.get(tag + ".xo", (M, d))
        ops.linear_fwd(s, self.W(pre + ".pw2"), xo, bias=p[pre + ".pb2"], R=x, compute=cmp,
                       drop=self._drop(c.dropout_rate, tag + ".o"))
        self.sv[tag] = (y, g2, glu, s)
        return xo

    def _conv_bwd(self, tag, pre, g, T, nb=None, nxt=None):
        p, gr, c = self.store.p, self.store.g, self.c
        M, d = g.shape
        cmp = self.cmp
        y, g2, glu, s = self.sv[tag]
        self._sub_begin()
        g16 = self._g16(g)
        ga = self._gm(g, tag + ".o", c.dropout_rate, gr[pre + ".pb2"], 1.0)
        self._lin_wgrad(ga, s, gr[pre + ".pw2"])
        ds = self.ws.get("tmp.ds", (M, d))
        self._lin_dgrad(ga, pre + ".pw2", ds)
        dz = self.ws.get("tmp.dz", (M, d))
        self._bn_bwd(tag, ds, pre + ".bn", ACT_SWISH, dz)
        dg = self._act(self._t("tmp.dg"), (M, 2 * d))
        ops.glu_dwconv_bwd(dz, g2, glu, p[pre + ".dw"], dg, gr[pre + ".dw"], gr[pre + ".db"], T,
                           dgsum=gr[pre + ".pb1"])
        self._lin_wgrad(dg, y, gr[pre + ".pw1"])
        dy = self._act("tmp.dy", (M, d))
        self._lin_dgrad(dg, pre + ".pw1", dy)
        self._pre_ln(ga, g, g16)
        self._ln_bwd(tag + ".ln", dy, pre + ".ln", g, g, g16, nb, nxt)
        self._sub_end()
        return g

    # ------------------------------------------------------------------ one Conformer block
    def block_fwd(self, pre, x, pos, keymask, B, T):
        # (encoder_layer.py:117-181)
        x = self._ffn_fwd(pre + ".ffm", pre + ".ffm", x, T)
        x = self._mha_fwd(pre + ".mha", pre + ".mha", x, pos, keymask, B, T)
        x = self._conv_fwd(pre + ".cnv", pre + ".cnv", x, T)
        x = self._ffn_fwd(pre + ".ff", pre + ".ff", x, T)
        return self._ln_fwd(pre + ".fin", x, pre + ".fin.ln", out_dtype=torch.float32)

    def block_bwd(self, pre, g, B, T):
        gr = self.store.g
        # every LayerNorm backward also delivers the bias gradient (and, with dropout, the masked gradient operand) of
        # the layer whose output it read = the next sub-layer of this schedule
        self._ln_bwd(pre + ".fin", g, pre + ".fin.ln", None, g, self._g16(g), (gr[pre + ".ff.b2"], 0.5), pre + ".ff.o")
        self._ffn_bwd(pre + ".ff", pre + ".ff", g, T, (gr[pre + ".cnv.pb2"], 1.0), pre + ".cnv.o")
        self._conv_bwd(pre + ".cnv", pre + ".cnv", g, T, (gr[pre + ".mha.bo"], 1.0), pre + ".mha.o")
        self._mha_bwd(pre + ".mha", pre + ".mha", g, B, T, (gr[pre + ".ffm.b2"], 0.5), pre + ".ffm.o")
        self._ffn_bwd(pre + ".ffm", pre + ".ffm", g, T, None)
        return g

    # ------------------------------------------------------------------ whole model
    def forward(self, batch: Dict[str, torch.Tensor], need_grad: bool = True, gscale: float = 1.0):
        c, p, ws = self.c, self.store.p, self.ws
        speech = batch["speech"].contiguous()
        B, Tm, idim = speech.shape
        text = batch["text"].contiguous()
        Tp = text.shape[1]
        T = Tm + Tp
        d = c.adim
        if self.bf16 and (T % 8 or Tm % 8):
            raise ValueError(f"compute='bf16' needs T_mel and T_mel+T_phn to be multiples of 8, got {Tm}, {T}")
        self.dims = (B, Tm, Tp, T)
        self._fused_now = (self.fused_attn_fwd_only and not need_grad) or \
            (self.fused_attn_auto and not need_grad and B * c.heads * ((T + 127) // 128) >= 64)
        nblk = B * c.heads * ((T + 127) // 128)
        self._fused_train_now = self.fused_attn_train and need_grad and not self._fused_now and \
            self.fused_attn_train_min <= nblk <= 65536 and T <= 2048
        self._need_grad = bool(need_grad)
        self._mode_tag = ops.gemm_mode_tag()
        self.step_seed += 1
        self.refresh_weights()
        self._arena_clear("fwd64")
        pp = c.positional_dropout_rate
        masked = batch["masked_position"].contiguous().view(torch.uint8)
        keymask = ws.get("keymask", (B, T), torch.uint8)
        keymask[:, :Tm].copy_(batch["speech_mask"].reshape(B, Tm).view(torch.uint8))
        keymask[:, Tm:].copy_(batch["text_mask"].reshape(B, Tp).view(torch.uint8))
        spos = batch["speech_segment_pos"].contiguous()
        tpos = batch["text_segment_pos"].contiguous()
        speech2 = speech.view(B * Tm, idim)
        # --- encoder prologue (conformer/encoder.py:522-553)
        xm = self._act("emb.xm", (B * Tm, idim))
        ops.mask_fill(speech2, masked, p["mask_feature"], xm)
        e0 = ws.get("emb.e0", (B * Tm, d))
        ops.linear_fwd(xm, self.W("emb.w"), e0, bias=p["emb.b"], compute=self.cmp)
        e = self._ln_fwd("emb.ln", e0, "emb.ln", eps=1e-5, out_dtype=torch.float32)
        xs = ws.get("emb.xs", (B * T, d))
        xscale = math.sqrt(d)
        spk = None
        if c.spk_embed_dim > 0 and batch.get("spembs") is not None:
            # x-vector conditioning (configs[3]): Linear(spembs) added to every token of the utterance, fused into the
            # prologue kernel.  fp32 (B x 512 x d: negligible); an extension, the reference ignores spembs
            se = batch["spembs"].to(self.dev, torch.float32).contiguous()
            spk = ws.get("emb.spk", (B, d))
            ops.linear_fwd(se, p["spk.w"], spk, bias=p["spk.b"], compute=F32)
            self.sv["spk"] = se
        else:
            self.sv.pop("spk", None)
        ops.embed_finish_fwd(e, p["temb"], p["seg"], text, spos, tpos, xs, B, Tm, Tp, d, xscale,
                             drop=self._drop(pp, "emb.x") or (0.0, 0), spk=spk)
        pos_e = self._act("pos.enc", (T, d))
        pos_d = self._act("pos.dec", (T, d))
        pos_on_side = None
        if self._drop(pp, "pos.enc"):      # dropout(pos_emb) (embedding.py:170)
            def drop_pos():
                pf = ws.get("pos.f32", (T, d))
                pf[:Tm].copy_(self.pe[:Tm])
                pf[Tm:].copy_(self.pe[:Tp])
                ops.dropout(pf, pos_e, *self._drop(pp, "pos.enc"))
                pf.copy_(self.pe[:T])
                ops.dropout(pf, pos_d, *self._drop(pp, "pos.dec"))
            if self.side is not None and self._pos_ahead:
                pos_on_side = drop_pos       # only the positional projections (side stream, below) read the dropped tables
            else:
                drop_pos()
            ws.pos_key = None
        elif getattr(ws, "pos_key", None) != (Tm, Tp, pos_e.data_ptr(), pos_d.data_ptr()):
            # (constant for a given (T_mel, T_phn): rebuilt only when the shape or the workspace buffers change)
            pos_e[:Tm].copy_(self.pe[:Tm])
            pos_e[Tm:].copy_(self.pe[:Tp])
            pos_d.copy_(self.pe[:T])
            ws.pos_key = (Tm, Tp, pos_e.data_ptr(), pos_d.data_ptr())     # (on the workspace: engines may share it)
        self.sv["embed"] = (xm, e, text, spos, tpos, masked, speech2)
        # linear_pos(pos_emb) of every block (attention.py:188-189) depends on nothing but the table and the weights: all of
        # them are projected up front on the side stream, which is idle in the forward (A3T_POS_AHEAD=0: inside each block)
        self._P_ahead, self._pos_ev = {}, None
        if self.side is not None and self._pos_ahead:
            def project(kind, posx, n):
                def run():
                    if kind == "enc" and pos_on_side is not None:
                        pos_on_side()
                    for i in range(n):
                        tag = f"{kind}.{i}.mha"
                        P = self._act(tag + ".P", (T, d))
                        ops.linear_fwd(posx, self.W(tag + ".wpos"), P, compute=self.cmp)
                        self._P_ahead[tag] = P
                return run
            ev_e = self._side(project("enc", pos_e, c.enc_blocks), want_event="pos.enc")
            ev_d = self._side(project("dec", pos_d, c.dec_blocks), want_event="pos.dec")     # (behind the decoder-side weight cast)
            self._pos_ev = {"enc": ev_e, "dec": ev_d}
        x = xs
        for i in range(c.enc_blocks):
            x = self.block_fwd(f"enc.{i}", x, pos_e, keymask, B, T)
        x = self._ln_fwd("enc.after", x, "enc.after", out_dtype=torch.float32)
        # --- decoder (conformer/encoder.py:568-614): x*sqrt(d), contiguous rel-pos table
        xd = ws.get("dec.in", (B * T, d))
        dd = self._drop(pp, "dec.x")
        if dd:
            ops.dropout(x, xd, dd[0], dd[1], scale=xscale)
        else:
            ops.scale(x, xd, xscale)
        x = xd
        self._wait_cast("_cast_ev")       # decoder / head shadows cast on the side stream (refresh_weights)
        for i in range(c.dec_blocks):
            x = self.block_fwd(f"dec.{i}", x, pos_d, keymask, B, T)
        x = self._ln_fwd("dec.after", x, "dec.after", out_dtype=torch.float32)
        # --- head: slice speech frames, sfc, postnet, loss (sedit_model.py:363-372,320-340)
        hs = self._act("head.hs", (B * Tm, d))
        ops.slice_rows(x, hs, B, T, Tm, d)
        before = ws.get("head.before", (B * Tm, c.odim))
        if self.bf16 and self.sfc_f32:
            # the mel projection on the exact-fp32 MFMA from the fp32 LayerNorm output and the fp32 master weights (B*T_mel x 80 x d:
            # ~20 us): `before` is what the postnet amplifies, its last rounding step is the cheapest one to remove
            hs32 = ws.get("head.hs32", (B * Tm, d))
            ops.slice_rows(x, hs32, B, T, Tm, d)
            ops.linear_fwd(hs32, p["sfc.w"], before, bias=p["sfc.b"], compute=F32)
        else:
            ops.linear_fwd(hs, self.W("sfc.w"), before, bias=p["sfc.b"], compute=self.cmp)
        y = before
        f32_first = self.bf16 and self.post_f32_first
        ylo = None
        if self.bf16 and c.postnet_layers > 0:
            y = ws.get("head.before16", (B * Tm, c.odim), torch.bfloat16)
            if f32_first:
                ylo = ws.get("head.before16lo", (B * Tm, c.odim), torch.bfloat16)
                ops.split_bf16(before, y, ylo)
            else:
                ops.cast_bf16(before, y)
        pad = (c.postnet_filts - 1) // 2
        for l in range(c.postnet_layers):
            W = self.W(f"post.{l}.w")
            oc = W.shape[0]
            last = (l == c.postnet_layers - 1)
            z = ws.get(f"post.{l}.z", (B * Tm, oc))
            ops.conv_fwd(y, W, z, Tm, pad, compute=self.cmp)
            if l == 0 and ylo is not None:
                ops.conv_fwd(ylo, W, z, Tm, pad, R=z, compute=self.cmp)        # z += conv(lo)
            o = ws.get(f"post.{l}.o", (B * Tm, oc), torch.float32 if last else self.adt)
            self._bn_fwd(f"post.{l}", z, f"post.{l}.bn", f"post.{l}.bn", ACT_NONE if last else ACT_TANH, o)
            pdr = self._drop(c.postnet_dropout_rate, f"post.{l}")
            if pdr:
                ops.dropout(o, o, *pdr)
            self.sv[f"post.{l}"] = y
            y = o
        if c.postnet_layers > 0:
            after = ws.get("head.after", (B * Tm, c.odim))
            after.copy_(before)
            ops.axpy(y, after, 1.0)
        else:
            after = None          # no postnet: the loss has no after-term (sedit_model.py:333-337), after_outs = before_outs
        loss = ws.get("head.loss", (1,))
        scratch = ws.get("head.lscratch", (ops.loss_scratch_floats(B * Tm),))
        db = ws.get("head.dbefore", (B * Tm, c.odim)) if need_grad else None
        da = ws.get("head.dafter", (B * Tm, c.odim)) if (need_grad and after is not None) else None
        ops.mlm_loss(before, after, speech2, masked, loss, db, da, scratch, l2=c.lsm_weight > 50, gscale=gscale)
        self.sv["head"] = (hs, before, after, db, da)
        return dict(loss=loss, before=before.view(B, Tm, c.odim),
                    after=(after if after is not None else before).view(B, Tm, c.odim))

    def backward(self, on_group_done=None):
        """Accumulates d loss / d param (times the gscale given to forward) into store.grad.
        on_group_done(first_param_name) is called each time every parameter at or above that flat
        offset has its final gradient (hook for overlapping the gradient all-reduce)."""
        hook = on_group_done or (lambda name: None)

        def done(name):          # every gradient at or above `name` is final once the side streams have drained: a hook that
            hook(name)           # consumes gradients calls join_side() first (no join at all without one -- it stalls the main stream)
        c, p, gr, ws = self.c, self.store.p, self.store.g, self.ws
        B, Tm, Tp, T = self.dims
        d = c.adim
        cmp = self.cmp
        self._wait_cast("_cast_ev")
        self._wait_cast("_wt_ev")         # transposed weight shadows cast on the side stream (refresh_weights)
        self._arena_clear("bwd64")
        self._arena_clear("bwd32")
        hs, before, after, db, da = self.sv["head"]
        pad = (c.postnet_filts - 1) // 2
        if c.postnet_layers > 0:
            g = da                                    # grad wrt last BN output (fp32)
            for l in reversed(range(c.postnet_layers)):
                W = self.W(f"post.{l}.w")
                oc = W.shape[0]
                last = (l == c.postnet_layers - 1)
                dz = ws.get(f"tmp.post.dz{oc}.{l}", (B * Tm, oc))      # (per layer: the side stream reads it after the main stream moved on)
                pdr = self._drop(c.postnet_dropout_rate, f"post.{l}")
                if pdr:
                    gd = ws.get(f"tmp.post.gd{oc}", (B * Tm, oc))
                    ops.dropout(g, gd, *pdr)
                    g = gd
                self._bn_bwd(f"post.{l}", g, f"post.{l}.bn", ACT_NONE if last else ACT_TANH, dz)
                if self.bf16:
                    dz16 = ws.get(f"tmp.post.dz16.{oc}.{l}", (B * Tm, oc), torch.bfloat16)
                    ops.cast_bf16(dz, dz16)
                    dz = dz16
                yin = self.sv[f"post.{l}"]
                ic = yin.shape[1]
                # (weight gradients of the head: the side stream is idle at the start of the backward)
                self._side(
                    lambda dz=dz, yin=yin, l=l: ops.conv_bwd_weight(dz, yin, gr[f"post.{l}.w"], Tm, pad, compute=cmp))
                gi = ws.get(f"tmp.post.g{l % 2}.{ic}", (B * Tm, ic))
                ops.conv_bwd_data(dz, W, gi, Tm, pad, compute=cmp)
                g = gi
            ops.axpy(da, db, 1.0)                     # after = before + postnet(before)
            ops.axpy(g, db, 1.0)
        dba = db
        if self.bf16:
            dba = ws.get("tmp.db16", (B * Tm, c.odim), torch.bfloat16)
            ops.cast_bf16(db, dba)
        dhs = ws.get("tmp.dhs", (B * Tm, d))
        ops.linear_bwd_data(dba, self.W("sfc.w"), dhs, compute=cmp)
        self._side(lambda: ops.linear_bwd_weight(dba, hs, gr["sfc.w"], compute=cmp))
        self._bias_grad(db, gr["sfc.b"])
        done("sfc.w")
        g = ws.get("grad.x", (B * T, d), zero=True)
        g16 = self._g16(g)
        ops.slice_rows(g, dhs, B, T, Tm, d, reverse_add=True)
        self._ln_bwd("dec.after", g, "dec.after", None, g, g16)
        for i in reversed(range(c.dec_blocks)):
            self.block_bwd(f"dec.{i}", g, B, T)
            done(f"dec.{i}.ffm.ln.g")
        dd = self._drop(c.positional_dropout_rate, "dec.x")
        if dd:
            ops.dropout(g, g, dd[0], dd[1], scale=math.sqrt(d))
        else:
            ops.scale(g, g, math.sqrt(d))
        self._ln_bwd("enc.after", g, "enc.after", None, g, g16)
        for i in reversed(range(c.enc_blocks)):
            self.block_bwd(f"enc.{i}", g, B, T)
            done(f"enc.{i}.ffm.ln.g")
        # --- prologue backward
        if "spk" in self.sv:     # d Linear(spembs): per-utterance column sums of the gradient of the token stream
            se = self.sv["spk"]
            dspk = ws.get("tmp.dspk", (B, d), zero=True)
            ops.segment_colsum(g, dspk, B, T)
            ops.linear_bwd_weight(dspk, se, gr["spk.w"], compute=F32)
            self._bias_grad(dspk, gr["spk.b"])
        xm, e, text, spos, tpos, masked, speech2 = self.sv["embed"]
        de = ws.get("tmp.de", (B * Tm, d))
        ops.embed_finish_bwd(g, e, text, spos, tpos, de, gr["temb"], gr["seg"], B, Tm, Tp, d, c.vocab, c.seg_table,
                             math.sqrt(d), drop=self._drop(c.positional_dropout_rate, "emb.x") or (0.0, 0))
        de0 = ws.get("tmp.de0", (B * Tm, d))
        de16 = ws.get("tmp.de016", (B * Tm, d), torch.bfloat16) if self.bf16 else None
        self._ln_bwd("emb.ln", de, "emb.ln", None, de0, de16, (gr["emb.b"], 1.0))
        dea = de16 if self.bf16 else de0
        ops.linear_bwd_weight(dea, xm, gr["emb.w"], compute=cmp)
        dxm = ws.get("tmp.dxm", (B * Tm, c.idim))
        ops.linear_bwd_data(dea, self.W("emb.w"), dxm, compute=cmp)
        s64 = self.scratch64[:c.idim]
        s64.zero_()
        ops.col_reduce(dxm, s64, rowmask=masked.view(-1), mode=0)
        ops.f64_to_f32_add(s64, gr["mask_feature"], 1.0)
        done("seg")
        self._side_join()        # every gradient is final on the caller's stream when backward returns
