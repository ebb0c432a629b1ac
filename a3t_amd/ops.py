"""Thin host wrappers over the C ABI (include/a3t_hip.h): torch tensors supply device memory and
the current HIP stream, every computation happens in liba3t_hip.so.  No torch math here."""
import ctypes
import os

import torch

from . import _lib as L
from ._lib import ACC_ADD, ACC_ATOMIC, ACC_SOLE, ACC_STORE, ACT_NONE, ACT_RELU, ACT_SWISH, ACT_TANH, BF16, F32

__all__ = ["gemm", "linear_fwd", "linear_bwd_data", "linear_bwd_weight", "conv_fwd", "conv_bwd_data",
           "conv_bwd_weight"]


# bench.py sets PROFILE = [] to bracket every GEMM launch with HIP events on the launch stream:
# entries are (kernel name as rocprofv3 prints it, algorithmic FLOPs of the launch, start, end).
PROFILE = None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def gemm(A, B, C, M, N, K, a_rs, a_cs, b_rs, b_cs, c_rs, *, b_ts=0, bias=None, R=None, S=None, batch=1,
         batch_inner=1, a_bs=(0, 0), b_bs=(0, 0), c_bs=(0, 0), taps=1, pad=0, dil=1, Tseq=0, kshift=0, alpha=1.0,
         act=ACT_NONE, acc=ACC_STORE, splitk=1, compute=F32, colsum=None, colsum_bs1=0, colsum_scale=1.0,
         drop=None, colsum_slots=1, colsum_ss=0, keep_out=None, keep_in=None, a_signmask=False, keep_layout=0, second=None, a_view=False):
    """C (op)= alpha*mask(act(A(m,k) B(n,k) + bias)) + R  -- see a3t_gemm_desc in include/a3t_hip.h."""
    lib = L.load()
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.R = R.data_ptr() if R is not None else None
    d.S = S.data_ptr() if S is not None else None
    d.M, d.N, d.K = M, N, K
    d.a_rs, d.a_cs, d.b_rs, d.b_cs, d.b_ts, d.c_rs = a_rs, a_cs, b_rs, b_cs, b_ts, c_rs
    d.batch, d.batch_inner = batch, batch_inner
    d.a_bs0, d.a_bs1 = a_bs
    d.b_bs0, d.b_bs1 = b_bs
    d.c_bs0, d.c_bs1 = c_bs
    d.taps, d.pad, d.dil, d.Tseq, d.kshift = taps, pad, dil, Tseq, kshift
    d.alpha, d.act, d.accumulate, d.splitk = alpha, act, acc, splitk
    d.a_dtype, d.b_dtype, d.c_dtype, d.compute = _dt(A), _dt(B), _dt(C), compute
    d.s_dtype = _dt(S) if S is not None else F32
    d.colsum = colsum.data_ptr() if colsum is not None else None
    d.colsum_bs1, d.colsum_scale = colsum_bs1, colsum_scale
    d.colsum_slots, d.colsum_ss = colsum_slots, colsum_ss
    d.drop_p, d.drop_key = (drop if drop is not None else (0.0, 0))
    d.keep_out = keep_out.data_ptr() if keep_out is not None else None
    d.keep_in = keep_in.data_ptr() if keep_in is not None else None
    d.a_signmask = 1 if a_signmask else 0
    d.keep_layout = keep_layout
    d.a_unaligned = 1 if a_view else 0      # A is a 2-byte aligned strided view (the compact dBD read off dS)
    if second is not None:      # (A2, B2, b2_cs, (b2_bs0, b2_bs1), colsum2[, a2_rs]): C = alpha (A B + A2 B2) in one launch of the
        A2, B2, b2_cs, b2_bs, cs2 = second[:5]      # streaming kernel; a2_rs given: A2 is such a view with that row stride
        d.A2, d.B2, d.b2_cs = A2.data_ptr(), B2.data_ptr(), b2_cs
        d.b2_bs0, d.b2_bs1 = b2_bs
        d.colsum2 = cs2.data_ptr() if cs2 is not None else None
        if len(second) > 5:
            d.a2_rs = second[5]
            d.a_unaligned |= 2
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()          # torch's current stream == the stream handed to a3t_gemm
        L.check(lib.a3t_gemm(ctypes.byref(d), _stream()), "a3t_gemm")
        e1.record()
        PROFILE.append((lib.a3t_gemm_last_kernel().decode(), (2.0 if second is None else 4.0) * M * N * K * batch, e0, e1,
                        (M, N, K, batch, taps, splitk)))
        return
    L.check(lib.a3t_gemm(ctypes.byref(d), _stream()), "a3t_gemm")


_SPLITK_TARGET = 1000      # workgroups of a split-K grid on the 128x128 kernel (500 / 700 / 1500 / 2000 measured in round 3: slower)


def _splitk_for(n_tiles, K, ktile=64):
    """Token-reduction GEMMs (weight gradients) have few output tiles: split K over workgroups so the
    grid is ONE resident wave of the kernel variant a3t_gemm will pick -- ~1000 workgroups for the
    single-buffer variant (4/CU, chosen when tiles*splitk >= 768), ~440 for the double-buffered one
    (2/CU); more splits only add fp32 atomics (measured on MI355X, tools/tn_bench*.py)."""
    target = _SPLITK_TARGET if n_tiles >= 64 else 440
    s = max(1, target // max(n_tiles, 1))
    s = min(s, max(1, K // (ktile * 8)))
    return int(s)


# ---- y = x W^T (+bias): torch.nn.Linear / 1x1 Conv1d -----------------------------------------
def linear_fwd(x, W, out, bias=None, R=None, alpha=1.0, act=ACT_NONE, compute=F32, drop=None):
    M, K = x.shape
    N = W.shape[0]
    gemm(x, W, out, M, N, K, K, 1, K, 1, N, bias=bias, R=R, alpha=alpha, act=act, compute=compute, drop=drop)


def linear_bwd_data(dy, W, dx, S=None, alpha=1.0, acc=ACC_STORE, compute=F32, colsum=None):
    """dx[M,K] = alpha * relu_mask_S(dy[M,N] @ W[N,K]);  colsum += column sums of dx"""
    M, N = dy.shape
    K = W.shape[1]
    gemm(dy, W, dx, M, K, N, N, 1, 1, K, K, S=S, alpha=alpha, acc=acc, compute=compute, colsum=colsum)


def linear_bwd_weight(dy, x, dW, alpha=1.0, compute=F32):
    """dW[N,K] += alpha * dy[M,N]^T @ x[M,K]   (reduction over tokens, split-K + fp32 atomics)"""
    M, N = dy.shape
    K = x.shape[1]
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    # ACC_SOLE: a parameter gradient has one writer at a time -- kernels that fold split-K partials need no atomics for it
    gemm(dy, x, dW, N, K, M, 1, N, 1, K, K, alpha=alpha, acc=ACC_SOLE, splitk=_splitk_for(tiles, M),
         compute=compute)


def linear_bwd_weight_group(items, compute=F32):
    """items = [(dy, x, dW, alpha), ...] over the same tokens: dW_i += alpha_i * dy_i^T @ x_i.  On the bf16 path all of them run
    as ONE launch of the 128 x 384-tile token-reduction kernel (a3t_gemm_tn3_group); when the library declines (fp32 compute,
    shapes outside the kernel's contract, kernel switched off) they are launched one by one."""
    if compute == BF16 and len(items) > 1 and len(items) <= 8:
        lib = L.load()
        arr = (L.GemmDesc * len(items))()
        for d, (dy, x, dW, alpha) in zip(arr, items):
            M, N = dy.shape
            K = x.shape[1]
            d.A, d.B, d.C = dy.data_ptr(), x.data_ptr(), dW.data_ptr()
            d.M, d.N, d.K = N, K, M
            d.a_rs, d.a_cs, d.b_rs, d.b_cs, d.b_ts, d.c_rs = 1, N, 1, K, 0, K
            d.batch, d.batch_inner, d.taps, d.dil = 1, 1, 1, 1
            d.alpha, d.act, d.accumulate, d.splitk = alpha, ACT_NONE, ACC_SOLE, 1
            d.a_dtype, d.b_dtype, d.c_dtype, d.compute, d.s_dtype = _dt(dy), _dt(x), _dt(dW), compute, F32
            d.colsum_slots = 1
        if PROFILE is not None:      # bench.py: one table row for the whole group
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = lib.a3t_gemm_tn3_group(ctypes.cast(arr, ctypes.c_void_p), len(items), _stream())
        if rc == 0:
            if PROFILE is not None:
                e1.record()
                tok, cin = items[0][0].shape[0], items[0][1].shape[1]
                cout = sum(dy.shape[1] for dy, _, _, _ in items)
                PROFILE.append((lib.a3t_gemm_last_kernel().decode() + f" x{len(items)} grouped", 2.0 * tok * cout * cin, e0, e1,
                                (cout, cin, tok, 1, 1, 1)))
            return True
        if rc != -1:
            L.check(rc, "a3t_gemm_tn3_group")
    for dy, x, dW, alpha in items:
        linear_bwd_weight(dy, x, dW, alpha=alpha, compute=compute)
    return False


# ---- Conv1d over time as implicit-im2col GEMM; weights kept as Wk[N][taps][Cin] --------------
def conv_fwd(x, Wk, out, Tseq, pad, dil=1, bias=None, R=None, alpha=1.0, act=ACT_NONE, compute=F32, drop=None,
             keep_out=None, keep_in=None, colsum=None, S=None, keep_layout=0):
    M, Cin = x.shape
    N, taps, _ = Wk.shape
    gemm(x, Wk, out, M, N, taps * Cin, Cin, 1, taps * Cin, 1, N, b_ts=Cin, bias=bias, R=R, taps=taps, pad=pad,
         dil=dil, Tseq=Tseq, alpha=alpha, act=act, compute=compute, drop=drop, keep_out=keep_out, keep_in=keep_in,
         colsum=colsum, S=S, keep_layout=keep_layout)


G8_BIAS_ACT, G8_DROP, G8_KEEP_OUT, G8_KEEP_IN, G8_F32_OR_RES, G8_COLSUM, G8_SMASK = 1, 2, 4, 8, 16, 32, 64


def gemm_8p_supported(M, N, K, taps=1, flags=0):
    """True when a3t_gemm runs this k-contiguous bf16 problem, with the epilogue `flags`, on the persistent 8-phase kernel
    (csrc/gemm_bf16_8p.hip)."""
    return bool(L.load().a3t_gemm_8p_supported(int(M), int(N), int(K), int(taps), int(flags)))


def gemm_pn_supported(M, N, K, taps=1, flags=0):
    """True when a3t_gemm runs this k-contiguous bf16 problem on the 384-column panel kernel (csrc/gemm_bf16_pn.hip)."""
    return bool(L.load().a3t_gemm_pn_supported(int(M), int(N), int(K), int(taps), int(flags)))


def gemm_tt_supported(M, N, K, batch):
    """True when a3t_gemm runs the batched score-sized product on the streaming kernel (csrc/gemm_bf16_tt.hip) -- the precondition
    of gemm(second=...)."""
    return bool(L.load().a3t_gemm_tt_supported(int(M), int(N), int(K), int(batch)))


def gemm_mode_tag():
    """(8-phase mode, panel mode) the dispatcher's cost models currently run under -- part of every cached GEMM plan key."""
    lib = L.load()
    m8 = lib.a3t_gemm_8p_mode(2)
    lib.a3t_gemm_8p_mode(m8)
    mp = lib.a3t_gemm_pn_mode(2)
    lib.a3t_gemm_pn_mode(mp)
    return (m8, mp)


def collate_paint(fs, fe, alen, sel, mspan, nms, flen, tlen, masked, speech_mask, text_mask, sp, tp, sega_emb):
    B, P = fs.shape
    Tm, Tp, S = masked.shape[1], text_mask.shape[1], mspan.shape[1]
    L.check(L.load().a3t_collate_paint(_ptr(fs), _ptr(fe), _ptr(alen), _ptr(sel), _ptr(mspan), _ptr(nms), _ptr(flen), _ptr(tlen),
                                       _ptr(masked), _ptr(speech_mask), _ptr(text_mask), _ptr(sp), _ptr(tp), B, Tm, Tp, P, S,
                                       int(bool(sega_emb)), _stream()), "collate_paint")


def segment_colsum(x, out, B, T):
    L.check(L.load().a3t_segment_colsum(_ptr(x), _ptr(out), B, T, x.shape[1], _stream()), "segment_colsum")


def gemm_keep_bytes(M, N):
    return int(L.load().a3t_gemm_keep_bytes(int(M), int(N)))


def conv_bwd_data(dy, Wk, dx, Tseq, pad, dil=1, S=None, alpha=1.0, compute=F32, colsum=None):
    """dx[m][c] = sum_tap sum_n dy[m-(tap-pad)*dil][n] Wk[n][tap][c]: the same im2col loader on dy
    with the taps read back to front (pad' = taps-1-pad) and B addressed as W^T via strides."""
    M, N = dy.shape
    _, taps, Cin = Wk.shape
    Wv = Wk.view(-1)[(taps - 1) * Cin:]
    gemm(dy, Wv, dx, M, Cin, taps * N, N, 1, 1, taps * Cin, Cin, b_ts=-Cin, S=S, taps=taps, pad=taps - 1 - pad,
         dil=dil, Tseq=Tseq, alpha=alpha, compute=compute, colsum=colsum)


def conv_bwd_weight(dy, x, dWk, Tseq, pad, dil=1, alpha=1.0, compute=F32):
    """dWk[n][tap][c] += alpha * sum_m dy[m][n] x[m+(tap-pad)*dil][c]  (one token-shifted TN GEMM per tap)"""
    M, N = dy.shape
    Cin = x.shape[1]
    taps = dWk.shape[1]
    if compute == BF16 and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and Cin % 128 == 0 \
            and N % 8 == 0 and taps > 1:
        # one launch: output columns (tap, c); every 128-column tile carries its own token shift
        tiles = ((N + 127) // 128) * (taps * Cin // 128)
        gemm(dy, x, dWk, N, taps * Cin, M, 1, N, 1, Cin, taps * Cin, taps=taps, pad=pad, dil=dil, Tseq=Tseq,
             alpha=alpha, acc=ACC_SOLE, splitk=_splitk_for(tiles, M), compute=compute)
        return
    tiles = ((N + 127) // 128) * ((Cin + 127) // 128)
    sk = _splitk_for(tiles, M)
    flat = dWk.view(-1)
    for tap in range(taps):
        gemm(dy, x, flat[tap * Cin:], N, Cin, M, 1, N, 1, Cin, taps * Cin, Tseq=Tseq, kshift=(tap - pad) * dil,
             alpha=alpha, acc=ACC_ATOMIC, splitk=sk, compute=compute)


# ---- row / column kernels --------------------------------------------------------------------
def _dto(t):
    return _dt(t) if t is not None else F32


def layernorm_fwd(x, g, b, y, mean, rstd, eps):
    M, D = x.shape
    L.check(L.load().a3t_layernorm_fwd(_ptr(x), _ptr(g), _ptr(b), _ptr(y), _dt(y), _ptr(mean), _ptr(rstd), M, D, eps,
                                       _stream()), "ln_fwd")


def layernorm_bwd(dy, x, g, mean, rstd, dres, dx, dg, db, dx16=None, dxsum=None, dxsum_scale=1.0, drop=(0.0, 0)):
    """drop=(p, key): dx16 / dxsum carry the dropout-masked gradient of the consumer sub-layer's branch."""
    M, D = x.shape
    L.check(L.load().a3t_layernorm_bwd(_ptr(dy), _dt(dy), _ptr(x), _ptr(g), _ptr(mean), _ptr(rstd), _ptr(dres),
                                       _ptr(dx), _ptr(dx16), _ptr(dg), _ptr(db), _ptr(dxsum), dxsum_scale, M, D,
                                       drop[0], drop[1], _stream()), "ln_bwd")


def col_reduce(x, out0, out1=None, y=None, rowmask=None, mode=0, ld=None):
    M, C = x.shape
    L.check(L.load().a3t_col_reduce(_ptr(x), _dt(x), _ptr(y), _ptr(rowmask), _ptr(out0), _ptr(out1), M, C,
                                    ld if ld is not None else x.stride(0), mode, _stream()), "col_reduce")


def f64_to_f32_add(src, dst, scale=1.0):
    L.check(L.load().a3t_f64_to_f32_add(_ptr(src), _ptr(dst), dst.numel(), scale, _stream()), "f64_to_f32_add")


def bias_grad(dy, dbias, scratch64, scale=1.0):
    """dbias += scale * colsum(dy) via the double-precision column reducer."""
    n = dy.shape[1]
    s = scratch64[:n]
    s.zero_()
    col_reduce(dy, s, mode=0)
    f64_to_f32_add(s, dbias, scale)


def bn_act_fwd(z, stats, g, b, rmean, rvar, mean_out, rstd_out, y, eps, momentum, training, act):
    M, C = z.shape
    L.check(L.load().a3t_bn_act_fwd(_ptr(z), _ptr(stats), _ptr(g), _ptr(b), _ptr(rmean), _ptr(rvar), _ptr(mean_out),
                                    _ptr(rstd_out), _ptr(y), _dt(y), M, C, eps, momentum, int(training), act,
                                    _stream()), "bn_act_fwd")


def bn_act_bwd(dy, z, mean, rstd, g, b, sums, dz, dg, db, training, act, zero=True):
    M, C = z.shape
    lib = L.load()
    if zero:
        sums.zero_()
    L.check(lib.a3t_bn_act_bwd_a(_ptr(dy), _dt(dy), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(g), _ptr(b), _ptr(sums), M, C,
                                 act, _stream()), "bn_bwd_a")
    L.check(lib.a3t_bn_act_bwd_b(_ptr(dy), _dt(dy), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(g), _ptr(b), _ptr(sums),
                                 _ptr(dz), _ptr(dg), _ptr(db), M, C, int(training), act, _stream()), "bn_bwd_b")


def glu_dwconv_fwd(g, wdw, bdw, glu, z, Tseq):
    M, C = glu.shape
    L.check(L.load().a3t_glu_dwconv_fwd(_ptr(g), _dt(g), _ptr(wdw), _ptr(bdw), _ptr(glu), _dt(glu), _ptr(z), M, C,
                                        wdw.shape[1], Tseq, _stream()), "glu_dwconv_fwd")


def glu_dwconv_bwd(dz, g, glu, wdw, dg, dwdw, dbdw, Tseq, dgsum=None):
    M, C = glu.shape
    L.check(L.load().a3t_glu_dwconv_bwd(_ptr(dz), _ptr(g), _dt(g), _ptr(glu), _dt(glu), _ptr(wdw), _ptr(dg), _dt(dg),
                                        _ptr(dwdw), _ptr(dbdw), _ptr(dgsum), M, C, wdw.shape[1], Tseq, _stream()),
            "glu_dwconv_bwd")


def add_pos_bias(qkv, u, v, qu, qv):
    M, d = qu.shape
    L.check(L.load().a3t_add_pos_bias(_ptr(qkv), _ptr(u), _ptr(v), _ptr(qu), _ptr(qv), _dt(qkv), M, d, _stream()),
            "pos_bias")


def add_pos_bias_bwd(dqu, dqv, dqkv):
    M, d = dqu.shape
    L.check(L.load().a3t_add_pos_bias_bwd(_ptr(dqu), _ptr(dqv), _ptr(dqkv), _dt(dqu), M, d, _stream()),
            "pos_bias_bwd")


def relpos_softmax_fwd(ac, bd, keymask, probs, B, H, T, scale, probs_drop=None, drop=(0.0, 0)):
    L.check(L.load().a3t_relpos_softmax_fwd(_ptr(ac), _ptr(bd), _dt(ac), _ptr(keymask), _ptr(probs), _dt(probs), B, H, T,
                                            T * T, T * T, T * T, scale, _ptr(probs_drop), drop[0], drop[1],
                                            _stream()), "softmax_fwd")


def relpos_softmax_bwd(probs, dprobs, ds, dbd, B, H, T, scale, probs_drop=None, drop_p=0.0, dbd_head_major=False,
                       drop_key=0, rowscale=None):
    """ds/dbd share a dtype; ds may be dprobs itself (fp32 in place).  dbd_head_major: dbd is laid out [H][B][T][T].
    drop_p > 0 with probs_drop=None: the mask is regenerated from (drop_key, index) instead of read off probs_drop."""
    bsb, bsh = (T * T, B * T * T) if dbd_head_major else (0, 0)
    L.check(L.load().a3t_relpos_softmax_bwd(_ptr(probs), _dt(probs), _ptr(dprobs), _dt(dprobs), _ptr(ds), _ptr(dbd), _dt(dbd), B, H,
                                            T, T * T, T * T, T * T, scale, _ptr(probs_drop), drop_p, bsb, bsh, drop_key,
                                            _ptr(rowscale), _stream()), "softmax_bwd")


def attn_fused_supported(dk, T):
    """Mirror of attn_shape_ok (csrc/attn_fused.hip): shapes outside fall back to the materialised attention path."""
    return dk % 32 == 0 and dk <= 192 and dk != 160 and T % 8 == 0 and 8 <= T <= 4096


def _attn_q_operands(qu, qv, qkv, pos_bias):
    """(qu pointer, qv pointer, ldq, bias_u pointer, bias_v pointer): with pos_bias = (pos_bias_u, pos_bias_v) the kernel reads q
    itself (first d columns of qkv) and adds the biases as it loads its query fragments; qu / qv are then not needed."""
    d = qkv.shape[1] // 3
    if pos_bias is None:
        return _ptr(qu), _ptr(qv), d, None, None
    return _ptr(qkv), _ptr(qkv), 3 * d, _ptr(pos_bias[0]), _ptr(pos_bias[1])


def attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, scale, drop=(0.0, 0), pos_bias=None):
    """Fused legacy rel-pos attention forward: ctx[b, :, h, :] = dropout(softmax(((q+u) k^T + shift((q+v) P^T)) * scale)) v.
    qu / qv / ctx [B*T][d], qkv [B*T][3d] (q | k | v), P [T][d], all bf16; lse [B][H][T] fp32.
    pos_bias=(u, v) (fp32 [d]): qu / qv may be None, q + u / q + v are formed inside the kernel."""
    d = qkv.shape[1] // 3
    dk = d // H
    kk = qkv.view(-1)[d:]
    vv = qkv.view(-1)[2 * d:]
    pqu, pqv, ldq, pbu, pbv = _attn_q_operands(qu, qv, qkv, pos_bias)
    L.check(L.load().a3t_attn_fwd(pqu, pqv, _ptr(kk), _ptr(vv), _ptr(P), _ptr(keymask), _ptr(ctx), _ptr(lse),
                                  B, H, T, dk, ldq, 3 * d, d, d, scale, drop[0], drop[1], pbu, pbv, _stream()), "attn_fwd")


def attn_fwd_train(qu, qv, qkv, P, keymask, ctx, lse, probs, probs_drop, rowscale, B, H, T, scale, drop=(0.0, 0), pos_bias=None):
    """a3t_attn_fwd for training steps: also stores un-normalised probabilities (probs, probs_drop [B][H][T][T] bf16) and
    rowscale [B][H][T] = 1 / row sum for the materialised backward.  probs_drop=None with dropout on: ONE tensor, the mask in the
    sign bits of probs (attn_bwd_ds(signed_probs=True), gemm(a_signmask=True, alpha=1/(1-p)) read it)."""
    d = qkv.shape[1] // 3
    dk = d // H
    kk = qkv.view(-1)[d:]
    vv = qkv.view(-1)[2 * d:]
    pqu, pqv, ldq, pbu, pbv = _attn_q_operands(qu, qv, qkv, pos_bias)
    e0 = None
    if PROFILE is not None:      # bench.py: the fused attention forward sits in the per-kernel table next to the GEMMs
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.load().a3t_attn_fwd_train(pqu, pqv, _ptr(kk), _ptr(vv), _ptr(P), _ptr(keymask), _ptr(ctx), _ptr(lse),
                                        _ptr(probs), _ptr(probs_drop), _ptr(rowscale), B, H, T, dk, ldq, 3 * d, d, d, scale,
                                        drop[0], drop[1], pbu, pbv, _stream()), "attn_fwd_train")
    if e0 is not None:
        e1.record()
        PROFILE.append((f"attn_fwd32_kernel<{dk // 32}, {'true' if drop[0] > 0 else 'false'}, true, false, false, {'true' if (drop[0] > 0 and probs_drop is None) else 'false'}>", 3 * 2.0 * B * H * T * T * dk, e0, e1,
                        (T, T, dk, B * H, 1, 1)))


def attn_scale_rows(x, rowscale, y, B, H, T):
    d = x.shape[1]
    L.check(L.load().a3t_attn_scale_rows(_ptr(x), _ptr(rowscale), _ptr(y), B, H, T, d // H, _stream()), "attn_scale_rows")


def attn_bwd_ds(dctx, ctx, qkv, probs, rowscale, ds, dbd, B, H, T, scale, drop=(0.0, 0), dbd_head_major=False, signed_probs=False,
                ds_bs=0):
    """dS and the compact dBD from the saved un-normalised probabilities of attn_fwd_train (dP = dctx V^T is never stored; the row
    term delta = dctx . ctx is formed in the kernel): replaces the dprobs GEMM + relpos_softmax_bwd.  qkv: [B*T, 3d], V in
    columns 2d..3d; ctx: the forward's output [B*T, d].  signed_probs: probs is the sign-tagged single tensor of
    attn_fwd_train(probs_drop=None) -- the dropout mask is read off its sign bits.  dbd=None: only dS is written, into (b, h) blocks
    ds_bs elements apart (T zeros in front of each): the compact dBD matrix is then the view ds.view(-1)[-(T-1):] with row stride T + 1."""
    d = dctx.shape[1]
    v = qkv.view(-1)[2 * d:]
    bsb, bsh = ((T * T, B * T * T) if dbd_head_major else (0, 0))
    L.check(L.load().a3t_attn_bwd_ds(_ptr(dctx), _ptr(ctx), _ptr(v), _ptr(probs), _ptr(rowscale), _ptr(ds), _ptr(dbd), B, H, T,
                                     d // H, d, 3 * d, bsb, bsh, scale, drop[0], drop[1], 1 if signed_probs else 0, ds_bs, _stream()), "attn_bwd_ds")


def mask_fill(speech, masked, mask_feature, out):
    M, C = out.shape
    L.check(L.load().a3t_mask_fill(_ptr(speech), _ptr(masked), _ptr(mask_feature), _ptr(out), _dt(out), M, C,
                                   _stream()), "mask_fill")


def embed_finish_fwd(e, emb, seg, text, spos, tpos, xs, B, Tm, Tp, D, xscale, drop=(0.0, 0), spk=None):
    L.check(L.load().a3t_embed_finish_fwd(_ptr(e), _ptr(emb), _ptr(seg), _ptr(text), _ptr(spos), _ptr(tpos), _ptr(xs),
                                          B, Tm, Tp, D, xscale, drop[0], drop[1], _ptr(spk), _stream()), "embed_fwd")


def embed_finish_bwd(dxs, e, text, spos, tpos, de, demb, dseg, B, Tm, Tp, D, V, nseg, xscale, drop=(0.0, 0)):
    L.check(L.load().a3t_embed_finish_bwd(_ptr(dxs), _ptr(e), _ptr(text), _ptr(spos), _ptr(tpos), _ptr(de),
                                          _ptr(demb), _ptr(dseg), B, Tm, Tp, D, V, nseg, xscale, drop[0], drop[1],
                                          _stream()), "embed_bwd")


def scale(x, y, s):
    L.check(L.load().a3t_scale(_ptr(x), _ptr(y), x.numel(), s, _stream()), "scale")


def scale_dev(x, y, s):
    L.check(L.load().a3t_scale_dev(_ptr(x), _ptr(y), x.numel(), _ptr(s), _stream()), "scale_dev")


def axpy(x, y, a=1.0):
    L.check(L.load().a3t_axpy(_ptr(x), _ptr(y), x.numel(), a, _stream()), "axpy")


def attn_bias_fold(slots, S, d, gu, gv, gbqkv):
    L.check(L.load().a3t_attn_bias_fold(_ptr(slots), S, d, _ptr(gu), _ptr(gv), _ptr(gbqkv), _stream()), "attn_bias_fold")


def cast_bf16(x, y):
    L.check(L.load().a3t_cast_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), "cast_bf16")


def cast_bf16_conv_t(src_flat, dst, src_off, dst_off, N, taps, C):
    """Transposed, tap-reversed bf16 shadows of src_off.numel() conv weights [N][taps][C] of the flat fp32 buffer."""
    L.check(L.load().a3t_cast_bf16_conv_t(_ptr(src_flat), _ptr(dst), _ptr(src_off), _ptr(dst_off), src_off.numel(), N, taps, C,
                                          _stream()), "cast_bf16_conv_t")


def split_bf16(x, hi, lo):
    L.check(L.load().a3t_split_bf16(_ptr(x), _ptr(hi), _ptr(lo), x.numel(), _stream()), "split_bf16")


def slice_rows(x, y, B, T, Tm, D, reverse_add=False):
    L.check(L.load().a3t_slice_rows(_ptr(x), _ptr(y), _dt(y), B, T, Tm, D, int(reverse_add), _stream()), "slice_rows")


def reflect_pad(x, out, pad):
    B, N = x.shape
    L.check(L.load().a3t_reflect_pad(_ptr(x), _ptr(out), B, N, pad, out.shape[1], _stream()), "reflect_pad")


def stft_amp(S, amp, nbins):
    L.check(L.load().a3t_stft_amp(_ptr(S), _ptr(amp), amp.shape[0], nbins, amp.shape[1], _stream()), "stft_amp")


def logmel_finish(mel, olens, B, F, C):
    L.check(L.load().a3t_logmel_finish(_ptr(mel), _ptr(olens), B, F, C, _stream()), "logmel_finish")


def mlm_loss(before, after, target, masked, loss_out, d_before, d_after, scratch, l2=False, gscale=1.0):
    M, C = before.shape
    L.check(L.load().a3t_mlm_loss(_ptr(before), _ptr(after), _ptr(target), _ptr(masked), _ptr(loss_out),
                                  _ptr(d_before), _ptr(d_after), _ptr(scratch), M, C, int(l2), gscale, _stream()),
            "mlm_loss")


def loss_scratch_floats(M):
    return L.load().a3t_mlm_loss_scratch_floats(M)


def sumsq(g, partial):
    L.check(L.load().a3t_sumsq(_ptr(g), g.numel(), _ptr(partial), _stream()), "sumsq")


def clip_adam(p, g, m, v, partial, norm_out, lr, step, clip=1.0, gscale=1.0, betas=(0.9, 0.999), eps=1e-8):
    L.check(L.load().a3t_clip_adam(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(partial), _ptr(norm_out), p.numel(), lr,
                                   betas[0], betas[1], eps, step, clip, gscale, _stream()), "clip_adam")


def clip_adam_noam(p, g, m, v, partial, norm_out, state, base_lr, model_size, warmup, clip=1.0, gscale=1.0,
                   betas=(0.9, 0.999), eps=1e-8):
    """clip + Adam + NoamLR with the step counter on the device (state int32[2]: applied, skipped)."""
    L.check(L.load().a3t_clip_adam_noam(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(partial), _ptr(norm_out), p.numel(),
                                        _ptr(state), base_lr, float(model_size), float(warmup), betas[0], betas[1], eps,
                                        clip, gscale, _stream()), "clip_adam_noam")


def pwg_gate(y, c, out):
    T, H = out.shape
    L.check(L.load().a3t_pwg_gate(_ptr(y), _ptr(c), _ptr(out), T, H, _stream()), "pwg_gate")


def pwg_block(x, cu, wt0, b0, wt1, b1, g, skips, B, Tw, dil):
    """Fused residual block (a3t_pwg_block): x, skips updated in place."""
    L.check(L.load().a3t_pwg_block(_ptr(x), _ptr(cu), _ptr(wt0), _ptr(b0), _ptr(wt1), _ptr(b1), _ptr(g), _ptr(skips),
                                   B, Tw, dil, _stream()), "pwg_block")


def pwg_res_skip(o, x, skips):
    T, R = x.shape
    L.check(L.load().a3t_pwg_res_skip(_ptr(o), _ptr(x), _ptr(skips), T, R, skips.shape[1], _stream()), "pwg_res_skip")


def pwg_upsample(c, w, out, scale_):
    """c [Tin][C] or [B][Tin][C] (contiguous) -> out [.., Tin*scale, C]"""
    B = c.shape[0] if c.dim() == 3 else 1
    Tin, C = c.shape[-2:]
    L.check(L.load().a3t_pwg_upsample(_ptr(c), _ptr(w), _ptr(out), B, Tin, C, scale_, _stream()), "pwg_upsample")


def replicate_pad(x, y, pad):
    """x [T][C] or [B][T][C] (contiguous) -> y [.., T + 2 pad, C]"""
    B = x.shape[0] if x.dim() == 3 else 1
    T, C = x.shape[-2:]
    L.check(L.load().a3t_replicate_pad(_ptr(x), _ptr(y), B, T, C, pad, _stream()), "replicate_pad")


def bias_act(x, bias, act, scale_=1.0):
    M, C = x.shape
    L.check(L.load().a3t_bias_act(_ptr(x), _ptr(bias), M, C, act, scale_, _stream()), "bias_act")


def dropout(x, y, p, key, scale=1.0):
    L.check(L.load().a3t_dropout(_ptr(x), _dt(x), _ptr(y), _dt(y), x.numel(), p, key, scale, _stream()), "dropout")


def dropout_bwd_cast(g, gm, p, key, colsum=None, colsum_scale=1.0):
    M, C = g.shape
    L.check(L.load().a3t_dropout_bwd_cast(_ptr(g), _ptr(gm), _dt(gm), _ptr(colsum), colsum_scale, M, C, p, key,
                                          _stream()), "dropout_bwd_cast")
