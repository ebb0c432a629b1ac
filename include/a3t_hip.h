/* a3t_hip.h -- C ABI of liba3t_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for
 * the A3T masked-mel training step and the ParallelWaveGAN inference path.
 *
 * The reference (richardbaihe/a3t) has no FFI for this path: every device op is a PyTorch
 * ATen call made from Python nn.Modules.  The boundary this library
 * replaces is therefore the ATen op sequence D1-D23 / V1-V6; each entry point below cites the
 * reference module whose forward (and autograd backward) it implements.  The Python host
 * (a3t_amd/) binds these symbols with ctypes -- see INTEGRATION.md for the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted.
 *   - `stream` is a hipStream_t passed as void*.  Launches are stream-ordered and do not sync the host.  The caller owns
 *     every buffer; the library keeps exactly three pieces of state of its own, all grow-only and allocated on first use
 *     (hipMalloc during warm-up, never in steady state): the kernel-selection modes (a3t_gemm_*_mode, a3t_attn_split_mode:
 *     process-wide switches for tests / A-B runs), the per-DEVICE key-split workspace + overflow flags of the fused attention
 *     forward (stream contract at a3t_attn_fwd), and the per-(device, STREAM) split-K partial slab of the weight-gradient
 *     kernel (a launch and its fold are ordered on their stream; different streams never share a slab).
 *   - every function returns 0 on success or a hipError_t / negative A3T_E* code.
 *   - activations are row-major (rows = tokens m = b*T + t, cols = channels), fp32 unless a
 *     dtype field says otherwise.
 */
#ifndef A3T_HIP_H
#define A3T_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A3T_F32 0
#define A3T_BF16 1

#define A3T_EINVAL (-22)

#define A3T_ACT_NONE 0
#define A3T_ACT_RELU 1
#define A3T_ACT_TANH 2
#define A3T_ACT_SWISH 3

#define A3T_ACC_STORE 0  /* C  = v           */
#define A3T_ACC_ADD 1    /* C += v           */
#define A3T_ACC_ATOMIC 2 /* atomicAdd(C, v)  */
#define A3T_ACC_SOLE 3   /* C += v, K may be split; the caller guarantees that nothing else writes C while the launch runs (a
                            parameter gradient has one writer per backward pass): kernels that fold split-K partials add them
                            with plain read-modify-writes, kernels that accumulate in place use atomics as for A3T_ACC_ATOMIC */

/* One descriptor drives every dense contraction on the path (torch.nn.Linear, Conv1d as
 * implicit-im2col GEMM, the attention bmm's and all their weight/data gradients):
 *
 *   C[z][m][n] (op)= alpha * relu_mask( act( sum_k A(m,k) * B(n,k) + bias[n] ) ) + R[m][n]
 *
 * A(m,k):  plain  : A[m*a_rs + k*a_cs]
 *          conv   : taps>1, k=(tap,c) with c in [0,K/taps): row m' = m + (tap-pad)*dil, zero when
 *                   (m % Tseq) + (tap-pad)*dil is outside [0,Tseq)  (Conv1d zero padding per
 *                   utterance; reference: transformer/multi_layer_conv.py:36-63,
 *                   tacotron2/decoder.py:165-267, wavenet/residual_block.py:82-96)
 * B(n,k):  B[n*b_rs + tap*b_ts + c*b_cs]          (k=(tap,c), taps as above; taps==1: k*b_cs)
 *          with kshift/Tseq (token-reduction GEMMs, taps==1): row k' = k + kshift, zero when
 *          (k % Tseq) + kshift is outside [0,Tseq)   (Conv1d weight gradient, one tap per launch)
 * One of a_rs/a_cs (and of b_rs/b_cs) must be 1.
 * Batching: z in [0,batch): z0 = z / batch_inner, z1 = z % batch_inner; X += z0*x_bs0 + z1*x_bs1.
 * splitk>1 splits K over extra workgroups; requires accumulate == A3T_ACC_ATOMIC or A3T_ACC_SOLE.
 * ZERO-INITIALISE the descriptor (`a3t_gemm_desc d = {0};`): optional fields are appended as the library grows (round 6: a_signmask,
 * keep_layout, A2 .. a_unaligned) and zero / NULL always means "off".
 */
typedef struct a3t_gemm_desc {
    const void* A;
    const void* B;
    void* C;
    const float* bias;  /* [N] or NULL */
    const float* R;     /* residual, same layout/strides as C, or NULL */
    const float* S;     /* relu-mask source (keep where S>0), same layout as C, or NULL */
    int32_t M, N, K;
    int64_t a_rs, a_cs;
    int64_t b_rs, b_cs, b_ts;
    int64_t c_rs;
    int32_t batch, batch_inner;
    int64_t a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
    int32_t taps, pad, dil, Tseq, kshift;
    float alpha;
    int32_t act;
    int32_t accumulate;
    int32_t splitk;
    int32_t a_dtype, b_dtype, c_dtype; /* A3T_F32 | A3T_BF16 (storage) */
    int32_t compute;                   /* A3T_F32: v_mfma_f32_32x32x2_f32 (exact f32);
                                          A3T_BF16: v_mfma_f32_32x32x16_bf16, fp32 accumulate */
    int32_t s_dtype;                   /* storage type of S (A3T_F32 | A3T_BF16) */
    int32_t colsum_slots;              /* 0/1: one accumulator row; S > 1: output tiles spread their atomics over S
                                          copies colsum + slot*colsum_ss (slot = (row tile + batch) % S) which the
                                          caller folds afterwards -- for single-round GEMMs whose tiles all finish
                                          together (hundreds of same-address atomics in one burst cost 40 us) */
    float* colsum;                     /* optional: colsum[z1*colsum_bs1 + n] += colsum_scale * sum_m C[m][n]
                                          (bias gradients fused into the data-gradient GEMM; bf16 path only) */
    int64_t colsum_bs1;
    float colsum_scale;
    uint32_t drop_key;                 /* dropout fused into the epilogue (after act / mask, before alpha): */
    float drop_p;                      /*   v = keep(drop_key, linear index in C) ? v/(1-p) : 0 ; p = 0 disables */
    int32_t colsum_ss;                 /* slot stride (floats) of the spread column sums */
    void* keep_out;                    /* optional (8-phase kernel only, N % 256 == 0): one bit per output = (stored value > 0), */
    const void* keep_in;               /*   in the kernel's tile-major image of a3t_gemm_keep_bytes(M, N) bytes; keep_in applies
                                            such an image as a mask in the place of S (the ReLU'/dropout mask of
                                            multi_layer_conv.py:52-63's hidden layer on its way back).  a3t_gemm fails with
                                            A3T_EINVAL when either is set and the 8-phase kernel does not take the problem:
                                            ask a3t_gemm_8p_supported first. */
    int32_t a_signmask;                /* 1: bf16 elements of A whose sign bit is set are read as zero -- the dV product of the
                                            attention backward (attention.py:64-96) over the probabilities a3t_attn_fwd_train
                                            stores with the dropout mask in their sign bits.  m-contiguous bf16 A only (the
                                            streaming kernel of csrc/gemm_bf16_tt.hip or the 128-row kernel); anything else:
                                            A3T_EINVAL */
    int32_t keep_layout;               /* layout of the keep_out / keep_in image.  0: the 8-phase kernel's tile-major bit image
                                            (above).  1: a row-major image of M * N / 4 bytes -- byte m * (N / 4) + n / 4 holds
                                            (stored value > 0) of columns n .. n + 3 in its bits 0-3: written by the 128-row
                                            kernel's vector epilogue (no residual, bf16 or fp32 C with c_rs == N, no batch, no
                                            split), read as the mask in the place of S by the 128-row kernel and by the 384-column
                                            panel kernel.  The first FFN conv of multi_layer_conv.py:52-63 hands its ReLU / dropout
                                            mask to the data gradient of the second one this way where the 8-phase kernel does
                                            not run (configs[1]): 14 MB instead of a 110-MB read of the saved activation. */
    const void* A2;                    /* optional SECOND product accumulated in the same launch: C = alpha (A B + A2 B2), one rounding. */
    const void* B2;                    /*   A2 has A's strides and batch strides, B2 [k][n] n-contiguous with row stride b2_cs and batch
                                            strides b2_bs0/1; same M, N, K.  dq = dS K + dBD P of the attention backward
                                            (attention.py:190-203 on its way back) as ONE launch of the streaming kernel
                                            (csrc/gemm_bf16_tt.hip, [m][k] operand, A3T_ACC_STORE): anything else A3T_EINVAL -- ask
                                            a3t_gemm_tt_supported first. */
    int64_t b2_cs, b2_bs0, b2_bs1;
    float* colsum2;                    /* with colsum: column sums of the second product's share (colsum takes the first's); same
                                            slots / strides as colsum */
    int64_t a2_rs;                     /* row stride of A2 (0: A's) */
    int32_t a_unaligned;               /* bit 0: A, bit 1: A2 is a strided VIEW whose base is only 2-byte aligned and whose leading
                                            stride is any number of elements (the 16-byte LDS-DMA of gfx950 takes such sources):
                                            the compact dBD matrix of the attention backward read off the dS tensor it is a shifted
                                            copy of (dbd[r][c] = ds_flat[r (T + 1) + c - (T - 1)]; a3t_attn_bwd_ds with dbd = NULL).
                                            Bit 0: the 128-row kernel, plain (taps == 1) bf16 products only; bit 1: with A2. */
} a3t_gemm_desc;

int a3t_gemm(const a3t_gemm_desc* d, void* stream);
/* 1 when a3t_gemm runs the k-contiguous bf16 problem (M, N, K = taps * channels) on the persistent 256x256 8-phase kernel
 * (csrc/gemm_bf16_8p.hip) with the epilogue `flags` (1 bias/activation, 2 dropout, 4 keep_out, 8 keep_in, 16 fp32 output or
 * residual, 32 column sums); bytes of a keep-bit image */
int a3t_gemm_8p_supported(int M, int N, int K, int taps, int flags);
int64_t a3t_gemm_keep_bytes(int M, int N);
/* 1 when a3t_gemm runs the k-contiguous bf16 problem on the 160-row x 384-column panel kernel (csrc/gemm_bf16_pn.hip: every
 * GEMM whose output is d_model = 384 wide -- the second FFN conv of multi_layer_conv.py:36-63, linear_out of
 * attention.py:64-96, pointwise_conv2 of the conformer ConvolutionModule, and their data gradients through transposed
 * weight shadows); same `flags` as above (4 and 8: never). */
int a3t_gemm_pn_supported(int M, int N, int K, int taps, int flags);
/* 1 when a3t_gemm runs the batched bf16 product (M x N per batch element over a reduction of K; A score-sized [m][k] or [k][m],
 * B [k][n]) on the streaming kernel (csrc/gemm_bf16_tt.hip) under its current mode: the precondition of a3t_gemm_desc::A2 */
int a3t_gemm_tt_supported(int M, int N, int K, int batch);

/* LayerNorm over the last dim (transformer/layer_norm.py:12-42 eps=1e-12; torch.nn.LayerNorm
 * eps=1e-5 in the speech embed, conformer/encoder.py:404).  mean/rstd: [M] saved for backward. */
int a3t_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                      float* mean, float* rstd, int M, int D, float eps, void* stream);
/* dx = (dres ? dres : 0) + LN'(dy) (dx may alias dres); dx_bf16: optional bf16 copy of dx for the
 * GEMMs that consume it; dgamma/dbeta are ACCUMULATED (atomicAdd). */
int a3t_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                      const float* rstd, const float* dres, float* dx, void* dx_bf16, float* dgamma,
                      float* dbeta, float* dx_colsum, float dx_colsum_scale, int M, int D, float drop_p,
                      uint32_t drop_key, void* stream);
/* dx_colsum (optional): += dx_colsum_scale * column sums of dx = the bias gradient of the layer whose
 * output gradient dx is.
 * drop_p > 0 (D % 128 == 0 only): dx is the gradient w.r.t. the OUTPUT of a "residual + dropout(branch)" sub-layer; the
 * branch's backward needs mask * dx / (1-p) (torch.nn.Dropout backward, same counter RNG as the forward mask): dx_bf16
 * and dx_colsum then carry the masked gradient, dx itself (the residual path) stays unmasked. */

/* Column reductions over rows of x[M][C] (row stride ld), accumulated with atomics:
 *   mode 0: out0[c] += sum x            (bias gradients)
 *   mode 1: out0 += sum x, out1 += sum x*x      (BatchNorm batch statistics)
 *   mode 2: out0 += sum x, out1 += sum x*y      (BatchNorm backward)
 * rowmask (optional, uint8 [M]): only rows with rowmask!=0 contribute. out are double[C]. */
int a3t_col_reduce(const void* x, int x_dtype, const float* y, const uint8_t* rowmask, double* out0,
                   double* out1, int M, int C, int64_t ld, int mode, void* stream);
int a3t_f64_to_f32_add(const double* src, float* dst, int n, float scale, void* stream);

/* BatchNorm1d(+activation) over channels-last [M][C] (conformer/convolution.py:74,
 * tacotron2/decoder.py:165-267).  stats = double[2][C] (sum, sumsq) from a3t_col_reduce when
 * training; running_mean/var updated in place when momentum > 0 (unbiased var).
 * saves mean/rstd (float[C]) for backward. */
int a3t_bn_act_fwd(const float* z, const double* stats, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float* mean_out, float* rstd_out, void* y,
                   int y_dtype, int M, int C, float eps, float momentum, int training, int act, void* stream);
/* step A: sums += (sum dbn, sum dbn*zhat) (double[2][C]) with dbn = dy * act'(bn) (recomputed, never stored) */
int a3t_bn_act_bwd_a(const void* dy, int dy_dtype, const float* z, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, double* sums, int M, int C, int act, void* stream);
/* step B: dz = gamma*rstd*(dbn - sum0/M - zhat*sum1/M) (training) or gamma*rstd*dbn (eval);
 * dgamma += sum1, dbeta += sum0 */
int a3t_bn_act_bwd_b(const void* dy, int dy_dtype, const float* z, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, const double* sums, float* dz, float* dgamma,
                     float* dbeta, int M, int C, int training, int act, void* stream);

/* GLU + depthwise Conv1d (conformer/convolution.py:66-72): g[M][2C] -> glu[M][C] (saved) and
 * z[m][c] = bdw[c] + sum_k wdw[c][k] * glu[m+k-(K-1)/2][c], zero padded per utterance (Tseq). */
int a3t_glu_dwconv_fwd(const void* g, int g_dtype, const float* wdw, const float* bdw, void* glu,
                       int glu_dtype, float* z, int M, int C, int K, int Tseq, void* stream);
/* dz -> dg[M][2C]; dwdw[C][K], dbdw[C] accumulated (atomics) */
int a3t_glu_dwconv_bwd(const float* dz, const void* g, int g_dtype, const void* glu, int glu_dtype,
                       const float* wdw, void* dg, int dg_dtype, float* dwdw, float* dbdw, float* dg_colsum,
                       int M, int C, int K, int Tseq, void* stream);
/* dg_colsum (optional, [2C]): += column sums of dg (bias gradient of pointwise_conv1) */

/* Attention helpers around the batched GEMMs (transformer/attention.py:167-209).
 * qkv [M][3d] (q|k|v); qu/qv [M][d] = q + pos_bias_{u,v}. */
int a3t_add_pos_bias(const void* qkv, const float* bias_u, const float* bias_v, void* qu, void* qv, int dtype,
                     int M, int d, void* stream);
/* dq = dqu + dqv written into dqkv[:, 0:d] (row stride 3d) */
int a3t_add_pos_bias_bwd(const void* dqu, const void* dqv, void* dqkv, int dtype, int M, int d, void* stream);
/* probs[z][i][j] = softmax_j( (ac[z][i][j] + shift(bd)[z][i][j]) * scale ) with key mask
 * (masked_fill(min) -> softmax -> masked_fill(0), attention.py:78-86).  bd is the COMPACT
 * (q+v)P^T matrix; the legacy rel_shift (attention.py:145-165) is applied on the fly in closed
 * form: j<=i -> bd[i][T-1-i+j], j==i+1 -> 0, j>i+1 -> bd[i+1][j-i-2].
 * z = b*H + h; keymask uint8 [B][T]; *_bs = per-z strides (elements); ac and bd share scores_dtype
 * (fp32, or bf16 in bf16 compute mode -- the usual bf16-training practice of bf16 logits, fp32 softmax math). */
int a3t_relpos_softmax_fwd(const void* ac, const void* bd, int scores_dtype, const uint8_t* keymask, void* probs,
                           int probs_dtype, int B, int H, int T, int64_t ac_bs, int64_t bd_bs, int64_t p_bs,
                           float scale, void* probs_drop, float drop_p, uint32_t drop_key, void* stream);
/* probs_drop (optional, drop_p > 0): the attention-dropout'ed probabilities fed to probs @ V
 * (attention.py:88 self.dropout(self.attn)); probs itself stays un-dropped for the backward. */
/* ds = probs * (dprobs - sum_j dprobs*probs) * scale (= gradient of ac; may alias dprobs when fp32)
 * and the same values scattered un-shifted into dbd (= gradient of the compact bd; fully
 * overwritten).  ds and dbd share out_dtype and the per-z stride o_bs. */
int a3t_relpos_softmax_bwd(const void* probs, int probs_dtype, const void* dprobs, int dprobs_dtype, void* ds,
                           void* dbd, int out_dtype, int B, int H, int T, int64_t p_bs, int64_t dp_bs, int64_t o_bs,
                           float scale, const void* probs_drop, float drop_p, int64_t dbd_bs_b, int64_t dbd_bs_h,
                           uint32_t drop_key, const float* rowscale, void* stream);
/* rowscale (optional, bf16 vector path): [B*H*T] fp32; probs then hold UN-normalised probabilities exp(s - m_ref) as
 * written by a3t_attn_fwd_train and row r of probs is multiplied by rowscale[r] (= 1 / its row sum) on load. */
/* drop_p > 0 with probs_drop == NULL (bf16 vector path): the dropout mask is regenerated from the counter RNG with drop_key
 * and the forward's element index instead of being read off the saved dropped probabilities. */
/* dbd_bs_b / dbd_bs_h: strides (elements) of the (b, h) blocks of dbd; 0 / 0 = the [B][H][T][T] layout of ds.  The
 * head-major layout [H][B][T][T] (dbd_bs_b = T*T, dbd_bs_h = B*T*T) makes the gradient of linear_pos
 * (attention.py:188: sum over the batch of dbd^T (q+v)) ONE reduction over K = B*T per head instead of B atomically
 * accumulated products. */

/* Fused legacy relative-position attention, bf16 operands (LegacyRelPositionMultiHeadedAttention.forward,
 * transformer/attention.py:167-209 incl. rel_shift :145-165 and forward_attention :64-96): one launch replaces the
 * ac / bd GEMMs, a3t_relpos_softmax_fwd and the probs @ V GEMM; no (T, T) tensor is written.
 *   qu, qv [B*T][ldq] = q + pos_bias_{u,v};  k, v [B*T][ldkv];  pos [T][ldp] = linear_pos(pos_emb);  head h occupies
 *   columns h*dk .. h*dk+dk-1 of every operand;  keymask uint8 [B][T];  ctx out [B*T][ldo] (bf16);
 *   lse out [B][H][T] fp32: log sum_j exp(scaled score) per query (+inf for a fully masked row), kept for backward.
 * drop_p / drop_key: attention dropout with the same counter RNG and element index ((b*H+h)*T+i)*T+j as
 * a3t_relpos_softmax_fwd's probs_drop.  dk in {32, 64, 96, 128, 192}, T % 8 == 0; A3T_EINVAL otherwise.
 * bias_u / bias_v (both or neither; fp32 [H*dk], 16-byte aligned): qu and qv then both point at q itself (e.g. the first d
 * columns of the fused q|k|v projection, ldq = 3d) and the kernel forms bf16(q + pos_bias_u) / bf16(q + pos_bias_v) as it loads
 * its query fragments -- bit for bit what a3t_add_pos_bias stores, without the two [B*T][d] tensors (attention.py:190-194).
 * STREAM CONTRACT (a3t_attn_fwd, a3t_attn_fwd_train, a3t_attn_split_mode): the overflow ("redo") flags and the key-split
 * partial-sum workspace are ONE set per device, and the workspace is (re)allocated with hipMalloc when a launch needs more
 * than any launch before it.  All fused-attention forward launches on a device must therefore be issued on one stream, or
 * be ordered by the caller (event / stream wait) so that no two of them are in flight at once; launches on different
 * DEVICES are independent.  Every other entry point of this header keeps no state between calls. */
int a3t_attn_fwd(const void* qu, const void* qv, const void* k, const void* v, const void* pos, const uint8_t* keymask,
                 void* ctx, float* lse, int B, int H, int T, int dk, int64_t ldq, int64_t ldkv, int64_t ldp, int64_t ldo,
                 float scale, float drop_p, uint32_t drop_key, const float* bias_u, const float* bias_v, void* stream);

/* The same forward for TRAINING steps whose backward runs on materialised probabilities (a3t_relpos_softmax_bwd + the
 * batched GEMMs): besides ctx / lse it stores, per score, probs = exp(s - m_ref) (bf16, UN-normalised: the reference
 * maximum of a row is fixed at the first key tile that holds a valid key, there is no rescaling pass), probs_drop = the
 * same after attention dropout (x 1/(1-p); NULL when drop_p == 0) and rowscale [B][H][T] = 1 / sum_j probs, the factor
 * a consumer applies per row.  No logits (ac, bd) and no second softmax kernel: one launch replaces two GEMMs, the
 * softmax and probs @ V of attention.py:190-209, 78-96.  probs / probs_drop: [B][H][T][T], 8-byte aligned, T % 8 == 0.
 * drop_p > 0 with probs_drop == NULL: ONE saved tensor -- probs holds exp(s - m_ref) with the SIGN BIT set on the elements the
 * dropout mask dropped (value |x|, mask = sign; a dropped zero is -0).  Its readers: a3t_attn_bwd_ds(signed_probs = 1) and the
 * dV product through a3t_gemm_desc::a_signmask with alpha = 1 / (1 - drop_p).  The dropped copy (one T x T write and one read per
 * head and layer) does not exist in that mode. */
int a3t_attn_fwd_train(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                       const uint8_t* keymask, void* ctx, float* lse, void* probs, void* probs_drop, float* rowscale,
                       int B, int H, int T, int dk, int64_t ldq, int64_t ldkv, int64_t ldp, int64_t ldo, float scale,
                       float drop_p, uint32_t drop_key, const float* bias_u, const float* bias_v, void* stream);
/* y[r][h*dk + c] = x[r][h*dk + c] * rowscale[(b*H + h)*T + i], r = b*T + i: folds the row normalisation of
 * a3t_attn_fwd_train's probabilities into the dctx operand of dV = probs_drop^T dctx (attention.py:96 backward). */
int a3t_attn_scale_rows(const void* x, const float* rowscale, void* y, int B, int H, int T, int dk, void* stream);
/* Score gradients from a3t_attn_fwd_train's saved probabilities in ONE launch (the backward of attention.py:64-96, 145-209
 * between dctx and the two T x T operands of the remaining GEMMs): dP = dctx V^T is formed tile by tile on the matrix cores and
 * never stored;  ds[b][h][i][j] = probs * rowscale[i] * (keep_ij / (1 - drop_p) * dP_ij - delta[i]) * scale  (keep from the counter
 * RNG with the forward's key and index; delta[i] = dctx_i . ctx_i = sum_j dP_ij P_ij, the row term of the softmax backward,
 * attention.py:86, formed in the kernel from the forward's output ctx, row stride ldo);  dbd = the same values in the compact dBD layout of
 * a3t_relpos_softmax_bwd (block (b, h) at b*dbd_bsb + h*dbd_bsh elements, both 0 = [B][H][T][T]; every entry is written).
 * Replaces the dprobs GEMM + a3t_relpos_softmax_bwd of the materialised backward.  dctx row stride ldo, v row stride ldkv
 * (head h at column h*dk), dk % 32 == 0 (<= 192, not 160), T % 8 == 0.
 * dbd == NULL: the compact matrix is not written -- it is the flat dS sequence shifted by T - 1 elements, and a consumer reads it as
 * a view of ds (a3t_gemm_desc::a_unaligned: base ds - (T - 1), row stride T + 1); ds_bs = elements between the (b, h) blocks of ds
 * (0: T * T; for the view T zeros in front of every block, i.e. ds_bs >= T * T + T with ds pointing behind the first block's zeros).
 * signed_probs != 0 (drop_p > 0): probs is the sign-tagged single tensor of a3t_attn_fwd_train -- the probability is |x|, keep_ij
 * is read off the sign bit instead of being regenerated (drop_key unused); signed_probs == 0 also accepts such a tensor (|x| is
 * taken either way) as long as drop_key is the forward's. */
int a3t_attn_bwd_ds(const void* dctx, const void* ctx, const void* v, const void* probs, const float* rowscale, void* ds,
                    void* dbd, int B, int H, int T, int dk, int64_t ldo, int64_t ldkv, int64_t dbd_bsb, int64_t dbd_bsh,
                    float scale, float drop_p, uint32_t drop_key, int signed_probs, int64_t ds_bs, void* stream);

/* Encoder prologue (conformer/encoder.py:522-553, mlm_encoder.py:57-70). */
int a3t_mask_fill(const float* speech, const uint8_t* masked, const float* mask_feature, void* out,
                  int out_dtype, int M, int C, void* stream);
/* xs[b][t] (t<Tm): relu(e[b*Tm+t]) * xscale + seg[spos];  (t>=Tm): emb[text]*xscale + seg[tpos];
 * spk (optional, [B][D]): projected speaker embedding added to every token of utterance b (x-vector conditioning of
 * BASELINE configs[3]; no reference behaviour: sedit_model.py:246 ignores spembs) */
int a3t_embed_finish_fwd(const float* e, const float* emb, const float* seg, const int64_t* text,
                         const int64_t* spos, const int64_t* tpos, float* xs, int B, int Tm, int Tp, int D,
                         float xscale, float drop_p, uint32_t drop_key, const float* spk, void* stream);
int a3t_embed_finish_bwd(const float* dxs, const float* e, const int64_t* text, const int64_t* spos,
                         const int64_t* tpos, float* de, float* demb, float* dseg, int B, int Tm, int Tp,
                         int D, int V, int nseg, float xscale, float drop_p, uint32_t drop_key, void* stream);
/* y = x * s (decoder entry xscale, conformer/encoder.py:585-588) */
int a3t_scale(const float* x, float* y, int64_t n, float s, void* stream);
int a3t_axpy(const float* x, float* y, int64_t n, float a, void* stream); /* y += a*x */
/* Fold of the slot-spread column sums (a3t_gemm_desc::colsum_slots) written by the four attention data-gradient GEMMs:
 * slots[S][4*d] = per slot (colsum d(q+u) | colsum d(q+v) | colsum dK | colsum dV); adds them into the gradients of
 * pos_bias_u, pos_bias_v (attention.py:137-140) and of the fused q/k/v bias [3*d] (linear_q/k/v.bias, :63-65). */
int a3t_attn_bias_fold(const float* slots, int S, int d, float* gu, float* gv, float* gbqkv, void* stream);
int a3t_scale_dev(const float* x, float* y, int64_t n, const float* s, void* stream); /* y = x * s[0], s on device */
/* copy rows [b][0:Tm] of x[B][T][D] into y[B][Tm][D] (sedit_model.py:363) and the reverse scatter-add */
int a3t_slice_rows(const float* x, void* y, int y_dtype, int B, int T, int Tm, int D, int reverse_add,
                   void* stream);
/* fp32 -> bf16 (round to nearest even); n % 4 == 0 (the flat parameter buffer once per step) */
int a3t_cast_bf16(const float* x, void* y, int64_t n, void* stream);
/* Transposed bf16 shadows of `count` Conv1d weights W[N][taps][C] (multi_layer_conv.py:36-50) living at element offsets
 * src_off[i] of the flat fp32 buffer `src`: dst[dst_off[i] + (c*taps + taps-1-t)*N + n] = bf16(W[n][t][c]) -- the B operand of
 * the conv's data gradient written as a forward conv of dy (k-contiguous, pad' = taps-1-pad).  Offsets live on the device. */
int a3t_cast_bf16_conv_t(const float* src, void* dst, const int64_t* src_off, const int64_t* dst_off, int count, int N,
                         int taps, int C, void* stream);
/* hi = bf16(x), lo = bf16(x - hi): a pair of bf16 GEMM operands that carries an fp32 tensor to ~2^-17 relative (the first
 * postnet conv reads the log-mel-scale `before`, tacotron2/decoder.py:165-267, whose bf16 ulp would otherwise be amplified
 * by five BatchNorm layers); n % 4 == 0 */
int a3t_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream);

/* Log-mel front end on the device (espnet2/layers/stft.py:56-124, log_mel.py:56-83,
 * tts/feats_extract/log_mel_fbank.py:88-106).  The STFT is a GEMM of overlapping frames
 * (A row stride = hop) with the windowed DFT basis; these are the element-wise stages around it. */
int a3t_reflect_pad(const float* x, float* out, int B, int N, int pad, int ld, void* stream);
int a3t_stft_amp(const float* S, float* amp, int64_t rows, int nbins, int ld, void* stream);
int a3t_logmel_finish(float* mel, const int64_t* olens, int B, int F, int C, void* stream);

/* Masked L1/L2 loss (sedit_model.py:320-340).  scratch: float[2 + nblk*2].
 * loss_out[0] = sum_masked(|before-y|+|after-y|)/(n_masked+1e-10); d_before/d_after = gradients
 * times gscale (may be NULL for no-grad).  after == NULL: model without a postnet -- the second term is absent
 * (sedit_model.py:333-337) and d_after is not written. */
int a3t_mlm_loss(const float* before, const float* after, const float* target, const uint8_t* masked,
                 float* loss_out, float* d_before, float* d_after, float* scratch, int M, int C, int l2,
                 float gscale, void* stream);
int a3t_mlm_loss_scratch_floats(int M);

/* Trainer tail (espnet2/train/trainer.py:631-679, torch.optim.Adam, schedulers/noam_lr.py:58-65) on
 * ONE flat parameter buffer: grad-norm partials, then clip + finite check + Adam, no host sync.
 * norm_out[0] = total L2 norm; the update is skipped on device when it is not finite. */
int a3t_sumsq(const float* g, int64_t n, double* partial /*[1024]*/, void* stream);
int a3t_clip_adam(float* p, const float* g, float* m, float* v, const double* partial, float* norm_out,
                  int64_t n, float lr, float beta1, float beta2, float eps, int step, float clip,
                  float gscale, void* stream);

/* The same update with the step count on the device: state = int32[2] {updates applied, steps skipped}; the Noam
 * learning rate (base_lr * model_size^-0.5 * min(t^-0.5, t * warmup^-1.5), t = state[0]+1) and Adam's bias corrections
 * are evaluated on the device, and a step whose gradient norm is not finite advances neither (trainer.py:640-679). */
int a3t_clip_adam_noam(float* p, const float* g, float* m, float* v, const double* partial, float* norm_out, int64_t n,
                       int32_t* state, float base_lr, float model_size, float warmup, float beta1, float beta2,
                       float eps, float clip, float gscale, void* stream);

/* ParallelWaveGAN helpers (espnet2/gan_tts/wavenet/residual_block.py:114-169,
 * parallel_wavegan/upsample.py:22-189), channels-last [T][C]. */
/* g = tanh(xa+ca)*sigmoid(xb+cb), y/c [T][2H] (a|b), out [T][H] */
int a3t_pwg_gate(const float* y, const float* c, float* out, int64_t T, int H, void* stream);
/* o [T][R+S] -> x = (o[:, :R] + x) * sqrt(.5); skips += o[:, R:] */
int a3t_pwg_res_skip(const float* o, float* x, float* skips, int64_t T, int R, int S, void* stream);
/* nearest stretch by `scale` then 1-D smoothing conv (2*scale+1 taps, zero pad) per channel; c [B][Tin][C] -> out [B][Tin*scale][C] */
int a3t_pwg_upsample(const float* c, const float* w, float* out, int64_t B, int64_t Tin, int C, int scale,
                     void* stream);
/* x [B][T][C] -> y [B][T + 2 pad][C], edge frames replicated per utterance */
int a3t_replicate_pad(const float* x, float* y, int64_t B, int64_t T, int C, int pad, void* stream);
int a3t_bias_act(float* x, const float* bias, int64_t M, int C, int act, float scale, void* stream);

/* Dropout (torch.nn.Dropout sites of the path).  Counter-based: keep = f(key, element index), so the
 * same key reproduces the mask in the backward pass and inside GEMM epilogues; no mask tensors.
 * y = scale * x * keep/(1-p); in place allowed; x / y may be fp32 or bf16. */
int a3t_dropout(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, float p, uint32_t key, float scale,
                void* stream);
/* Backward of "residual + alpha*dropout(branch)": gm = g*mask/(1-p) (fp32 or bf16 operand of the branch's
 * GEMMs) and colsum += colsum_scale * column sums of gm (the branch's output-bias gradient). */
int a3t_dropout_bwd_cast(const float* g, void* gm, int gm_dtype, float* colsum, float colsum_scale, int M, int C,
                         float p, uint32_t key, void* stream);

/* One ParallelWaveGAN residual block (residual_block.py:114-169), fused: x and skips updated in place.
 * x [B*Tw][64], cu [B*Tw][80] (upsampled mel), g scratch [B*Tw][64], skips [B*Tw][64], all fp32 channels-last.
 * wt0 [272][128]: row k = tap*64 + in_channel (dilated k=3 conv, taps at t-dil, t, t+dil) | 192 + aux channel; column
 * n' = permuted gate channel: n' = 64*(c/32) + 32*half + c%32 for gate channel c (0..63), half 0 = tanh, 1 = sigmoid;
 * b0 [128] permuted the same way.  wt1 [64][128] = conv1x1_out.weight^T (columns 0..63 residual, 64..127 skip), b1 [128]. */
int a3t_pwg_block(float* x, const float* cu, const float* wt0, const float* b0, const float* wt1, const float* b1,
                  float* g, float* skips, int B, int Tw, int dil, void* stream);

/* On-device half of MLMCollateFn (espnet2/train/collate_fn.py:330-385): masked_position, speech / text segment ids and the
 * two padding masks painted from integer span lists.  fs / fe [B][P] int32: frame span of phone j (floor(fs * t / hop) taken
 * on the host in the alignment's dtype, collate_fn.py:236-237); alen [B] phones per utterance; sel [B][P] uint8: phone j is
 * masked (decided by the host's numpy-RNG draws, random_spans_noise_mask :387-446); mspan [B][S][2] int32 + nms [B]: explicit
 * frame spans to mask besides (span_boundary, the mean_phn_span == 0 and mlm_prob == 1 cases); flen / tlen [B] int32: valid
 * frames / phones.  Outputs: masked, speech_mask [B][Tm] uint8 (0 / 1), text_mask [B][Tp] uint8, sp [B][Tm] and tp [B][Tp]
 * int64 (all 0 when sega_emb == 0).  Later phones overwrite earlier ones where spans overlap (:335-341). */
int a3t_collate_paint(const int* fs, const int* fe, const int* alen, const uint8_t* sel, const int* mspan, const int* nms,
                      const int* flen, const int* tlen, uint8_t* masked, uint8_t* speech_mask, uint8_t* text_mask,
                      int64_t* sp, int64_t* tp, int B, int Tm, int Tp, int P, int S, int sega_emb, void* stream);
/* out[b][c] += sum_t x[b*T + t][c], fp32: gradient of a per-utterance vector added to every token of its utterance. */
int a3t_segment_colsum(const float* x, float* out, int B, int T, int C, void* stream);

const char* a3t_version(void);
/* Name (as rocprofv3 prints it, without "void " / "(GP)") of the kernel variant a3t_gemm's dispatcher launched last on
 * the calling thread -- lets a profiler harness attribute event-bracketed launches to kernel-trace rows. */
const char* a3t_gemm_last_kernel(void);
/* Kernel-selection override for A/B measurements and tests: 0 = never use the persistent 256x256 8-phase GEMM, 1 = whenever
 * the descriptor is legal for it, 2 = the built-in heuristic, -1 = re-read A3T_GEMM_8P.  Returns the previous mode. */
int a3t_gemm_8p_mode(int mode);
/* The same switch for the 384-column panel GEMM (A3T_GEMM_PN). */
int a3t_gemm_pn_mode(int mode);
/* The same switch for the streaming GEMM of the attention backward's score-sized products (dS K, dBD P, P^T dctx, dS^T (q+u):
 * espnet attention.py:64-96,145-209; gemm_bf16_tt.hip): 0 never, 1 whenever legal, 2 (default) when the grid fills the chip
 * (A3T_GEMM_TT). */
int a3t_gemm_tt_mode(int mode);
/* Weight gradients (token reductions, multi_layer_conv.py:52-63 / torch.nn.Linear backward): 0 = never use the 128 x 384-tile
 * 8-phase kernel with the deterministic split-K fold, 1 = whenever the descriptor is legal for it, 2 (default) = when its tiles
 * cover the output to >= 85 % and the launch has >= 96 workgroups, -1 = re-read A3T_GEMM_8P_TN3.  Returns the previous mode.
 * (a3t_gemm_8p_mode(0) switches it off together with every other 8-phase kernel.) */
int a3t_gemm_tn3_mode(int mode);
/* n (1..8) Linear weight gradients over the SAME K tokens -- dW_i[M_i][N_i] (op_i)= alpha_i * dy_i^T x_i, descriptors as a3t_gemm
 * takes them for a token reduction (a_rs == b_rs == 1, bf16 operands, fp32 C, taps <= 1, no epilogue options) -- in ONE launch of
 * the 128 x 384-tile kernel + one fold: the four small gradients of a Conformer block (linear_out and linear_q/k/v of
 * attention.py:40-96, pointwise_conv1/2 of convolution.py:56-79) share the K splits of one grid instead of paying four
 * prologues / folds on 16-20 K-tiles each.  Returns 0, a hipError_t, A3T_EINVAL (bad arguments) or -1: a member does not meet
 * the kernel's contract or the kernel is switched off -- nothing was launched, call a3t_gemm per member. */
int a3t_gemm_tn3_group(const a3t_gemm_desc* d, int n, void* stream);
/* Fused attention forward (a3t_attn_fwd, a3t_attn_fwd_train): 1 (default) = when the last round of 128-query blocks would fill at
 * most half of the chip, those blocks run as a launch of their own, split into 2..4 key ranges whose partial sums a small kernel
 * folds (attention.py:64-96 is associative in the keys once every range uses the block's one reference maximum); 0 = one
 * workgroup per block always; -1 = re-read A3T_ATTN_SPLIT.  Returns the previous mode. */
int a3t_attn_split_mode(int mode);
/* The library's grow-only device workspaces (the split-K slabs of the weight-gradient kernels, one per (device, stream), at most
 * 16 cached; the key-split workspace of the fused attention forward, one per device) are drained and freed; the next launch that
 * needs one allocates it again.  For a host that destroys streams or wants the memory back between jobs.  Returns 0. */
int a3t_release_workspaces(void);

#ifdef __cplusplus
}
#endif
#endif
