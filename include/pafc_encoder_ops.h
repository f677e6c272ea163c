/*
 * pafc_encoder_ops.h -- C ABI of the HBM-bound glue kernels around the WKV-6 scan and the GEMMs.
 *
 * These have no native counterpart in the reference: there they are chains of PyTorch element-wise ops
 * (each a full (B,T,C) round trip through HBM).  Each entry point cites the reference lines it fuses.
 * Conventions as in pafc_wkv6.h: plain device pointers, caller-owned buffers, asynchronous on `stream`,
 * 0 / negative PAFC_ERR_* return.  dtype is PAFC_F32 or PAFC_BF16 and applies to activations and parameters
 * alike unless stated.  All activations are channels-last (B, T, C) contiguous.
 */
#ifndef PAFC_ENCODER_OPS_H
#define PAFC_ENCODER_OPS_H

#include "pafc_wkv6.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Depthwise 1-D convolution over time in channels-last layout, bias fused.
 *   y[b][t][c] = bias[c] + sum_k w[c][k] * x[b][t + k - left_pad][c]        (x = 0 outside [0, T_in))
 * Replaces ConvolutionModule.depthwise_conv -- nn.Conv1d(C, C, K, padding=(K-1)/2, groups=C) between two
 * transposes, wenet/transformer/convolution.py:60-68,93,131 -- without the transposes.
 *   x: (B, T_in, C), w: (C, 1, K) (the Conv1d weight as stored), bias: (C) or NULL, y: (B, T_out, C).
 *   non-causal: left_pad = (K-1)/2, T_out = T_in; causal with a left-padded/cached input: left_pad = 0,
 *   T_out = T_in - (K-1).  C % 128 == 0, K <= 31.
 * glu != 0: x is (B, T_in, 2C) and the convolution input is x[..., :C] * sigmoid(x[..., C:]) (F.glu,
 *   convolution.py:128), rounded to the activation dtype as the reference does before convolving.
 * lens (int32, (B)) or NULL: frames t >= lens[b] are read as zero (the masked_fill_ of convolution.py:109-110).
 */
int pafc_dwconv1d_cl(int dtype, int B, int T_in, int C, int K, int left_pad, int T_out, const void *x,
                     const void *w, const void *bias, void *y, int glu, const int32_t *lens,
                     pafc_stream_t stream);

/* The same kernel with an explicit row stride of x (ldx elements between frames: the input may be a column slice of a
 * wider activation, as xBC inside Mamba-2's in_proj output) and act: 0 none, 1 GLU on the input (x holds 2C columns),
 * 2 SiLU on the output (Mamba-2's causal conv1d(k = 4) + SiLU: left_pad = K - 1, T_out = T_in).  K in {3, 4, 7, 15, 31}.
 * Batch entries are T_in * ldx elements apart. */
int pafc_dwconv1d_cl_ex(int dtype, int B, int T_in, int C, int K, int left_pad, int T_out, const void *x, long ldx,
                        const void *w, const void *bias, void *y, int act, const int32_t *lens, pafc_stream_t stream);

/* The depthwise convolution with the conv module's LayerNorm + SiLU as its epilogue -- `self.activation(self.norm(x))` after
 * `depthwise_conv`, wenet/transformer/convolution.py:131-138 with cnn_module_norm: layer_norm -- in one pass:
 *   y = SiLU(LayerNorm_C(conv(x)) * gamma + beta), every intermediate rounded to bf16 where the module chain stores one
 *   (convolution output, LayerNorm output), two-pass variance.
 * bf16 only, C == 512 (a block of the kernel owns all the channels of its frames), K in {15, 31}; other shapes: PAFC_ERR_UNSUPPORTED
 * (callers then run pafc_dwconv1d_cl + pafc_add_layernorm).  x: (B, T_in, ldx >= C) (ldx: row stride in elements), gamma / beta: (C). */
int pafc_dwconv1d_cl_ln_silu(int dtype, int B, int T_in, int C, int K, int left_pad, int T_out, const void *x, long ldx,
                             const void *w, const void *bias, const void *gamma, const void *beta, float eps, void *y,
                             const int32_t *lens, pafc_stream_t stream);

/* Gradients of the same convolution for the training step (config c4; the reference differentiates nn.Conv1d through
 * autograd: convolution.py:131 under train_utils.py:646-660).  The input gradient is the forward kernel itself on dy with
 * the taps reversed and left_pad' = K - 1 - left_pad; this entry point is the other half:
 *   dw[c][k] = sum_{b,t} dy[b][t][c] * x[b][t + k - left_pad][c],   dbias[c] = sum_{b,t} dy[b][t][c]
 *   x: (B, T_in, ldx >= C), dy: (B, T_out, C) in `dtype`; dw: (C, K) and dbias: (C) or NULL in float32 (summed in a
 *   fixed order: deterministic).  workspace: pafc_dwconv1d_cl_wgrad_workspace_bytes(B, T_out, C, K) bytes. */
size_t pafc_dwconv1d_cl_wgrad_workspace_bytes(int B, int T_out, int C, int K);
int pafc_dwconv1d_cl_wgrad(int dtype, int B, int T_in, int C, int K, int left_pad, int T_out, const void *x, long ldx,
                           const void *dy, float *dw, float *dbias, void *workspace, size_t workspace_bytes,
                           pafc_stream_t stream);

/* Residual add + LayerNorm (+ SiLU, + second LayerNorm, + padding masks) in one pass over (rows, C).
 *   x_new = x + alpha * y            (y == NULL: x_new = x; mask_y: rows with t >= lens[b] of y count as zero)
 *   out1  = LN(x_new; gamma1, beta1) (silu1: SiLU on top; zero1: rows with t >= lens[b] are written as zero)
 *   out2  = LN(out1;  gamma2, beta2) (optional)
 * Fuses the reference's `x = residual + ff_scale * dropout(branch(x))` followed by the next sub-block's pre-norm
 * (ConformerEncoderLayer.forward, wenet/transformer/encoder_layer.py:201-259), the conv module's
 * masked_fill / norm / activation (convolution.py:109-110,132-136,140-141), ln_x of the time-mix
 * (src/model.py:323) and norm_final + the next layer's first pre-norm.  rows = B*T, row index = b*T + t.
 * dtype: x, y, x_out, gamma*, beta*; dtype_out: out1/out2 (bf16 out of an fp32 stream = the slot's input cast,
 * rwkv_wrapper_bidirectional.py:40-41).  ld1/ld2: row strides of out1/out2 in elements (a (rows, 2C) buffer can
 * receive two LayerNorms side by side).  C % 8 == 0, C <= 1024.  eps as nn.LayerNorm (1e-5).
 * dtype_out PAFC_SPLIT_BF16 (fp32 streams only): out1 / out2 rows are [hi (C) | lo (C)] bf16 planes of the fp32 result
 * (ld >= 2C) -- the A operand of pafc_gemm_ph_ex(a_split = 1); nothing is rounded to bf16 on the way. */
int pafc_add_layernorm(int dtype, int dtype_out, int rows, int C, const void *x, const void *y, float alpha,
                       const int32_t *lens, int T, int mask_y, void *x_out, const void *gamma1, const void *beta1,
                       void *out1, long ld1, int silu1, int zero1, const void *gamma2, const void *beta2, void *out2,
                       long ld2, float eps, pafc_stream_t stream);

/* The same with a second output form: dtype_out2 = dtype_out, or PAFC_SPLIT_BF16 beside dtype_out = PAFC_F32 (norm_final
 * in fp32 for the caller + the next layer's first pre-norm as the planes its GEMM reads), and with row statistics for a
 * LayerNorm folded into the projection that follows (pafc_gemm_bf16_ph_ln): stats_x / stats_out1 (either may be NULL) receive
 * float2 [rows][8] -- pair 0 = (sum, sum of squares) of the row of x_new / of out1 as stored, pairs 1..7 zero. */
int pafc_add_layernorm_ex(int dtype, int dtype_out, int dtype_out2, int rows, int C, const void *x, const void *y, float alpha,
                          const int32_t *lens, int T, int mask_y, void *x_out, const void *gamma1, const void *beta1,
                          void *out1, long ld1, int silu1, int zero1, const void *gamma2, const void *beta2, void *out2,
                          long ld2, float eps, float *stats_x, float *stats_out1, pafc_stream_t stream);
/* A pre-norm LayerNorm folded into the two bf16 GEMMs either side of it (csrc/gemm_ph.hip, LNF; encoder_layer.py:201-259:
 * `x = residual + branch(...)` then `norm(x)` then the branch's first projection):
 *   LN(x) W^T + b = rstd (x W'^T - mean csum) + b',   W' = gamma * W (rounded to bf16), csum[n] = sum_k W'[n][k], b' = b + W beta.
 * ln_mode 2, the residual GEMM that produces x (N == ln_c == 512): as pafc_gemm_bf16_ph with a residual, and stats
 *            (float2 [M][8]) receives (sum, sum of squares) of every row's eight 64-column slices (of the fp32 result).
 * ln_mode 1, the projection that consumes LN(x) (act 1 SiLU or 4 GLU, no residual, alpha 1, K == ln_c): A = x itself,
 *            W = W', bias = b' (bf16), csum fp32 (N); mean / rstd per row from stats.  The normalised tensor never exists. */
int pafc_gemm_bf16_ph_ln(long M, int N, int K, const void *A, long lda, const void *W, long ldw, const void *bias,
                         const void *residual, long ldr, void *out, long ldo, float alpha, int act, int ln_mode, float *stats,
                         const float *csum, int ln_c, float ln_eps, int tile_m, pafc_stream_t stream);

/* fp32 (rows, cols), rows ldx apart -> bf16 planes: out row = [hi | lo] with lo at column lo_off (triple = 0: an
 * activation for pafc_gemm_ph_ex(a_split = 1)) or [hi | hi | lo] at columns 0, cols, 2 cols (triple = 1: its weight).
 * cols % 8 == 0. */
int pafc_split_planes(long rows, int cols, const float *x, long ldx, void *out, long ldo, long lo_off, int triple,
                      pafc_stream_t stream);

/* LayerNorm backward for the training step (config c4): the seven nn.LayerNorm of the layer under autograd
 * (encoder_layer.py:201-259, convolution.py:136, src/model.py:323; train_utils.py:646-660).  mean / rstd are recomputed
 * from x (one wave per row, as pafc_add_layernorm):
 *   xhat = (x - mean) rstd;  dx = rstd (dy gamma - mean_c(dy gamma) - xhat mean_c(dy gamma xhat))
 *   dgamma_dbeta: float32 (2, C) = [sum_rows dy xhat ; sum_rows dy], summed in a fixed order (deterministic).
 * x, gamma, dx in dtype_x; dy in dtype_dy (a bf16 output of an fp32 norm has a bf16 gradient); C % 8 == 0, C <= 1024.
 * workspace: pafc_layernorm_bwd_workspace_bytes(rows, C) bytes. */
size_t pafc_layernorm_bwd_workspace_bytes(long rows, int C);
int pafc_layernorm_bwd(int dtype_x, int dtype_dy, long rows, int C, const void *x, const void *dy, const void *gamma,
                       float eps, void *dx, float *dgamma_dbeta, void *workspace, size_t workspace_bytes,
                       pafc_stream_t stream);
/* ... + dx_add (dtype_x, may be null): dx = the norm's input gradient + dx_add, the gradient that reaches x past the norm (the
 * residual path of a pre-norm branch, encoder_layer.py:201-256) -- autograd's accumulation pass folded into this one. */
int pafc_layernorm_bwd_add(int dtype_x, int dtype_dy, long rows, int C, const void *x, const void *dy, const void *gamma, float eps,
                           const void *dx_add, void *dx, float *dgamma_dbeta, void *workspace, size_t workspace_bytes,
                           pafc_stream_t stream);

/* LayerNorm + SiLU of the conv module in the training step (convolution.py:136-138, `activation(norm(x))` between the depthwise
 * convolution and pointwise_conv2), one kernel each way: y = silu(LayerNorm(x)), arithmetic in fp32 against the norm's own
 * parameters (dtype_g: fp32 under autocast, or bf16), x / y / dy / dx in dtype_x (bf16: the convolution's output; fp32 with fp32
 * parameters).  Backward recomputes mean / rstd / z from x: dx, dgamma_dbeta float32 (2, C) as pafc_layernorm_bwd (same workspace). */
int pafc_layernorm_silu_fwd(int dtype_x, int dtype_g, long rows, int C, const void *x, const void *gamma, const void *beta, float eps,
                            void *y, pafc_stream_t stream);
int pafc_layernorm_silu_bwd(int dtype_x, int dtype_g, long rows, int C, const void *x, const void *dy, const void *gamma,
                            const void *beta, float eps, void *dx, float *dgamma_dbeta, void *workspace, size_t workspace_bytes,
                            pafc_stream_t stream);

/* Skinny bf16 GEMM for the streaming chunk step (csrc/gemm_skinny.hip): out = act(alpha * A W^T + bias [+ residual]) for FEW
 * rows -- the projections of a layer while it serves 64-frame chunks with state carry (the same nn.Linear / 1x1 Conv1d call
 * sites as pafc_gemm_bf16: positionwise_feed_forward.py:47-55, convolution.py:118-141, src/model.py:286-324,
 * encoder_layer.py:201-259).  One block per 16 output columns walks all rows and all of K; ~4 us per launch at 64 rows.
 * A: (M, K) row stride lda; W: (N, K) row stride ldw (nn.Linear layout); batch entries strideX elements apart (0 = shared).
 * act: 0 none, 1 SiLU, 2 tanh, 3 ReLU, 4 GLU (F.glu over the N rows of W: value rows [0, N/2), gate rows [N/2, N); out has N/2
 * columns; no residual).  K % 32 == 0, N % 16 == 0 (GLU: 32), 16-byte aligned rows.
 * LayerNorm in front of the projection, folded (as pafc_gemm_bf16_ph_ln): ln_stats_in = float2 [M][ln_parts_in] partial
 * (sum, sum of squares) of the UN-normalised A rows, ln_csum[N] = column sums of W' = bf16(gamma * W), bias = b + W beta;
 * out = act(rstd (A W'^T - mean csum) + bias).  ln_stats_out (or null) receives float2 [batch][M][N_out / 16] partial
 * statistics of the rows written (as stored, after rounding) for the next folded LayerNorm. */
int pafc_gemm_skinny_bf16(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W, long ldw,
                          long strideW, const void *bias, long strideBias, const void *residual, long ldr, long strideR,
                          void *out, long ldo, long strideO, float alpha, int act, const float *ln_stats_in, int ln_parts_in,
                          const float *ln_csum, float ln_eps, float *ln_stats_out, pafc_stream_t stream);
/* The same with the other things a chunk step wants inside the launch:
 *   round_first: out = bf16(alpha * A W^T) + bias, rounded again -- where `ww = t @ time_decay_w2; w = time_decay + ww`
 *                rounds (src/model.py:289);
 *   ln_self:     the folded LayerNorm's row statistics are formed from the operand itself (ln_stats_in null, ln_csum given);
 *   mix_maa:     the operand is the token shift + first lerp of the time-mix, xxx = x + (x_prev - x) * maa_x
 *                (src/model.py:274-276), formed in registers from A = x (M = B * mix_T rows), its predecessor row and maa_x (K);
 *                mix_prev (B, K) or null = the frame before each sequence (the streaming carry; null: zero as ZeroPad2d);
 *                with act = 2 and W = time_maa_rkvw_w1^T this is pafc_tmix_lora_down_bf16_prev for a handful of rows;
 *   norm_gamma, norm_beta (K) + norm_eps: the operand is silu(LayerNorm(A)) -- the conv module's norm and activation in front
 *                of pointwise_conv2 (convolution.py:136-139) -- each rounded to bf16 as the two separate passes round. */
int pafc_gemm_skinny_bf16_ex(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W, long ldw,
                             long strideW, const void *bias, long strideBias, const void *residual, long ldr, long strideR,
                             void *out, long ldo, long strideO, float alpha, int act, int round_first,
                             const float *ln_stats_in, int ln_parts_in, int ln_self, const float *ln_csum, float ln_eps,
                             float *ln_stats_out, const void *mix_maa, const void *mix_prev, int mix_T, const void *norm_gamma,
                             const void *norm_beta, float norm_eps, pafc_stream_t stream);

/* The decay LoRA of the time-mix for a handful of rows in ONE launch (src/model.py:286-289):
 *   w = bf16(bf16(tanh(x D1)) D2) + time_decay, rounded again -- the two few-rows GEMMs of the chunk step
 *   (pafc_gemm_skinny_bf16_ex with act = 2, then with round_first) with the 64-wide hidden tile kept in LDS; same K split, same
 *   order of the partial sums and same roundings, so the result is bit-identical to the two launches.
 *   x: (M, ldx >= C) bf16 rows; d1n: (H = 64, C) = time_decay_w1^T; d2n: (C, H) = time_decay_w2^T; bias: (C) time_decay or NULL
 *   (NULL: no second rounding); out: (M, ldo >= C). */
int pafc_decay_lora_skinny_bf16(long M, int C, int H, const void *x, long ldx, const void *d1n, const void *d2n, const void *bias,
                                void *out, long ldo, pafc_stream_t stream);

/* Token shift + first lerp of the time-mix for ndir directions from one read of x (src/model.py:274-276):
 *   xx_d = shift_d(x) - x,  out[d] = x + xx_d * maa_x_d;   shift_0 = x_{t-1} (or x_{t+1} when reverse0), shift_1 = x_{t+1}
 * x: (B, T, C); maa_x0/1: (C); out: (ndir, B, T, C).  Zero beyond the sequence ends, like ZeroPad2d((0,0,1,-1)). */
int pafc_tmix_shift_mix(int dtype, int B, int T, int C, int ndir, int reverse0, const void *x, const void *maa_x0,
                        const void *maa_x1, void *out, pafc_stream_t stream);

/* The four data-dependent lerps of the time-mix (src/model.py:280-284): z_q = x + xx * (maa_q + m_q), q = r,k,v,w.
 * m: (ndir, 4, B*T, C) LoRA outputs; maa: (ndir, 4, C); z: (4, ndir, B*T, C). */
int pafc_tmix_mix4(int dtype, int B, int T, int C, int ndir, int reverse0, const void *x, const void *m,
                   const void *maa, void *z, pafc_stream_t stream);

/* pafc_tmix_mix4 with the LoRA up-projection fused in (bf16): m_q = bf16(t[:, 32q:32q+32] . W2[q]) is computed on the
 * matrix cores inside the pass, so the 4 x ndir LoRA maps never touch HBM (the K = 32 `torch.bmm` of src/model.py:278).
 * t: (ndir, B*T, 128) = tanh(xxx W1); w2t: (ndir, 4, C, 32) = time_maa_rkvw_w2 with K innermost. C % 64 == 0. */
int pafc_tmix_lora_mix4_bf16(int B, int T, int C, int ndir, int reverse0, const void *x, const void *t, const void *w2t,
                             const void *maa, void *z, pafc_stream_t stream);

/* Token shift + the first lerp + the LoRA down-projection + tanh in one pass (bf16; src/model.py:273-277):
 * t = bf16(tanh((x + (x_neighbour - x) * maa_x) @ time_maa_rkvw_w1)).  x: (B, T, C); maa_x: (ndir, C); w1n: (ndir, N, C) =
 * time_maa_rkvw_w1^T (K innermost); t: (ndir, B*T, N).  Direction 0 looks back (or forward with reverse0), direction 1
 * forward, as pafc_tmix_shift_mix.  The weights stay in LDS and xxx never exists in memory.  Built for C = 512, N = 128
 * (PAFC_ERR_UNSUPPORTED otherwise: pafc_tmix_shift_mix + a GEMM). */
int pafc_tmix_lora_down_bf16(int B, int T, int C, int N, int ndir, int reverse0, const void *x, const void *maa_x,
                             const void *w1n, void *t, pafc_stream_t stream);

/* The three passes above for a STREAMING chunk (recurrent-state carry between chunks, BASELINE configs[2]; the reference's
 * time-mix has no streaming form -- `self.time_shift = nn.ZeroPad2d((0, 0, 1, -1))`, src/model.py:262,274 -- so the carried
 * quantity is defined by what makes chunked == whole-sequence: the last frame of the previous chunk):
 * prev (B, C) = the frame before each sequence's first one, used by the backward-looking direction instead of the zero the
 * offline form pads with; null = the offline form.  Everything else as in the functions without `_prev`. */
int pafc_tmix_shift_mix_prev(int dtype, int B, int T, int C, int ndir, int reverse0, const void *x, const void *maa_x0,
                             const void *maa_x1, const void *prev, void *out, pafc_stream_t stream);
int pafc_tmix_lora_mix4_bf16_prev(int B, int T, int C, int ndir, int reverse0, const void *x, const void *t, const void *w2t,
                                  const void *maa, const void *prev, void *z, pafc_stream_t stream);
int pafc_tmix_lora_down_bf16_prev(int B, int T, int C, int N, int ndir, int reverse0, const void *x, const void *maa_x,
                                  const void *w1n, const void *prev, void *t, pafc_stream_t stream);

/* The decay LoRA of the time-mix in one pass (bf16): w = bf16( bf16(tanh(zw @ time_decay_w1)) @ time_decay_w2 ) [+ bias]
 * (src/model.py:286-287: ww = tanh(xw @ time_decay_w1) @ time_decay_w2; w = time_decay + ww).  zw: (ndir, rows, C) the
 * fourth lerp; d1n: (ndir, H, C) = time_decay_w1^T, d2n: (ndir, C, H) = time_decay_w2^T (K innermost); bias: (ndir, C)
 * time_decay or null (the bidirectional scan adds it itself); w: (ndir, rows, C).  Both weight matrices stay in LDS and the
 * H-wide hidden tensor never exists in memory.  Built for C = 512, H = 64 (PAFC_ERR_UNSUPPORTED otherwise: two GEMMs). */
int pafc_decay_lora_bf16(long rows, int C, int H, int ndir, const void *zw, const void *d1n, const void *d2n,
                         const void *bias, void *w, pafc_stream_t stream);

/* Backward of the two element-wise groups of the time-mix block for the training step (config c4; the reference
 * differentiates src/model.py:274-284 op by op through autograd).  One direction per call (reverse: the shift is
 * x_{t+1}); sums in fp32, maa gradients in float32 summed in a fixed order (deterministic).
 *   shift_mix:  xxx = x + (shift(x) - x) maa_x.  dxxx (B, T, C) -> dx (B, T, C), dmaa_x float32 (C).
 *   mix4:       z_q = x + (shift(x) - x) (maa_q + m_q), q = r, k, v, w.  m, dm: (4, B*T, C); maa: (4, C); dz_q (B*T, C)
 *               -> dx (B, T, C), dm, dmaa float32 (4, C).
 * C % 8 == 0, C <= 1024; workspace: pafc_tmix_bwd_workspace_bytes(B*T, C) bytes. */
size_t pafc_tmix_bwd_workspace_bytes(long rows, int C);
int pafc_tmix_shift_mix_bwd(int dtype, int B, int T, int C, int reverse, const void *x, const void *maa_x, const void *dxxx,
                            void *dx, float *dmaa_x, void *workspace, size_t workspace_bytes, pafc_stream_t stream);
int pafc_tmix_mix4_bwd(int dtype, int B, int T, int C, int reverse, const void *x, const void *m, const void *maa,
                       const void *dz_r, const void *dz_k, const void *dz_v, const void *dz_w, void *dx, void *dm,
                       float *dmaa, void *workspace, size_t workspace_bytes, pafc_stream_t stream);
/* ... with dm laid out (B*T, 4, C) instead of (4, B*T, C) (round 6: the gradients of the LoRA-up matrices src/model.py:277-278 read it
 * as one (B*T, 4 C) operand of pafc_gemm_tn_bf16 and as four column blocks of pafc_gemm_bf16). */
int pafc_tmix_mix4_bwd_rows(int dtype, int B, int T, int C, int reverse, const void *x, const void *m, const void *maa,
                       const void *dz_r, const void *dz_k, const void *dz_v, const void *dz_w, void *dx, void *dm,
                       float *dmaa, void *workspace, size_t workspace_bytes, pafc_stream_t stream);

/* 3x3 stride-2 convolution + bias (+ ReLU), NHWC, bf16, as an implicit GEMM on the matrix cores:
 *   out[b][t2][f2][co] = act(bias[co] + sum_{kh,kw,ci} w[co][ci][kh][kw] in[b][2 t2 + kh][2 f2 + kw][ci]),
 *   T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1.
 * Replaces the second Conv2d + ReLU of Conv2dSubsampling4 (wenet/transformer/subsampling.py:187-192).
 * in: (B, T1, F1, Ci); w_tap_co_ci: the Conv2d weight re-laid out as (9, Co, Ci) = weight.permute(2,3,0,1);
 * bias: (Co) or NULL; out: (B, T2, F2, Co).  Ci % 64 == 0, Co % 128 == 0. */
int pafc_conv3x3s2_nhwc_bf16(int B, int T1, int F1, int Ci, int Co, const void *in, const void *w_tap_co_ci,
                             const void *bias, void *out, int relu, pafc_stream_t stream);

/* out[b][t1][f1][c] = relu(bias[c] + sum_{kh,kw} w[c][kh][kw] * x[b][2 t1 + kh][2 f1 + kw]): the first Conv2d(1, C, 3, 2)
 * + ReLU of Conv2dSubsampling4 (wenet/transformer/subsampling.py:185-186) written NHWC, the layout the second
 * convolution reads.  x: (B, T, F) bf16 features; w_c_9: the Conv2d weight (C, 1, 3, 3) as stored; out: (B, T1, F1, C) with
 * T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1.  fp32 accumulation, one rounding.  C % 8 == 0 and 256 % (C / 8) == 0. */
int pafc_conv3x3s2_c1_nhwc_bf16(int B, int T, int F, int C, const void *x, const void *w_c_9, const void *bias, void *out,
                                int relu, pafc_stream_t stream);

/* Weight / bias gradient of that first convolution for the training step (config c4; autograd through
 * Conv2dSubsampling4.conv[0:2], subsampling.py:201-226).  act: the forward's ReLU output (B, T1, F1, C), dact: its incoming
 * gradient; the ReLU mask is applied here.  dw_db: float32 (10, C) = [9 taps (kh, kw) ; bias], summed in a fixed order.
 * workspace: pafc_conv3x3s2_c1_wgrad_workspace_bytes(B, T, C) bytes.  (The input has no gradient: it is the features.) */
size_t pafc_conv3x3s2_c1_wgrad_workspace_bytes(int B, int T, int C);
int pafc_conv3x3s2_c1_wgrad_bf16(int B, int T, int F, int C, const void *x, const void *act, const void *dact, float *dw_db,
                                 void *workspace, size_t workspace_bytes, pafc_stream_t stream);

/* The two subsampling convolutions for fp32 activations at bf16 matrix-core speed: every fp32 value travels as
 * hi + lo, hi = bf16(x), lo = bf16(x - hi) (16 significant bits); a product is three bf16 MFMAs (hi hi + lo hi + hi lo)
 * accumulated in fp32 -- ~1e-5 relative to the fp32 convolution.  _c1_: x (B, T, F) fp32, w (C, 1, 3, 3) fp32, bias
 * fp32 -> out_hi / out_lo (B, T1, F1, C) bf16 planes.  second convolution: those planes + the (9, Co, Ci) weight split
 * the same way by the caller -> out (B, T2, F2, Co) fp32 (+ bias, ReLU).  Shapes and limits as the bf16 entry points. */
int pafc_conv3x3s2_c1_nhwc_f32split(int B, int T, int F, int C, const float *x, const float *w_c_9, const float *bias,
                                    void *out_hi, void *out_lo, int relu, pafc_stream_t stream);
/* ... with pixel_stride elements between the pixels of each plane (out_lo = out_hi + C, pixel_stride = 2 C: one tensor
 * (B, T1, F1, 2 C) = [hi C | lo C] per pixel, the input form of pafc_conv3x3s2_nhwc_split_ph). */
int pafc_conv3x3s2_c1_nhwc_f32split_ps(int B, int T, int F, int C, const float *x, const float *w_c_9, const float *bias,
                                       void *out_hi, void *out_lo, long pixel_stride, int relu, pafc_stream_t stream);
int pafc_conv3x3s2_nhwc_f32split(int B, int T1, int F1, int Ci, int Co, const void *in_hi, const void *in_lo,
                                 const void *w_hi_tap_co_ci, const void *w_lo_tap_co_ci, const float *bias, float *out,
                                 int relu, pafc_stream_t stream);

/* Weight gradient of nn.Linear for the training step (config c4): dw[m][n] = sum_r dy[r][m] * x[r][n] -- what autograd
 * computes as grad_output^T @ input for every projection of the encoder layer (positionwise_feed_forward.py:47-55,
 * convolution.py:118-141, rwkv_v6/src/model.py:286-324 under train_utils.py:646-660).  dy: (R, M) and x: (R, N) bf16,
 * row strides lda / ldb elements (multiples of 8), R = batch x time; M, N multiples of 8.  dw: (M, N) contiguous in
 * dw_dtype (PAFC_F32: fp32 master weights get the fp32 sum, no intermediate bf16 rounding; PAFC_BF16).  dbias: (M) in
 * dw_dtype or NULL -- the bias gradient sum_r dy[r][m], from the same pass over dy.  The R axis is split over several
 * blocks per output tile and the partial tiles are added in a fixed order (deterministic).
 * workspace: pafc_gemm_tn_workspace_bytes(R, M, N) bytes, 16-byte aligned like dy, x and dw. */
size_t pafc_gemm_tn_workspace_bytes(long R, int M, int N);
int pafc_gemm_tn_bf16(long R, int M, int N, const void *dy, long lda, const void *x, long ldb, void *dw, void *dbias,
                      int dw_dtype, void *workspace, size_t workspace_bytes, pafc_stream_t stream);
/* `batch` products of one shape in one launch pair (the r / k / v weight gradients of a time-mix block): entry z reads
 * dy + z * stride_dy and x + z * stride_x (elements, multiples of 8) and writes dw + z * M * N (dbias + z * M).
 * workspace: pafc_gemm_tn_batched_workspace_bytes(R, M, N, batch). */
size_t pafc_gemm_tn_batched_workspace_bytes(long R, int M, int N, int batch);
int pafc_gemm_tn_bf16_batched(long R, int M, int N, int batch, const void *dy, long lda, long stride_dy, const void *x, long ldb,
                              long stride_x, void *dw, void *dbias, int dw_dtype, void *workspace, size_t workspace_bytes,
                              pafc_stream_t stream);

/* fp32 results from bf16 operands on the small tiles of csrc/gemm_bf16.hip (128 x 128 / 128 x 64 / 64 x 64, picked by row count): the
 * operand forms of pafc_gemm_ph_ex for problems too small for its 256-wide tiles -- a_split != 0: A = planes [hi K | lo K] of an
 * fp32 activation, W = [hi | hi | lo] (N x 3 K), three bf16 products per fp32 product (~2^-16); a_split == 0: plain bf16 A (M, K),
 * W (N, K).  out_kind 1: fp32 out (+ fp32 residual, which may alias out); 2: the fp32 result as bf16 planes hi | lo (lo at column
 * offset lo_off, ldo in bf16 elements; no residual).  bias fp32 (added as given), act 0 none / 1 SiLU / 2 tanh / 3 ReLU applied to
 * alpha * product + bias (before the residual add, as pafc_gemm_ph_ex has it).  N % 8 == 0, K % 64 == 0, 16-byte aligned rows.
 * The same call sites as pafc_gemm_ph_ex (positionwise_feed_forward.py:47-55, convolution.py:118-141, encoder_layer.py:201-259,
 * rwkv_wrapper_bidirectional.py:55-56) for decode batches of a few hundred to a few thousand rows. */
int pafc_gemm_bf16_f32out(long M, int N, int K, const void *A, long lda, int a_split, const void *W, long ldw, const float *bias,
                          const float *residual, long ldr, void *out, int out_kind, long ldo, long lo_off, float alpha, int act,
                          void *workspace, size_t workspace_bytes, pafc_stream_t stream);
/* ... with the split A's planes alternating in blocks of a_plane_block columns ([hi PB | lo PB] ..., PB a power of two >= 64 that
 * divides K; 0 = [hi K | lo K]), as pafc_gemm_ph_ex2 takes it: Conv2dSubsampling4's plane output into Linear(F' C, odim) at few rows. */
int pafc_gemm_bf16_f32out_pb(long M, int N, int K, const void *A, long lda, int a_split, int a_plane_block, const void *W, long ldw,
                             const float *bias, const float *residual, long ldr, void *out, int out_kind, long ldo, long lo_off,
                             float alpha, int act, void *workspace, size_t workspace_bytes, pafc_stream_t stream);
/* Bytes of workspace with which pafc_gemm_bf16_f32out splits K over several blocks per tile (a few hundred rows against a long K:
 * partials + a second, reducing launch, deterministic); 0: the problem runs as one launch, workspace may be null. */
size_t pafc_gemm_bf16_f32out_workspace_bytes(long M, int N, int K, int a_split);

/* ---- CTC loss of the training step, from the logits (csrc/ctc_loss.hip) -------------------------------------------------------
 * `ys_hat.log_softmax(2)` + `torch.nn.CTCLoss(reduction="sum", zero_infinity=True)` of CTC.forward (wenet/transformer/ctc.py:
 * 53-82) and their autograd, without the (B, T, V) log-probability tensor and without a host round trip.
 * logits (B, T, V) bf16 or fp32, rows ldl elements apart; hlens (B) int32 valid frames; ys (B, ldy) int64 targets (entries beyond
 * ylens[b] are not read); ylens (B) int32; max_target_len >= every ylens[b] sizes the workspace.
 * forward:  nll[b] = -log p(y_b | logits_b), 0 for an utterance without an alignment (zero_infinity); leaves the row
 *           statistics and label occupancies in `workspace` for backward.
 * backward: dlogits (B, T, ldg >= V) in the logits' dtype = grad_out[0] * scale * d(sum_b nll[b]) / d logits -- the gradient
 *           through the log-softmax; columns V .. ldg - 1 and rows t >= hlens[b] are written as zeros.  grad_out: one float on
 *           the device.  Same workspace, unmodified since forward. */
size_t pafc_ctc_loss_workspace_bytes(int B, int T, int max_target_len);
int pafc_ctc_loss_forward(int dtype, int B, int T, int V, const void *logits, long ldl, const int32_t *hlens, const int64_t *ys,
                          int ldy, const int32_t *ylens, int max_target_len, int blank, float *nll, void *workspace,
                          size_t workspace_bytes, pafc_stream_t stream);
int pafc_ctc_loss_backward(int dtype, int B, int T, int V, const void *logits, long ldl, const int32_t *hlens, const int64_t *ys,
                           int ldy, const int32_t *ylens, int max_target_len, int blank, const float *nll, const float *grad_out,
                           float scale, void *dlogits, long ldg, const void *workspace, size_t workspace_bytes,
                           pafc_stream_t stream);

/* ---- fp32 GEMM with a fused epilogue on the fp32 matrix cores (csrc/gemm_f32.hip) --------------------------------------
 * out (M, N) = act(alpha * A (M, K) . W (N, K)^T + bias (N) + residual (M, N)), batch entries strideX elements apart (0 = shared;
 * bias may be null, residual may be null or alias out).  Exact fp32 products with fp32 accumulation: the arithmetic of the
 * reference's fp32 nn.Linear / 1 x 1 Conv1d / torch.bmm call sites (positionwise_feed_forward.py:47-55, convolution.py:118-141,
 * rwkv_v6/src/model.py:277-324, subsampling.py:218-224, ctc.py:106-114, encoder_layer.py:201-259) for a model without the bf16
 * slot, and of the few-rows fp32 products of the other precision modes.  act: 0 none, 1 SiLU, 2 tanh, 3 ReLU.  bias is added as
 * given (not scaled by alpha); everything is applied to the fp32 accumulator.  K, lda, ldw and the batch strides of A and W
 * multiples of 4, A and W 16-byte aligned (else PAFC_ERR_UNSUPPORTED / PAFC_ERR_ALIGNMENT); any M, N.  Asynchronous on
 * `stream`, no workspace, no global state, safe to issue from several streams at once. */
int pafc_gemm_f32(long M, int N, int K, int batch, const float *A, long lda, long strideA, const float *W, long ldw, long strideW,
                  const float *bias, long strideBias, const float *residual, long ldr, long strideR, float *out, long ldo,
                  long strideO, float alpha, int act, pafc_stream_t stream);

/* Glue of the Mamba-2 block around the chunked scan (restated from the published block -- the reference only wraps the
 * third-party mamba_ssm Mamba2: mamba_att_wrapper.py:24-35, mamba2_bidirectional.py:12-36; PARITY UNPINNED; headdim 64,
 * d_state 128, ngroups 1).  xbc: (B, L, d_inner + 256) = [x | B | C] after conv1d + SiLU; dt_raw: the dt columns of
 * in_proj's output (rows ld_dt apart); z: its gate columns (rows ld_z apart); dt_bias, A_log, D: (d_inner / 64) fp32.
 * prep -> the fp32 operand planes (B, L, d_inner) of the two scans: r0, r1 = C halves, k0, k1 = a_{t+1} * B halves,
 * v = dt * x, w = log(-log a_{t+1}) with dt = softplus(dt_raw + dt_bias), a = exp(dt * -exp(A_log)).
 * finish: out = RMSNorm((y0 + y1 + ((B . C) dt + D) x) * SiLU(z)) * norm_weight, roundings as the module's ops. */
int pafc_mamba2_prep(int dtype, int B, int L, int d_inner, const void *xbc, const void *dt_raw, long ld_dt,
                     const float *dt_bias, const float *A_log, float *r0, float *r1, float *k0, float *k1, float *v,
                     float *w, pafc_stream_t stream);
/* y1 may be NULL; diag != 0 adds the (B . C) dt x term (needed when the scans ran on the WKV kernel, which leaves the
 * s = t term out; pafc_mamba2_scan includes it). */
int pafc_mamba2_finish(int dtype, int B, int L, int d_inner, const float *y0, const float *y1, const void *xbc,
                       const void *dt_raw, long ld_dt, const void *z, long ld_z, const float *dt_bias, const float *D,
                       const void *norm_weight, float eps, int diag, void *out, pafc_stream_t stream);

/* Mamba-2 selective scan (SSD), bf16 inputs, chunk-parallel on the matrix cores (csrc/mamba2_scan.hip; PARITY UNPINNED):
 *   h_t = a_t h_{t-1} + dt_t B_t x_t^T,  y_t = C_t . h_t   per head (head dim 64, state dim 128, B / C shared by the heads)
 * xbc: (B, L, ldx) bf16 rows [x (H * 64) | B (128) | C (128)] (conv1d + SiLU output); dt, log_a: (B, L, H) fp32 =
 * softplus(dt_raw + dt_bias) and dt * A (<= 0); y: (B, L, H * 64) fp32 (without the D x skip term).  workspace:
 * pafc_mamba2_scan_workspace_bytes() bytes (chunk states), chunk_len 0 = library heuristic. */
int pafc_mamba2_scan_chunk_len(int B, int L, int H);
size_t pafc_mamba2_scan_workspace_bytes(int B, int L, int H, int chunk_len);
int pafc_mamba2_scan(int B, int L, int H, const void *xbc, long ldx, const float *dt, const float *log_a, float *y,
                     int chunk_len, void *workspace, size_t workspace_bytes, pafc_stream_t stream);
/* The same scan run right-to-left on un-flipped tensors (reverse != 0: step s of the recurrence is time index L - 1 - s,
 * inputs read and y written at their own time index): the second Mamba2 of Mamba2Bidirectional without its two
 * torch.flip copies (mamba2_bidirectional.py:130-145). */
int pafc_mamba2_scan_dir(int B, int L, int H, const void *xbc, long ldx, const float *dt, const float *log_a, float *y,
                         int reverse, int chunk_len, void *workspace, size_t workspace_bytes, pafc_stream_t stream);

/* The scan as mamba_ssm's returns it: y in bf16 with the skip term inside, y = bf16(scan + D[h] x), one rounding
 * (D: float32 (H); y_bf16: (B, L, H * 64)).  Halves the scan's output bytes and leaves the block's tail to
 * pafc_mamba2_gate_norm: out = RMSNorm(y * silu(z)) * norm_weight (rows = B * L, z with row stride ld_z, all in `dtype`). */
int pafc_mamba2_scan_skip_bf16(int B, int L, int H, const void *xbc, long ldx, const float *dt, const float *log_a,
                               const float *D, void *y_bf16, int reverse, int chunk_len, void *workspace,
                               size_t workspace_bytes, pafc_stream_t stream);
int pafc_mamba2_gate_norm(int dtype, long rows, int d_inner, const void *y, const void *z, long ld_z,
                          const void *norm_weight, float eps, void *out, pafc_stream_t stream);

/* Hand-written bf16 GEMM with fused epilogue, batched (csrc/gemm_bf16.hip: 128 x 128 tiles, two blocks per CU;
 * csrc/gemm_ph.hip: persistent 256-wide phase-pipelined tiles for problems that fill the chip with them -- the entry point
 * picks):
 *   out[z][m][n] = act(alpha * sum_k A[z][m][k] * W[z][n][k] + bias[z][n] + residual[z][m][n]),   z < batch
 * A: (M, K) rows lda apart; W: (N, K) = nn.Linear.weight layout, rows ldw apart; out / residual: (M, N), rows ldo / ldr
 * apart; stride*: elements between consecutive batch entries (strideBias = 0 shares one bias); bias, residual may be
 * NULL; residual may alias out.  act: 0 none, 1 SiLU, 2 tanh, 3 ReLU, 4 GLU (out has N / 2 columns: the weight rows come
 * in blocks of 2h rows, h value rows followed by the h gate rows of the same h output channels,
 * out[m][h t + c] = a * sigmoid(b) with a, b = columns 2h t + c, 2h t + h + c; h = pafc_gemm_bf16_glu_half(M, N, K, batch)
 * = 64 or 32 by the kernel the problem goes to; N % 256 == 0 (h = 32) or N % 128 == 0 (h = 64), no residual -- F.glu after
 * pointwise_conv1, convolution.py:118-128, with the weight rows interleaved by the caller).  bias is added as given (not
 * scaled by alpha); everything is applied to the fp32 accumulator, one rounding to bf16.  N % 8 == 0, K % 64 == 0, all leading
 * dimensions / strides multiples of 8 elements, pointers 16-byte aligned.  Replaces the Linear / 1x1-Conv1d calls of
 * PositionwiseFeedForward, ConvolutionModule, RWKV_Tmix_x060c and the residual adds of ConformerEncoderLayer.forward
 * (positionwise_feed_forward.py:47-55, convolution.py:118-141, rwkv_v6/src/model.py:286-324, encoder_layer.py:201-259). */
int pafc_gemm_bf16(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W, long ldw,
                   long strideW, const void *bias, long strideBias, const void *residual, long ldr, long strideR,
                   void *out, long ldo, long strideO, float alpha, int act, pafc_stream_t stream);
int pafc_gemm_bf16_glu_half(long M, int N, int K, int batch);
/* pafc_conv3x3s2_nhwc_bf16 on the phase-pipelined kernel (implicit GEMM: K-step = one tap x 64 input channels; Ci / 64 a
 * power of two): the entry point above dispatches here for problems that fill the chip; exported for tests and A/B runs.
 * tile_m 256 / 192 / 128.  PAFC_ERR_UNSUPPORTED: take the other kernel. */
int pafc_conv3x3s2_nhwc_bf16_ph(int B, int T1, int F1, int Ci, int Co, const void *in, const void *w_tap_co_ci, const void *bias,
                                void *out, int relu, int tile_m, pafc_stream_t stream);
/* The phase-pipelined kernel by itself (A/B measurements, tests): tile_n = 256 columns, tile_m 256 / 192 / 128 / 64 rows
 * per tile; K % 128 == 0; a residual excludes an activation; GLU blocks are 64 rows (h = 32). */
int pafc_gemm_bf16_ph(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W, long ldw,
                      long strideW, const void *bias, long strideBias, const void *residual, long ldr, long strideR,
                      void *out, long ldo, long strideO, float alpha, int act, int tile_n, int tile_m, pafc_stream_t stream);
/* The same kernel with every operand form it takes -- what an fp32 MODEL needs to run its projections on the bf16 matrix
 * cores (the reference's default precision is an fp32 model around the bf16 slot: rwkv_wrapper_bidirectional.py:40-56,
 * the rwkv_do_bfloat16 key of conf/rwkv):
 *   a_split != 0: A holds an fp32 operand x as two bf16 planes [hi | lo] (M x 2K; hi = bf16(x), lo = bf16(x - hi)) and W the
 *                 matching [hi_w | hi_w | lo_w] (N x 3K): x w = hi hi_w + lo hi_w + hi lo_w, three bf16 products, fp32
 *                 accumulation, ~2^-16 relative.  K is the logical K.
 *   out_kind:     0 bf16; 1 fp32; 2 fp32 written as planes hi | lo (lo at column offset lo_off of the same row; ldo counts
 *                 bf16 elements) -- the form the next GEMM takes as its split A.
 *   res_kind:     0 none; 1 bf16 residual (out_kind 0); 2 fp32 residual (out_kind 1).
 *   bias:         bf16 for out_kind 0, fp32 otherwise; strides / leading dimensions in elements of each tensor's own type.
 * act as pafc_gemm_bf16 (GLU: h = 32); out_kind 2 takes act 0 or 1 only, out_kind 1 act 0 or GLU. */
int pafc_gemm_ph_ex(long M, int N, int K, int batch, const void *A, long lda, long strideA, int a_split, const void *W, long ldw,
                    long strideW, const void *bias, long strideBias, const void *residual, int res_kind, long ldr, long strideR,
                    void *out, int out_kind, long ldo, long lo_off, long strideO, float alpha, int act, int tile_m,
                    pafc_stream_t stream);
/* ... with the planes of a split A alternating in blocks of a_plane_block columns, [hi PB | lo PB] [hi PB | lo PB] ...
 * (PB = 64 << n, K % PB == 0; 0 = one block, the row is [hi K | lo K]): the form pafc_conv3x3s2_nhwc_split_ph writes. */
int pafc_gemm_ph_ex2(long M, int N, int K, int batch, const void *A, long lda, long strideA, int a_split, int a_plane_block,
                     const void *W, long ldw, long strideW, const void *bias, long strideBias, const void *residual, int res_kind,
                     long ldr, long strideR, void *out, int out_kind, long ldo, long lo_off, long strideO, float alpha, int act,
                     int tile_m, pafc_stream_t stream);
/* Conv2d(Ci, Co, 3, stride 2) + bias (+ ReLU) of an fp32 model (subsampling.py:187-192) on the same kernel with split
 * operands: in_planes (B, T1, F1, 2 Ci) bf16 = [hi Ci | lo Ci] per pixel of the fp32 image, w3 (9, Co, 3 Ci) = [hi | hi | lo]
 * of the fp32 weight per tap (tap-major, then Co), fp32 bias, out_planes (B, T2, F2, 2 Co) = [hi Co | lo Co] per position.
 * Ci % 128 == 0, Co % 8 == 0; tile_m 256 / 192 / 128. */
int pafc_conv3x3s2_nhwc_split_ph(int B, int T1, int F1, int Ci, int Co, const void *in_planes, const void *w3_tap_co_3ci,
                                 const float *bias, void *out_planes, int relu, int tile_m, pafc_stream_t stream);

/* Element-wise groups of the TRAINING step, one kernel forward and one backward each (csrc/train_elementwise.hip).
 *
 * Residual branch with dropout -- `x = x + dropout(y)` / `x + ff_scale * dropout(y)`, wenet/transformer/encoder_layer.py:
 * 205-206,232,247,254-255 -- replacing the framework's dropout + scale + cast + add (forward) and masked scale + scale + cast
 * (backward):   forward:  out = x + (scale / (1 - p)) * keep * y        backward: dy = (scale / (1 - p)) * keep * dout
 * (dx = dout needs no kernel).  keep = [u >= p] with u a counter-based uniform of (seed, offset, element index): the mask is
 * never stored, the backward call passes the forward call's (seed, offset).  p = 0: plain x + scale * y.
 * backward = 0: x (dtype_x), y (dtype_y) -> out (dtype_x);  backward = 1: x = dout (dtype_x) -> out = dy (dtype_y), y unused.
 * (dtype_x, dtype_y) = (f32, bf16) [fp32 stream, bf16 branch: autocast], (f32, f32), (bf16, bf16).  n % 8 == 0, 16-byte
 * aligned pointers, contiguous tensors. */
int pafc_residual_dropout(int backward, int dtype_x, int dtype_y, long n, const void *x, const void *y, void *out, float scale,
                          float p, unsigned long long seed, unsigned long long offset, pafc_stream_t stream);
/* `dropout(activation(h))` of the feed-forward module with activation SiLU (positionwise_feed_forward.py:47-55):
 *   forward:  out = bf16(silu(h)) * keep / (1 - p)        backward: dh = dout * keep / (1 - p) * silu'(h)
 * h, dout, out in `dtype` (f32 or bf16); same mask generator; n % 8 == 0. */
int pafc_silu_dropout(int backward, int dtype, long n, const void *h, const void *dout, void *out, float p,
                      unsigned long long seed, unsigned long long offset, pafc_stream_t stream);

/* bf16 transposed copies of many matrices in ONE launch -- the training step's weights in (K, N) layout, so that the input
 * gradient dX = dY W of every nn.Linear (autograd through positionwise_feed_forward.py:47-55, convolution.py:118-141,
 * src/model.py:286-324 under train_utils.py:609-729) is the forward GEMM pafc_gemm_bf16(dY, W^T).  `table` points to n
 * descriptors of 32 bytes in DEVICE memory: { const void *src; void *dst; int rows, cols, src_f32, tile0; } -- src (rows, cols)
 * row-major fp32 (src_f32 = 1) or bf16, dst (cols, rows) row-major bf16, tile0 = the running sum of
 * ceil(rows / 64) * ceil(cols / 64) over the earlier descriptors; total_tiles = that sum over all n.  Sources and destinations
 * must not overlap; nothing is allocated, the launch is asynchronous on `stream`. */
int pafc_multi_transpose_bf16(const void *table, int n, int total_tiles, pafc_stream_t stream);
/* The same descriptor table without the transposition: dst (bf16) = src (fp32 | bf16) element for element, rows x cols elements per
 * tensor, tile0 counting chunks of 4 096 elements: the bf16 copies of the fp32 master weights, refreshed once per training step. */
int pafc_multi_cast_bf16(const void *table, int n, int total_chunks, pafc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PAFC_ENCODER_OPS_H */
