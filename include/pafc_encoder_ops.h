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

#ifdef __cplusplus
}
#endif
#endif /* PAFC_ENCODER_OPS_H */
