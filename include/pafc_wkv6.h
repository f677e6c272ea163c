/*
 * pafc_wkv6.h -- C ABI of the MI355X (gfx950) WKV-6 recurrence.
 *
 * This is the drop-in boundary for the reference's only native code,
 * wenet/rwkv_v6/cuda/{wkv6_op.cpp,wkv6_cuda.cu,wkv6state_op.cpp,wkv6state_cuda.cu}.
 * Every entry point takes plain device pointers, sizes and a hipStream_t, owns
 * no memory, keeps no global state, is asynchronous on the given stream and
 * returns 0 or a negative PAFC_ERR_* instead of asserting (the reference's
 * launchers only `assert`, wkv6_cuda.cu:267-268).
 *
 * Tensor layouts are the reference's (wkv6_op.cpp:9-33, src/model.py:108-133):
 *   r, k, v, w, y, gy, gr, gk, gv, gw : (B, T, C) contiguous, C = H * N
 *   u                                 : (H, N)
 *   gu                                : (B, C) per-batch partial (summed over B by the caller, model.py:151)
 *   state s                           : float32 (B, H, N, N) indexed [b][h][i = value][j = key]
 *                                       (wkv6state_cuda.cu:15,23-25)
 * N must be 64 (the only head size the paper's configs use, the conf/rwkv YAML files, lines 5-6).
 * Unlike the reference there is no compile-time _T_ limit on T.
 *
 * How T-parallelism works (what `workspace` is for): T is cut into chunks of
 * `chunk_len` steps; pass A computes every chunk's local end state from zero
 * and its cumulative decay, pass B scans the (decay, state) pairs over chunks
 * -- S <- diag(d) S + k v^T is associative -- and pass C replays every chunk
 * from its true incoming state and writes y.  Only multiplications by decays
 * in (0,1) occur, never divisions, so nothing under/overflows.  With
 * workspace == NULL (or chunk_len >= T) the kernel runs one chunk per (b,h):
 * the reference's serial schedule.
 */
#ifndef PAFC_WKV6_H
#define PAFC_WKV6_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pafc_stream_t; /* hipStream_t */

enum {
    PAFC_OK = 0,
    PAFC_ERR_NULL_POINTER = -1,
    PAFC_ERR_BAD_DIMS = -2,      /* B, T, C, H <= 0 or C != H * N */
    PAFC_ERR_HEAD_SIZE = -3,     /* N != 64 */
    PAFC_ERR_WORKSPACE = -4,     /* workspace too small for the requested chunk_len */
    PAFC_ERR_LAUNCH = -5,        /* hipLaunchKernel failed */
    PAFC_ERR_DTYPE = -6,
    PAFC_ERR_UNSUPPORTED = -7,
    PAFC_ERR_ALIGNMENT = -8      /* r, k, v, w, y must be 16-byte aligned (torch allocations and (B,T,C) views are) */
};

enum { PAFC_F32 = 0, PAFC_BF16 = 1,
       PAFC_SPLIT_BF16 = 2 /* OUTPUT form only: an fp32 value as two bf16 planes, hi = bf16(x) | lo = bf16(x - hi) */ };

/* Library/ABI version, bumped when a signature changes. */
int pafc_abi_version(void);

/* Hardware self-check of the in-row lane permutations the matrix-core kernel relies on (DPP control codes): writes,
 * for x = lane + 1, 4 results per lane [lane^1, lane^2, mirror in the 8-lane half (i -> 7-i), mirror in the 16-lane
 * row (i -> 15-i)] into out[0:256], and the exchanges between the four 16-lane rows (v_permlane16_swap / v_permlane32_swap with
 * both operands x): [x at my position in the even row of my row pair, in the odd row of my pair, in the lower 32 lanes, in the
 * upper 32 lanes] into out[256:512] (device, 512 floats); tests compare them with the definitions. */
int pafc_selftest_lane_ops(float *out_2x64x4, pafc_stream_t stream);

/* ---- workspace sizing -------------------------------------------------------------------- */
/* Chunk length the library would pick for this shape (fills the 256 CUs; returns T when the
 * B*H*ndir sequences alone already do, in which case no workspace is needed). */
int pafc_wkv6_pick_chunk_len(int B, int T, int C, int H, int ndir);
/* Bytes of scratch the chunked schedule needs (0 when chunk_len >= T). */
size_t pafc_wkv6_fwd_workspace_bytes(int B, int T, int C, int H, int ndir, int chunk_len);

/* ---- forward, one direction -------------------------------------------------------------- */
/* Replaces torch.ops.wkv6.forward / forward_fp32 (wkv6_op.cpp:9-11,18-20; kernel_forward
 * wkv6_cuda.cu:8-63).  chunk_len <= 0 lets the library choose. */
int pafc_wkv6_forward_bf16(int B, int T, int C, int H, const void *r, const void *k, const void *v,
                           const void *w, const void *u, void *y, int chunk_len, void *workspace,
                           size_t workspace_bytes, pafc_stream_t stream);
int pafc_wkv6_forward_f32(int B, int T, int C, int H, const void *r, const void *k, const void *v,
                          const void *w, const void *u, void *y, int chunk_len, void *workspace,
                          size_t workspace_bytes, pafc_stream_t stream);

/* ---- forward with recurrent-state carry and direction ------------------------------------ */
/* Spec: wkv6state kernel_forward (wkv6state_cuda.cu:6-65) for s_in; s_out is the final state, which the
 * reference never writes back.  Either may be NULL (zeros / not wanted).  reverse != 0 walks t = T-1..0,
 * which is what the bidirectional wrapper gets by flipping its input and output
 * (rwkv_wrapper_bidirectional.py:44-48) -- here without the two flip copies. */
int pafc_wkv6_forward_state(int dtype, int B, int T, int C, int H, const void *r, const void *k,
                            const void *v, const void *w, const void *u, void *y, const float *s_in,
                            float *s_out, int reverse, int chunk_len, void *workspace,
                            size_t workspace_bytes, pafc_stream_t stream);

/* ---- forward, both directions in one launch ---------------------------------------------- */
/* The left-to-right pass over (r_f..y_f) and the right-to-left pass over (r_b..y_b) of
 * RWKV_TmixWrapper_bidirectional.forward (rwkv_wrapper_bidirectional.py:44-49): two parameter sets, so
 * two sets of projections, one grid.  The R2L pass runs over the whole padded length T exactly like the
 * reference's flip does. */
int pafc_wkv6_forward_bidir(int dtype, int B, int T, int C, int H, const void *r_f, const void *k_f,
                            const void *v_f, const void *w_f, const void *u_f, void *y_f,
                            const void *r_b, const void *k_b, const void *v_b, const void *w_b,
                            const void *u_b, void *y_b, int chunk_len, void *workspace,
                            size_t workspace_bytes, pafc_stream_t stream);

/* As above with the decay bias fused: the scan uses w + wb (rounded to the element type, as the reference's
 * `self.time_decay + lora` does, src/model.py:289), wb_f / wb_b: (H, N) or NULL.  Saves one (B,T,C) pass per direction. */
int pafc_wkv6_forward_bidir_wbias(int dtype, int B, int T, int C, int H, const void *r_f, const void *k_f,
                                  const void *v_f, const void *w_f, const void *u_f, const void *wb_f, void *y_f,
                                  const void *r_b, const void *k_b, const void *v_b, const void *w_b, const void *u_b,
                                  const void *wb_b, void *y_b, int chunk_len, void *workspace, size_t workspace_bytes,
                                  pafc_stream_t stream);

/* ---- backward ---------------------------------------------------------------------------- */
/* Replaces torch.ops.wkv6.backward / backward_fp32 (wkv6_op.cpp:12-14,21-23; kernel_backward_101/102/103/201
 * wkv6_cuda.cu:65-263).  gu is the (B, C) per-batch partial.  reverse as above. */
size_t pafc_wkv6_bwd_workspace_bytes(int B, int T, int C, int H, int chunk_len);
int pafc_wkv6_backward(int dtype, int B, int T, int C, int H, const void *r, const void *k, const void *v,
                       const void *w, const void *u, const void *gy, void *gr, void *gk, void *gv,
                       void *gw, void *gu, int reverse, int chunk_len, void *workspace,
                       size_t workspace_bytes, pafc_stream_t stream);

/* The same with an initial state and its gradient: replaces torch.ops.wkv6state.backward (kernel_backward_111/222,
 * wkv6state_cuda.cu:66-296; WKV_6STATE src/model.py:54-103, which the reference builds but never runs).
 *   s_in: float32 (B, H, N, N) [value i][key j] -- the layout of pafc_wkv6_forward_state -- or NULL (zero state);
 *   gs:   float32 (B, H, N, N), same layout, dL/ds_in per batch entry, or NULL when not wanted (the reference sums it
 *         over the batch on the host, model.py:100).
 * With a state gw at the first step is no longer zero (the state it decays is not); gw at the last step stays zero. */
int pafc_wkv6_backward_state(int dtype, int B, int T, int C, int H, const void *r, const void *k, const void *v,
                             const void *w, const void *u, const float *s_in, const void *gy, void *gr, void *gk, void *gv,
                             void *gw, void *gu, float *gs, int reverse, int chunk_len, void *workspace,
                             size_t workspace_bytes, pafc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PAFC_WKV6_H */
