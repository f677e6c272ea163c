/*
 * pafc_search.h -- C ABI of the GPU-resident part of the decode step.
 *
 * CTC greedy search of the reference (wenet/transformer/search.py:106-121 + remove_duplicates_and_blank,
 * wenet/utils/ctc_utils.py:22-32): argmax over the vocabulary per frame, frames beyond the utterance length count as
 * blank, runs of equal ids collapse to one, blanks are dropped.  There: topk on the device, a (B, T) copy to the
 * host and a Python loop per frame.  Here: two kernels; only the collapsed token lists leave the device.
 * Conventions as in pafc_wkv6.h.
 */
#ifndef PAFC_SEARCH_H
#define PAFC_SEARCH_H

#include "pafc_wkv6.h"

#ifdef __cplusplus
extern "C" {
#endif

/* scores: (B, T, V) contiguous, PAFC_F32 or PAFC_BF16 -- log-probabilities or logits (same argmax).
 * lens: (B) int64 valid frames per utterance, or NULL (all T frames valid).
 * best: (B, T) int32 scratch that receives the per-frame argmax (ties: lowest index, like torch.topk / argmax;
 *       padded frames: blank_id).
 * tokens: (B, T) int32, row b holds ntok[b] collapsed token ids; ntok: (B) int32.
 * frames: (B, T) int32 or NULL: frame index of the FIRST frame of each emitted token (for time stamps). */
int pafc_ctc_greedy(int dtype, int B, int T, int V, const void *scores, const int64_t *lens, int blank_id,
                    int32_t *best, int32_t *tokens, int32_t *ntok, int32_t *frames, pafc_stream_t stream);

/* Row-wise log-softmax over the vocabulary: out[r][v] = x[r][v] - max_v x[r] - log sum_v exp(x[r][v] - max), fp32
 * arithmetic, one rounding to the element type (CTC.log_softmax, wenet/transformer/ctc.py:106-114, behind
 * ASRModel.ctc_logprobs, asr_model.py:324-335).  x, out: (rows, V) contiguous, may alias.  One pass over HBM: a wave
 * keeps its row in registers (V <= 8192 elements in bf16, 4096 in fp32; longer rows are re-read from cache). */
int pafc_log_softmax_rows(int dtype, long rows, int V, const void *x, void *out, pafc_stream_t stream);

/* CTC prefix beam search (ctc_prefix_beam_search, wenet/transformer/search.py:124-248, without context graph and time
 * stamps), one wave per utterance, frames walked on the device.
 * top_logp / top_idx: (B, T, K) the K best log-probabilities and token ids per frame, best first (torch.topk of the CTC
 *   log-probs, as the reference takes them per frame); K <= 16, beam <= 16.  lens: (B) int64 valid frames, or NULL.
 * out_tokens: (B, beam, T) int32, entry (b, n) holds out_len[b][n] ids of the n-th best prefix (best first);
 * out_len: (B, beam) int32, -1 for unused entries (fewer prefixes than beam); out_score: (B, beam) float64 total
 * log-probabilities (arithmetic in float64 like the reference's Python floats).
 * workspace: pafc_ctc_prefix_beam_workspace_bytes(B, T, beam) bytes (the per-utterance prefix tries). */
size_t pafc_ctc_prefix_beam_workspace_bytes(int B, int T, int beam);
int pafc_ctc_prefix_beam_search(int B, int T, int K, const float *top_logp, const int32_t *top_idx, const int64_t *lens,
                                int beam, int blank_id, int32_t *out_tokens, int32_t *out_len, double *out_score,
                                void *workspace, size_t workspace_bytes, pafc_stream_t stream);

/* CTC-fused RNN-T prefix beam search (PrefixBeamSearch.prefix_beam_search_decode_batch,
 * wenet/transducer/search/prefix_beam_search.py:428-574): the per-frame candidate walk on the device.  The caller keeps
 * B x beam fixed slots; per frame it runs predictor step + joint + log-softmax + fusion + top-`beam` for all slots with
 * framework ops and hands the (B, beam, beam) values / token ids to pafc_rnnt_beam_step, which updates the beams in
 * `workspace` and returns, per slot, where the survivor's LSTM state comes from -- next_idx[slot] indexes the
 * concatenation [old states (B*beam) | new states (B*beam)] -- and the token to feed the predictor next (last_tok).
 * No host synchronisation between frames.  beam <= 16.  lens: (B) int64 valid frames or NULL.  t_dev (device int64, or
 * NULL): when given, the frame index is read from it instead of `t`, so the whole frame body can be captured once in a
 * hipGraph and replayed (the caller increments it); frames t >= T are no-ops.
 * pafc_rnnt_beam_finish writes the n-best lists like pafc_ctc_prefix_beam_search (the leading blank is not included). */
size_t pafc_rnnt_beam_workspace_bytes(int B, int T, int beam);
int pafc_rnnt_beam_init(int B, int T, int beam, int blank_id, void *workspace, size_t workspace_bytes, int64_t *next_idx,
                        int64_t *last_tok, pafc_stream_t stream);
int pafc_rnnt_beam_step(int B, int T, int beam, int blank_id, int t, const int64_t *t_dev, const int64_t *lens,
                        const float *top_val, const int64_t *top_idx, void *workspace, size_t workspace_bytes,
                        int64_t *next_idx, int64_t *last_tok, pafc_stream_t stream);
int pafc_rnnt_beam_finish(int B, int T, int beam, void *workspace, size_t workspace_bytes, int32_t *out_tokens,
                          int32_t *out_len, double *out_score, pafc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
