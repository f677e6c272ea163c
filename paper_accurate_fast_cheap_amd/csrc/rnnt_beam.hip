// Bookkeeping of the CTC-fused RNN-T prefix beam search on the GPU (C ABI: include/pafc_search.h: pafc_rnnt_beam_*).
//
// Reference: PrefixBeamSearch.prefix_beam_search_decode_batch, wenet/transducer/search/prefix_beam_search.py:428-574.  Per
// frame and utterance the reference sorts the beam x beam candidates (beam score + fused log-prob of the top tokens),
// walks them best first, merges candidates that spell the same hypothesis with log_add, stops as soon as `beam`
// distinct hypotheses are collected, sorts those and keeps them -- reading every candidate with .item().  Here the
// predictor / joint / fusion / top-k stay batched framework ops on fixed (B x beam) slots and this kernel does the walk
// on the device: no host synchronisation between frames.  It also emits, per slot, which LSTM state the survivor
// carries (old state of its parent beam for a blank, new state for an emitted token) and the token to feed next.
//
// Hypotheses are nodes of a per-utterance trie (as in ctc_beam.hip): a blank keeps the parent's node, a token either
// lands on a beam member that already spells parent + token or becomes a new node.  Scores are float64 like the
// reference's Python floats; each candidate's score is float32(beam score) + float32 log-prob added in float32, the
// rounding point of the reference (`torch.tensor(scores) + top_k_logp`).
#include "pafc_common.h"
#include "../../include/pafc_search.h"

namespace pafc {
namespace {

constexpr int RB = 16;                      // beam limit
constexpr double RNEG_INF = -__builtin_huge_val();

__device__ __forceinline__ double rlog_add2(double a, double b) {
    if (a == RNEG_INF && b == RNEG_INF) return RNEG_INF;
    const double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

struct RnntState {
    int32_t *nb;                                   // (B) live beams
    int32_t *node, *parent, *last;                 // (B, beam)
    double *score;                                 // (B, beam)
    int32_t *pool_parent, *pool_token;             // (B, 1 + T * beam)
};

__device__ __forceinline__ RnntState carve(void *ws, int B, int T, int beam) {
    RnntState s;
    char *p = (char *)ws;
    s.score = (double *)p; p += sizeof(double) * (size_t)B * beam;
    s.nb = (int32_t *)p; p += sizeof(int32_t) * (size_t)((B + 1) & ~1);
    s.node = (int32_t *)p; p += sizeof(int32_t) * (size_t)B * beam;
    s.parent = (int32_t *)p; p += sizeof(int32_t) * (size_t)B * beam;
    s.last = (int32_t *)p; p += sizeof(int32_t) * (size_t)B * beam;
    s.pool_parent = (int32_t *)p; p += sizeof(int32_t) * (size_t)B * (1 + (size_t)T * beam);
    s.pool_token = (int32_t *)p;
    return s;
}

__global__ void rnnt_beam_init_kernel(void *ws, int B, int T, int beam, int blank, int64_t *next_idx, int64_t *last_tok) {
    const RnntState s = carve(ws, B, T, beam);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * beam) return;
    const int b = i / beam, m = i % beam;
    s.node[i] = 0; s.parent[i] = -1; s.last[i] = blank;
    s.score[i] = m == 0 ? 0.0 : RNEG_INF;
    next_idx[i] = i;
    last_tok[i] = blank;
    if (m == 0) {
        s.nb[b] = 1;
        s.pool_parent[(size_t)b * (1 + (size_t)T * beam)] = -1;
        s.pool_token[(size_t)b * (1 + (size_t)T * beam)] = blank;
    }
}

__global__ __launch_bounds__(64) void rnnt_beam_step_kernel(void *ws, int B, int T, int beam, int blank, int t_host,
                                                            const int64_t *t_dev, const int64_t *lens,
                                                            const float *top_val, const int64_t *top_idx,
                                                            int64_t *next_idx, int64_t *last_tok) {
    // the frame index comes from device memory when the frame body is replayed from a captured graph
    const int t = t_dev != nullptr ? (int)*t_dev : t_host;
    __shared__ float c_val[RB * RB];
    __shared__ int c_tok[RB * RB], c_order[RB * RB];
    __shared__ int m_node[RB], m_parent[RB], m_last[RB];
    __shared__ double m_score[RB];
    // collected hypotheses (beam_A of the reference), in first-seen order
    __shared__ double a_score[RB];
    __shared__ int a_node[RB], a_parent[RB], a_tok[RB], a_last[RB], a_src[RB], a_new[RB], a_rank[RB];
    __shared__ int a_count;

    const RnntState s = carve(ws, B, T, beam);
    const int b = blockIdx.x, lane = threadIdx.x;
    const int base = b * beam;
    if (t >= T || (lens != nullptr && t >= lens[b])) {   // finished utterance: every slot keeps its state and its beam
        if (lane < beam) next_idx[base + lane] = base + lane;
        return;
    }
    const int nbm = s.nb[b];
    if (lane < beam) {
        m_node[lane] = s.node[base + lane]; m_parent[lane] = s.parent[base + lane]; m_last[lane] = s.last[base + lane];
        m_score[lane] = s.score[base + lane];
    }
    __syncthreads();
    const int ncand = nbm * beam;
    for (int c = lane; c < ncand; c += 64) {
        const int m = c / beam, k = c % beam;
        // float32(beam score) + float32 log-prob, added in float32 (prefix_beam_search.py:515-520)
        c_val[c] = (float)m_score[m] + top_val[((size_t)base + m) * beam + k];
        c_tok[c] = (int)top_idx[((size_t)base + m) * beam + k];
    }
    __syncthreads();
    for (int c = lane; c < ncand; c += 64) {        // descending by value; equal values keep their flat order
        const float v = c_val[c];
        int r = 0;
        for (int j = 0; j < ncand; ++j) r += (c_val[j] > v || (c_val[j] == v && j < c)) ? 1 : 0;
        c_order[r] = c;
    }
    __syncthreads();
    if (lane == 0) {
        int cnt = 0;
        for (int r = 0; r < ncand && cnt < beam; ++r) {
            const int c = c_order[r], m = c / beam, tk = c_tok[c];
            const double v = (double)c_val[c];
            int node = -1, par = -1, tok = -1, lastt;
            if (tk == blank) {
                node = m_node[m]; par = m_parent[m]; tok = m_last[m]; lastt = m_last[m];
            } else {
                par = m_node[m]; tok = tk; lastt = tk;
                for (int qm = 0; qm < nbm; ++qm)
                    if (m_parent[qm] == par && m_last[qm] == tk && m_node[qm] != 0) node = m_node[qm];   // already a member
            }
            int hit = -1;
            for (int e = 0; e < cnt; ++e) {
                const bool same = node >= 0 ? a_node[e] == node : (a_node[e] < 0 && a_parent[e] == par && a_tok[e] == tok);
                if (same) { hit = e; break; }
            }
            if (hit >= 0) {
                a_score[hit] = rlog_add2(a_score[hit], v);
            } else {
                a_score[cnt] = v; a_node[cnt] = node; a_parent[cnt] = par; a_tok[cnt] = tok; a_last[cnt] = lastt;
                a_src[cnt] = m; a_new[cnt] = tk != blank;
                ++cnt;
            }
        }
        a_count = cnt;
    }
    __syncthreads();
    const int cnt = a_count;
    if (lane < cnt) {                               // stable sort by score, descending (Python's list.sort, :556)
        int r = 0;
        for (int e = 0; e < cnt; ++e) r += (a_score[e] > a_score[lane] || (a_score[e] == a_score[lane] && e < lane)) ? 1 : 0;
        a_rank[lane] = r;
    }
    __syncthreads();
    const size_t pstride = 1 + (size_t)T * beam;
    if (lane < cnt) {
        const int p = a_rank[lane];
        int node = a_node[lane];
        if (node < 0) {
            node = 1 + t * beam + p;
            s.pool_parent[b * pstride + node] = a_parent[lane];
            s.pool_token[b * pstride + node] = a_tok[lane];
        }
        s.node[base + p] = node;
        s.parent[base + p] = a_node[lane] < 0 ? a_parent[lane] : (a_new[lane] ? a_parent[lane] : m_parent[a_src[lane]]);
        s.last[base + p] = a_last[lane];
        s.score[base + p] = a_score[lane];
        next_idx[base + p] = (int64_t)(base + a_src[lane]) + (a_new[lane] ? (int64_t)B * beam : 0);
        last_tok[base + p] = a_last[lane];
    } else if (lane < beam) {                       // unused slot: inert
        s.score[base + lane] = RNEG_INF; s.node[base + lane] = 0; s.parent[base + lane] = -1; s.last[base + lane] = blank;
        next_idx[base + lane] = base + lane;
        last_tok[base + lane] = blank;
    }
    if (lane == 0) s.nb[b] = cnt;
}

__global__ __launch_bounds__(64) void rnnt_beam_finish_kernel(void *ws, int B, int T, int beam, int32_t *out_tokens,
                                                              int32_t *out_len, double *out_score) {
    const RnntState s = carve(ws, B, T, beam);
    const int b = blockIdx.x, lane = threadIdx.x;
    if (lane >= beam) return;
    const size_t pstride = 1 + (size_t)T * beam;
    const int32_t *pp = s.pool_parent + b * pstride, *pt = s.pool_token + b * pstride;
    int32_t *ot = out_tokens + ((size_t)b * beam + lane) * T;
    if (lane < s.nb[b]) {
        int len = 0;
        for (int n = s.node[b * beam + lane]; n > 0; n = pp[n]) ++len;
        int pos = len;
        for (int n = s.node[b * beam + lane]; n > 0; n = pp[n]) ot[--pos] = pt[n];
        out_len[b * beam + lane] = len;
        out_score[b * beam + lane] = s.score[b * beam + lane];
    } else {
        out_len[b * beam + lane] = -1;
        out_score[b * beam + lane] = RNEG_INF;
    }
}

}  // namespace
}  // namespace pafc

extern "C" size_t pafc_rnnt_beam_workspace_bytes(int B, int T, int beam) {
    if (B <= 0 || T <= 0 || beam <= 0) return 0;
    return sizeof(double) * (size_t)B * beam + sizeof(int32_t) * ((size_t)((B + 1) & ~1) + 3 * (size_t)B * beam +
                                                                   2 * (size_t)B * (1 + (size_t)T * beam));
}

static int rnnt_check(int B, int T, int beam, const void *ws, size_t ws_bytes) {
    if (!ws) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || beam <= 0) return PAFC_ERR_BAD_DIMS;
    if (beam > pafc::RB) return PAFC_ERR_UNSUPPORTED;
    if (ws_bytes < pafc_rnnt_beam_workspace_bytes(B, T, beam)) return PAFC_ERR_WORKSPACE;
    return PAFC_OK;
}

extern "C" int pafc_rnnt_beam_init(int B, int T, int beam, int blank_id, void *workspace, size_t workspace_bytes,
                                   int64_t *next_idx, int64_t *last_tok, pafc_stream_t stream) {
    const int rc = rnnt_check(B, T, beam, workspace, workspace_bytes);
    if (rc) return rc;
    if (!next_idx || !last_tok) return PAFC_ERR_NULL_POINTER;
    hipLaunchKernelGGL(pafc::rnnt_beam_init_kernel, dim3((B * beam + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace,
                       B, T, beam, blank_id, next_idx, last_tok);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_rnnt_beam_step(int B, int T, int beam, int blank_id, int t, const int64_t *t_dev, const int64_t *lens,
                                   const float *top_val, const int64_t *top_idx, void *workspace, size_t workspace_bytes,
                                   int64_t *next_idx, int64_t *last_tok, pafc_stream_t stream) {
    const int rc = rnnt_check(B, T, beam, workspace, workspace_bytes);
    if (rc) return rc;
    if (!top_val || !top_idx || !next_idx || !last_tok) return PAFC_ERR_NULL_POINTER;
    if (!t_dev && (t < 0 || t >= T)) return PAFC_ERR_BAD_DIMS;
    hipLaunchKernelGGL(pafc::rnnt_beam_step_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, workspace, B, T, beam, blank_id, t,
                       t_dev, lens, top_val, top_idx, next_idx, last_tok);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_rnnt_beam_finish(int B, int T, int beam, void *workspace, size_t workspace_bytes, int32_t *out_tokens,
                                     int32_t *out_len, double *out_score, pafc_stream_t stream) {
    const int rc = rnnt_check(B, T, beam, workspace, workspace_bytes);
    if (rc) return rc;
    if (!out_tokens || !out_len || !out_score) return PAFC_ERR_NULL_POINTER;
    hipLaunchKernelGGL(pafc::rnnt_beam_finish_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, workspace, B, T, beam, out_tokens,
                       out_len, out_score);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
