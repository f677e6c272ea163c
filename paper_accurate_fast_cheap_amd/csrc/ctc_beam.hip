// CTC prefix beam search on the GPU (C ABI: include/pafc_search.h: pafc_ctc_prefix_beam_search).
//
// Reference: ctc_prefix_beam_search, wenet/transformer/search.py:124-248 (without context graph and time stamps): per
// frame the top-`beam` tokens extend / repeat / blank the current prefixes, equal prefixes merge by log-add, the best
// `beam` survive.  There it is a Python loop per utterance, per frame, per candidate with .item() syncs; here one wave
// per utterance walks the frames on the device and only the n-best lists come back.
//
// Prefixes are nodes of a per-utterance trie (parent, token); a frame's candidates are "slots":
//   S_b        the beam member b itself (blank, or its last token again)           -> one slot per member
//   E_{b,r}    member b extended by the r-th best token u                          -> unless that prefix already IS a
//              member Q (parent(Q) = b, token(Q) = u): then the contribution goes to S_Q (the reference's dict key)
// Every slot receives at most two contributions (one blank-ending, one or two non-blank-ending), so each slot GATHERS
// its own instead of the reference's sequential scatter: log-add of two numbers is commutative, a first log-add with
// -inf returns the other argument exactly, hence the values are those of the reference's loop.  Ties of the total score
// are broken like Python's stable sort over the dict's insertion order: the position of the slot's first touch in
// the reference's (token rank, member rank) loop nest.  Arithmetic is float64 like the reference's Python floats; exp
// and log come from the device math library, so scores agree to the last few ulps, token lists exactly.
#include "pafc_common.h"
#include "../../include/pafc_search.h"

namespace pafc {
namespace {

constexpr int MAXB = 16;                   // beam size and top-k limit
constexpr int NSLOT = MAXB + MAXB * MAXB;  // S slots then E slots
constexpr double NEG_INF = -__builtin_huge_val();

__device__ __forceinline__ double log_add2(double a, double b) {
    if (a == NEG_INF && b == NEG_INF) return NEG_INF;
    const double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

struct BeamParams {
    int T, K, beam, blank;
    const float *top_logp;     // (B, T, K)
    const int32_t *top_idx;    // (B, T, K)
    const int64_t *lens;       // (B) or null
    int32_t *pool_parent;      // (B, 1 + T * beam)
    int32_t *pool_token;       // (B, 1 + T * beam)
    int32_t *out_tokens;       // (B, beam, T)
    int32_t *out_len;          // (B, beam)   -1 for unused entries
    double *out_score;         // (B, beam)
};

__global__ __launch_bounds__(64) void ctc_prefix_beam_kernel(const BeamParams p) {
    __shared__ double c_s[MAXB], c_ns[MAXB], c_sc[MAXB];          // current beam: blank-ending, non-blank-ending, total
    __shared__ int c_node[MAXB], c_last[MAXB], c_parent[MAXB];
    __shared__ double s_s[NSLOT], s_ns[NSLOT], s_tot[NSLOT];
    __shared__ int s_order[NSLOT], s_node[NSLOT], s_tok[NSLOT], s_par[NSLOT];
    __shared__ int tok[MAXB];
    __shared__ double lp[MAXB];
    __shared__ int n_node[MAXB], n_last[MAXB], n_parent[MAXB];              // next beam staging
    __shared__ double n_bs[MAXB], n_bns[MAXB], n_bsc[MAXB];
    __shared__ int s_nb;

    const int b = blockIdx.x, lane = threadIdx.x;
    const int T = p.lens ? (int)min((int64_t)p.T, p.lens[b]) : p.T;
    const int K = p.K, beam = p.beam;
    const long pool_stride = 1 + (long)p.T * beam;
    int32_t *pparent = p.pool_parent + b * pool_stride, *ptoken = p.pool_token + b * pool_stride;
    constexpr int UNTOUCHED = 0x7fffffff;

    if (lane == 0) {
        c_node[0] = 0; c_last[0] = -1; c_parent[0] = -1; c_s[0] = 0.0; c_ns[0] = NEG_INF; c_sc[0] = 0.0;
        pparent[0] = -1; ptoken[0] = -1;
        s_nb = 1;
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int nb = s_nb;
        if (lane < K) {
            tok[lane] = p.top_idx[((long)b * p.T + t) * K + lane];
            lp[lane] = (double)p.top_logp[((long)b * p.T + t) * K + lane];
        }
        __syncthreads();
        // rank of the blank token in the top-k (or -1)
        int rblank = -1;
        for (int r = 0; r < K; ++r) if (tok[r] == p.blank) rblank = r;

        // ---- S slots: lane m < nb gathers what lands on member m itself ---------------------------------------
        if (lane < nb) {
            const int m = lane;
            double s = NEG_INF, ns = NEG_INF;
            int order = UNTOUCHED;
            if (rblank >= 0) { s = c_sc[m] + lp[rblank]; order = min(order, (rblank * nb + m) * 2); }
            int rq = -1;                                            // rank of the member's own last token
            if (c_last[m] >= 0) for (int r = 0; r < K; ++r) if (tok[r] == c_last[m]) rq = r;
            if (rq >= 0 && c_last[m] != p.blank) {
                ns = c_ns[m] + lp[rq];                             // *uu -> *u
                order = min(order, (rq * nb + m) * 2);
                // the same prefix reached by extending its parent, if the parent is in the beam too
                for (int pb = 0; pb < nb; ++pb) {
                    if (c_node[pb] == c_parent[m]) {
                        const bool rep = c_last[pb] == c_last[m];   // parent ends in the same token: only its blank path
                        ns = log_add2(ns, (rep ? c_s[pb] : c_sc[pb]) + lp[rq]);
                        order = min(order, (rq * nb + pb) * 2 + (rep ? 1 : 0));
                    }
                }
            }
            s_s[m] = s; s_ns[m] = ns; s_order[m] = order; s_node[m] = c_node[m]; s_tok[m] = c_last[m]; s_par[m] = c_parent[m];
        } else if (lane < MAXB) {
            s_order[lane] = UNTOUCHED;
        }
        // ---- E slots: (member m, token rank r) -> a new prefix, unless it already is a member ---------------
        for (int e = lane; e < MAXB * MAXB; e += 64) {
            const int m = e / MAXB, r = e % MAXB;
            int order = UNTOUCHED;
            double ns = NEG_INF;
            if (m < nb && r < K && tok[r] != p.blank) {
                bool is_member = false;
                for (int qm = 0; qm < nb; ++qm) is_member |= (c_parent[qm] == c_node[m] && c_last[qm] == tok[r]);
                if (!is_member) {
                    const bool rep = tok[r] == c_last[m];
                    ns = (rep ? c_s[m] : c_sc[m]) + lp[r];
                    order = (r * nb + m) * 2 + (rep ? 1 : 0);
                }
            }
            const int si = MAXB + e;
            s_s[si] = NEG_INF; s_ns[si] = ns; s_order[si] = order; s_node[si] = -1;
            s_tok[si] = (r < K) ? tok[r] : -1; s_par[si] = (m < nb) ? c_node[m] : -1;
        }
        __syncthreads();
        for (int i = lane; i < NSLOT; i += 64) s_tot[i] = s_order[i] == UNTOUCHED ? NEG_INF : log_add2(s_s[i], s_ns[i]);
        __syncthreads();
        // ---- rank the touched slots: score descending, first-touch order ascending -----------------------------
        for (int i = lane; i < NSLOT; i += 64) {
            if (s_order[i] == UNTOUCHED) continue;
            int rank = 0;
            const double sc = s_tot[i];
            const int oi = s_order[i];
            for (int j = 0; j < NSLOT; ++j) {
                if (s_order[j] == UNTOUCHED) continue;
                rank += (s_tot[j] > sc || (s_tot[j] == sc && s_order[j] < oi)) ? 1 : 0;
            }
            if (rank < beam) {
                int node = s_node[i];
                if (node < 0) {                                     // a new prefix: its node id is fixed by (t, rank)
                    node = 1 + t * beam + rank;
                    pparent[node] = s_par[i];
                    ptoken[node] = s_tok[i];
                }
                n_node[rank] = node; n_last[rank] = s_tok[i]; n_parent[rank] = s_par[i];
                n_bs[rank] = s_s[i]; n_bns[rank] = s_ns[i]; n_bsc[rank] = sc;
            }
        }
        __syncthreads();
        if (lane == 0) {
            int cnt = 0;
            for (int i = 0; i < NSLOT; ++i) cnt += s_order[i] != UNTOUCHED;
            s_nb = min(cnt, beam);
        }
        __syncthreads();
        if (lane < s_nb) {
            c_node[lane] = n_node[lane]; c_last[lane] = n_last[lane]; c_parent[lane] = n_parent[lane];
            c_s[lane] = n_bs[lane]; c_ns[lane] = n_bns[lane]; c_sc[lane] = n_bsc[lane];
        }
        __syncthreads();
    }

    // ---- n-best lists: walk the trie back from each surviving node ------------------------------------------
    const int nb = s_nb;
    if (lane < beam) {
        int32_t *ot = p.out_tokens + ((long)b * beam + lane) * p.T;
        if (lane < nb) {
            int len = 0;
            for (int n = c_node[lane]; n > 0; n = pparent[n]) ++len;
            int pos = len;
            for (int n = c_node[lane]; n > 0; n = pparent[n]) ot[--pos] = ptoken[n];
            p.out_len[b * beam + lane] = len;
            p.out_score[b * beam + lane] = c_sc[lane];
        } else {
            p.out_len[b * beam + lane] = -1;
            p.out_score[b * beam + lane] = NEG_INF;
        }
    }
}

}  // namespace
}  // namespace pafc

extern "C" size_t pafc_ctc_prefix_beam_workspace_bytes(int B, int T, int beam) {
    if (B <= 0 || T <= 0 || beam <= 0) return 0;
    return (size_t)2 * B * (1 + (size_t)T * beam) * sizeof(int32_t);
}

extern "C" int pafc_ctc_prefix_beam_search(int B, int T, int K, const float *top_logp, const int32_t *top_idx,
                                           const int64_t *lens, int beam, int blank_id, int32_t *out_tokens,
                                           int32_t *out_len, double *out_score, void *workspace, size_t workspace_bytes,
                                           pafc_stream_t stream) {
    if (!top_logp || !top_idx || !out_tokens || !out_len || !out_score || !workspace) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || K <= 0 || beam <= 0 || blank_id < 0) return PAFC_ERR_BAD_DIMS;
    if (K > pafc::MAXB || beam > pafc::MAXB) return PAFC_ERR_UNSUPPORTED;
    if (workspace_bytes < pafc_ctc_prefix_beam_workspace_bytes(B, T, beam)) return PAFC_ERR_WORKSPACE;
    pafc::BeamParams p{};
    p.T = T; p.K = K; p.beam = beam; p.blank = blank_id;
    p.top_logp = top_logp; p.top_idx = top_idx; p.lens = lens;
    p.pool_parent = (int32_t *)workspace;
    p.pool_token = p.pool_parent + (size_t)B * (1 + (size_t)T * beam);
    p.out_tokens = out_tokens; p.out_len = out_len; p.out_score = out_score;
    hipLaunchKernelGGL(pafc::ctc_prefix_beam_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
