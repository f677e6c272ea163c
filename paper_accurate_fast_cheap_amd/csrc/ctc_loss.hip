// CTC loss of the training step, from the LOGITS, on gfx950 (C ABI: include/pafc_encoder_ops.h: pafc_ctc_loss_*).
//
// Replaces `ys_hat.log_softmax(2)` + `torch.nn.CTCLoss(reduction="sum", zero_infinity=True)` + their autograd of CTC.forward
// (wenet/transformer/ctc.py:53-82; the loss of the DDP training step, wenet/utils/train_utils.py:609-729, config c4: 32
// utterances x T' <= 499 frames x V = 5000 per GPU).  The framework's path writes the (B, T', V) log-probabilities, reads them
// back in its alpha / beta kernels, writes a dense gradient, and runs log-softmax's backward over it: four passes over the
// largest activation of the step, and its CTC kernels copy the length tensors back to the host (five synchronising calls).
// Here the dense tensor is read twice and written once, the recursions only ever see the <= 2 L + 1 label columns of a row, and
// nothing waits for the host:
//
//   rows    one block per (b, t < hlen[b]) row of logits: lse = log sum exp (fp32), and the S = 2 L_b + 1 extended-label entries
//           lp[b][t][s] = logit[l'_s] - lse, l' = (blank, y_1, blank, y_2, ..., blank).
//   lattice one block per utterance, a thread per extended label s: alpha_t(s) = lp_t(s) + logsumexp(alpha_{t-1}(s),
//           alpha_{t-1}(s-1), [alpha_{t-1}(s-2) if l'_s != blank and l'_s != l'_{s-2}]) forward in t, stored; the same backwards
//           for beta; log-likelihood ll = logsumexp(alpha_{T-1}(S-1), alpha_{T-1}(S-2)); the posterior occupancy of (t, s),
//           occ = exp(alpha + beta - lp - ll), overwrites alpha.  nll[b] = -ll, or 0 when no alignment exists (zero_infinity).
//   grad    one block per row: d loss / d logit[c] = g (softmax(logit)[c] - sum over s with l'_s = c of occ(t, s)) -- the
//           gradient THROUGH the log-softmax (its rows sum to zero) -- written in the logits' dtype into rows of `ldg` columns
//           whose tail beyond V is zeroed (the head's input-gradient GEMM wants K a multiple of 64); rows t >= hlen[b] and
//           utterances without an alignment get zeros.  g = grad_out[0] * scale (scale = 1 / B: ctc.py:77).
//
// fp32 arithmetic throughout; logits bf16 or fp32.
#include <math.h>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr float NEG = -1e30f;     // "log 0": large enough that exp underflows to 0, finite so that NEG - NEG = 0 never makes a NaN

__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    return m <= NEG ? NEG : m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    return m <= NEG ? NEG : m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float *sm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();                          // (sm may still be read from a previous reduction)
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = sm[0];
    for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, sm[i]) : r + sm[i];
    return r;
}

// extended label s of utterance b
__device__ __forceinline__ int ext_label(const int64_t *ys, int s, int blank) { return (s & 1) ? (int)ys[s >> 1] : blank; }

template <typename ET>
__global__ __launch_bounds__(256) void ctc_rows_kernel(int T, int V, long ldl, const ET *logits, const int32_t *hlens,
                                                       const int64_t *ys, int ldy, const int32_t *ylens, int blank, int Smax,
                                                       float *lse, float *lp) {
    using E = Elem<ET>;
    __shared__ float sm[4];
    const int t = blockIdx.x, b = blockIdx.y;
    if (t >= hlens[b]) return;
    const ET *row = logits + ((long)b * T + t) * ldl;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, E::load(row + c));
    mx = block_reduce(mx, true, sm);
    float sum = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) sum += __expf(E::load(row + c) - mx);
    sum = block_reduce(sum, false, sm);
    const float l = mx + __logf(sum);
    if (threadIdx.x == 0) lse[(long)b * T + t] = l;
    const int L = ylens[b], S = 2 * L + 1;
    const int64_t *y = ys + (long)b * ldy;
    float *out = lp + ((long)b * T + t) * Smax;
    for (int s = threadIdx.x; s < S; s += 256) {
        const int c = ext_label(y, s, blank);
        out[s] = (c >= 0 && c < V) ? E::load(row + c) - l : NEG;
    }
}

// one block per utterance; blockDim.x threads stride over the S extended labels; lp / ab: [b][t][Smax]
__global__ __launch_bounds__(512) void ctc_lattice_kernel(int T, const int32_t *hlens, const int64_t *ys, int ldy, const int32_t *ylens,
                                                          int blank, int Smax, const float *lp, float *ab, float *nll, float *ok) {
    extern __shared__ float sh[];                   // [2][Smax + 2]: the previous time step, two guard entries in front
    const int b = blockIdx.x, Tb = hlens[b], L = ylens[b], S = 2 * L + 1;
    const int64_t *y = ys + (long)b * ldy;
    const float *lpb = lp + (long)b * T * Smax;
    float *abb = ab + (long)b * T * Smax;
    const int W = Smax + 2;
    if (Tb <= 0 || L > Tb) {                        // no frames, or more labels than frames: no alignment (zero_infinity -> 0)
        if (threadIdx.x == 0) { nll[b] = 0.f; ok[b] = 0.f; }
        for (long i = threadIdx.x; i < (long)max(Tb, 0) * Smax; i += blockDim.x) abb[i] = 0.f;
        return;
    }
    // ---- alpha -----------------------------------------------------------------------------------------------------
    for (int s = threadIdx.x; s < W; s += blockDim.x) sh[s] = sh[W + s] = NEG;
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const float a = s < 2 ? lpb[s] : NEG;
        sh[2 + s] = a;
        abb[s] = a;
    }
    __syncthreads();
    for (int t = 1; t < Tb; ++t) {
        const float *prev = sh + ((t - 1) & 1) * W;
        float *cur = sh + (t & 1) * W;
        for (int s = threadIdx.x; s < S; s += blockDim.x) {
            const int c = ext_label(y, s, blank);
            const bool skip = (s & 1) && s >= 2 && c != (int)y[(s >> 1) - 1];
            const float a = lpb[(long)t * Smax + s] + lse3(prev[2 + s], prev[1 + s], skip ? prev[s] : NEG);
            cur[2 + s] = a;
            abb[(long)t * Smax + s] = a;
        }
        __syncthreads();
    }
    const float *last = sh + ((Tb - 1) & 1) * W;
    const float ll = lse2(last[2 + S - 1], S >= 2 ? last[2 + S - 2] : NEG);
    const bool feasible = ll > -1e29f;
    if (threadIdx.x == 0) { nll[b] = feasible ? -ll : 0.f; ok[b] = feasible ? 1.f : 0.f; }
    __syncthreads();
    if (!feasible) {
        for (long i = threadIdx.x; i < (long)Tb * Smax; i += blockDim.x) abb[i] = 0.f;
        return;
    }
    // ---- beta, and the occupancies over alpha ----------------------------------------------------------------------
    // guards BEHIND the row this time: index s, s + 1, s + 2 of the next time step
    for (int s = threadIdx.x; s < W; s += blockDim.x) sh[s] = sh[W + s] = NEG;
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const float be = s >= S - 2 ? lpb[(long)(Tb - 1) * Smax + s] : NEG;
        sh[((Tb - 1) & 1) * W + s] = be;
        const long i = (long)(Tb - 1) * Smax + s;
        abb[i] = __expf(abb[i] + be - lpb[i] - ll);
    }
    __syncthreads();
    for (int t = Tb - 2; t >= 0; --t) {
        const float *next = sh + ((t + 1) & 1) * W;
        float *cur = sh + (t & 1) * W;
        for (int s = threadIdx.x; s < S; s += blockDim.x) {
            const int c = ext_label(y, s, blank);
            const bool skip = (s & 1) && s + 2 < S && c != (int)y[(s >> 1) + 1];
            const long i = (long)t * Smax + s;
            const float l = lpb[i];
            const float be = l + lse3(next[s], next[s + 1], skip ? next[s + 2] : NEG);
            cur[s] = be;
            abb[i] = __expf(abb[i] + be - l - ll);
        }
        __syncthreads();
    }
}

template <typename ET>
__global__ __launch_bounds__(256) void ctc_grad_kernel(int T, int V, long ldl, const ET *logits, const int32_t *hlens,
                                                       const int64_t *ys, int ldy, const int32_t *ylens, int blank, int Smax,
                                                       const float *lse, const float *occ, const float *ok, const float *grad_out,
                                                       float scale, long ldg, ET *dlogits) {
    using E = Elem<ET>;
    extern __shared__ float acc[];                  // [V]: occupancy per vocabulary entry of this row
    const int t = blockIdx.x, b = blockIdx.y;
    ET *drow = dlogits + ((long)b * T + t) * ldg;
    const int L = ylens[b], Tb = hlens[b];
    const bool live = t < Tb && ok[b] != 0.f;       // (an utterance without an alignment: zero gradient, as zero_infinity asks)
    if (!live) {
        for (int c = threadIdx.x; c < ldg; c += 256) E::store(drow + c, 0.f);
        return;
    }
    for (int c = threadIdx.x; c < V; c += 256) acc[c] = 0.f;
    __syncthreads();
    const int S = 2 * L + 1;
    const int64_t *y = ys + (long)b * ldy;
    const float *o = occ + ((long)b * T + t) * Smax;
    for (int s = threadIdx.x; s < S; s += 256) {
        const int c = ext_label(y, s, blank);
        if (c >= 0 && c < V) atomicAdd(&acc[c], o[s]);
    }
    __syncthreads();
    const ET *row = logits + ((long)b * T + t) * ldl;
    const float l = lse[(long)b * T + t], g = grad_out[0] * scale;
    for (int c = threadIdx.x; c < V; c += 256) E::store(drow + c, g * (__expf(E::load(row + c) - l) - acc[c]));
    for (int c = V + threadIdx.x; c < ldg; c += 256) E::store(drow + c, 0.f);
}

}  // namespace
}  // namespace pafc

extern "C" {

size_t pafc_ctc_loss_workspace_bytes(int B, int T, int max_target_len) {
    if (B <= 0 || T <= 0 || max_target_len < 0) return 0;
    const size_t Smax = 2 * (size_t)max_target_len + 1;
    return ((size_t)B * T * (1 + 2 * Smax) + B) * sizeof(float);  // lse + lp + alpha / occupancies + one flag per utterance
}

int pafc_ctc_loss_forward(int dtype, int B, int T, int V, const void *logits, long ldl, const int32_t *hlens, const int64_t *ys,
                          int ldy, const int32_t *ylens, int max_target_len, int blank, float *nll, void *workspace,
                          size_t workspace_bytes, pafc_stream_t stream) {
    if (!logits || !hlens || !ys || !ylens || !nll || !workspace) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || V <= 0 || max_target_len < 0 || ldl < V || ldy < max_target_len || B > 65535) return PAFC_ERR_BAD_DIMS;
    if (dtype != PAFC_F32 && dtype != PAFC_BF16) return PAFC_ERR_DTYPE;
    if (workspace_bytes < pafc_ctc_loss_workspace_bytes(B, T, max_target_len)) return PAFC_ERR_WORKSPACE;
    const int Smax = 2 * max_target_len + 1;
    if ((size_t)2 * (Smax + 2) * sizeof(float) > 60 * 1024) return PAFC_ERR_UNSUPPORTED;
    float *lse = (float *)workspace, *lp = lse + (size_t)B * T, *ab = lp + (size_t)B * T * Smax, *ok = ab + (size_t)B * T * Smax;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_F32)
        hipLaunchKernelGGL(pafc::ctc_rows_kernel<float>, dim3(T, B), dim3(256), 0, s, T, V, ldl, (const float *)logits, hlens, ys, ldy,
                           ylens, blank, Smax, lse, lp);
    else
        hipLaunchKernelGGL(pafc::ctc_rows_kernel<pafc::bf16_t>, dim3(T, B), dim3(256), 0, s, T, V, ldl, (const pafc::bf16_t *)logits,
                           hlens, ys, ldy, ylens, blank, Smax, lse, lp);
    const int threads = Smax <= 64 ? 64 : Smax <= 128 ? 128 : Smax <= 256 ? 256 : 512;
    hipLaunchKernelGGL(pafc::ctc_lattice_kernel, dim3(B), dim3(threads), (size_t)2 * (Smax + 2) * sizeof(float), s, T, hlens, ys, ldy,
                       ylens, blank, Smax, lp, ab, nll, ok);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

int pafc_ctc_loss_backward(int dtype, int B, int T, int V, const void *logits, long ldl, const int32_t *hlens, const int64_t *ys,
                           int ldy, const int32_t *ylens, int max_target_len, int blank, const float *nll, const float *grad_out,
                           float scale, void *dlogits, long ldg, const void *workspace, size_t workspace_bytes,
                           pafc_stream_t stream) {
    if (!logits || !hlens || !ys || !ylens || !nll || !grad_out || !dlogits || !workspace) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || V <= 0 || max_target_len < 0 || ldl < V || ldg < V || ldy < max_target_len || B > 65535)
        return PAFC_ERR_BAD_DIMS;
    if (dtype != PAFC_F32 && dtype != PAFC_BF16) return PAFC_ERR_DTYPE;
    if (workspace_bytes < pafc_ctc_loss_workspace_bytes(B, T, max_target_len)) return PAFC_ERR_WORKSPACE;
    if ((size_t)V * sizeof(float) > 150 * 1024) return PAFC_ERR_UNSUPPORTED;
    const int Smax = 2 * max_target_len + 1;
    const float *lse = (const float *)workspace, *ab = lse + (size_t)B * T + (size_t)B * T * Smax, *ok = ab + (size_t)B * T * Smax;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)V * sizeof(float);
    if (dtype == PAFC_F32) {
        auto k = pafc::ctc_grad_kernel<float>;
        if (lds > 48 * 1024 && hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PAFC_ERR_LAUNCH;
        hipLaunchKernelGGL(k, dim3(T, B), dim3(256), lds, s, T, V, ldl, (const float *)logits, hlens, ys, ldy, ylens, blank, Smax, lse,
                           ab, ok, grad_out, scale, ldg, (float *)dlogits);
    } else {
        auto k = pafc::ctc_grad_kernel<pafc::bf16_t>;
        if (lds > 48 * 1024 && hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PAFC_ERR_LAUNCH;
        hipLaunchKernelGGL(k, dim3(T, B), dim3(256), lds, s, T, V, ldl, (const pafc::bf16_t *)logits, hlens, ys, ldy, ylens, blank, Smax,
                           lse, ab, ok, grad_out, scale, ldg, (pafc::bf16_t *)dlogits);
    }
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // extern "C"
