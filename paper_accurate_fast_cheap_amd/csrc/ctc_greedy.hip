// CTC greedy search on the GPU (C ABI: include/pafc_search.h).
//
// Kernel 1 (argmax): one wave per (utterance, frame) row of V scores, 16 bytes per lane per access; each lane keeps
// the first maximum of its own ascending index sequence, the wave reduction prefers the larger value and, on equal
// values, the lower index -- the tie rule of torch.topk(1) / argmax that the reference relies on
// (wenet/transformer/search.py:112-116).  HBM-bound: V * elem bytes per frame (10 kB at V = 5000 bf16), the largest
// activation of the whole pass.
// Kernel 2 (collapse): one block per utterance walks its frames in tiles of 256: keep[t] = id[t] != blank and
// id[t] != id[t-1]; a ballot/popcount prefix sum compacts the kept ids (remove_duplicates_and_blank,
// wenet/utils/ctc_utils.py:22-32).
#include "pafc_common.h"
#include "../../include/pafc_search.h"

namespace pafc {
namespace {

template <typename ET>
__global__ __launch_bounds__(256) void ctc_argmax_kernel(long rows, int T, int V, const ET *scores, const int64_t *lens,
                                                         int blank, int32_t *best) {
    using E = Elem<ET>;
    constexpr int EPL = E::kPerLane;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)(row / T), t = (int)(row % T);
    if (lens != nullptr && t >= lens[b]) {
        if (lane == 0) best[row] = blank;
        return;
    }
    const ET *p = scores + row * (long)V;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    // rows start 16-byte aligned only when V * sizeof(ET) is a multiple of 16: otherwise peel to scalar loads
    const bool vec = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    const int nvec = vec ? V / EPL : 0;
    for (int c = lane; c < nvec; c += 64) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p + (long)c * EPL);
        float f[EPL];
        E::unpack(q, f);
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (f[e] > bv || (bi == 0x7fffffff && !(f[e] < bv))) { bv = f[e]; bi = c * EPL + e; }
    }
    for (int i = nvec * EPL + lane; i < V; i += 64) {
        const float f = E::load(p + i);
        if (f > bv || (bi == 0x7fffffff && !(f < bv))) { bv = f; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) best[row] = bi == 0x7fffffff ? 0 : bi;
}

__global__ __launch_bounds__(256) void ctc_collapse_kernel(int T, int blank, const int32_t *best, int32_t *tokens,
                                                           int32_t *ntok, int32_t *frames) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *ids = best + (long)b * T;
    int32_t *out = tokens + (long)b * T;
    int32_t *fout = frames ? frames + (long)b * T : nullptr;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        int id = blank, prev = blank;
        bool keep = false;
        if (t < T) {
            id = ids[t];
            prev = t > 0 ? ids[t - 1] : -1;
            keep = id != blank && id != prev;
        }
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (keep) {
            out[off + before] = id;
            if (fout) fout[off + before] = t;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) ntok[b] = s_base;
}

// log-softmax of one row per wave.  RC 16-byte chunks per lane live in registers between the reduction and the write
// (one read of HBM); rows longer than 64 * RC chunks (or not 16-byte aligned) take the three-sweep path, whose second
// and third sweeps hit the cache.
template <typename ET, int RC>
__global__ __launch_bounds__(256) void log_softmax_kernel(long rows, int V, const ET *x, ET *out) {
    using E = Elem<ET>;
    constexpr int EPL = E::kPerLane;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const ET *p = x + row * (long)V;
    ET *o = out + row * (long)V;
    const int nvec = V / EPL;
    const bool fast = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(o)) & 15) == 0 && nvec <= 64 * RC;
    float mx = -INFINITY, sum = 0.f;
    if (fast) {
        uint4 q[RC];
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            const int idx = c * 64 + lane;
            if (idx < nvec) {
                q[c] = *reinterpret_cast<const uint4 *>(p + (long)idx * EPL);
                float f[EPL];
                E::unpack(q[c], f);
#pragma unroll
                for (int e = 0; e < EPL; ++e) mx = fmaxf(mx, f[e]);
            }
        }
        float tail = -INFINITY;
        const int ti = nvec * EPL + lane;
        if (ti < V) { tail = E::load(p + ti); mx = fmaxf(mx, tail); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            if (c * 64 + lane < nvec) {
                float f[EPL];
                E::unpack(q[c], f);
#pragma unroll
                for (int e = 0; e < EPL; ++e) sum += __expf(f[e] - mx);
            }
        }
        if (ti < V) sum += __expf(tail - mx);
        sum = wave_sum(sum);
        const float lse = __logf(sum);
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            const int idx = c * 64 + lane;
            if (idx < nvec) {
                float f[EPL];
                E::unpack(q[c], f);
                if constexpr (EPL == 8) {
                    uint4 w;
                    w.x = f32_to_bf16_bits(f[0] - mx - lse) | (f32_to_bf16_bits(f[1] - mx - lse) << 16);
                    w.y = f32_to_bf16_bits(f[2] - mx - lse) | (f32_to_bf16_bits(f[3] - mx - lse) << 16);
                    w.z = f32_to_bf16_bits(f[4] - mx - lse) | (f32_to_bf16_bits(f[5] - mx - lse) << 16);
                    w.w = f32_to_bf16_bits(f[6] - mx - lse) | (f32_to_bf16_bits(f[7] - mx - lse) << 16);
                    *reinterpret_cast<uint4 *>(o + (long)idx * EPL) = w;
                } else {
                    const float4 w = {f[0] - mx - lse, f[1] - mx - lse, f[2] - mx - lse, f[3] - mx - lse};
                    *reinterpret_cast<float4 *>(o + (long)idx * EPL) = w;
                }
            }
        }
        if (ti < V) E::store(o + ti, tail - mx - lse);
        return;
    }
    for (int i = lane; i < V; i += 64) mx = fmaxf(mx, E::load(p + i));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    for (int i = lane; i < V; i += 64) sum += __expf(E::load(p + i) - mx);
    sum = wave_sum(sum);
    const float lse = __logf(sum);
    for (int i = lane; i < V; i += 64) E::store(o + i, E::load(p + i) - mx - lse);
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_ctc_greedy(int dtype, int B, int T, int V, const void *scores, const int64_t *lens, int blank_id,
                               int32_t *best, int32_t *tokens, int32_t *ntok, int32_t *frames, pafc_stream_t stream) {
    if (!scores || !best || !tokens || !ntok) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || V <= 0 || blank_id < 0 || blank_id >= V) return PAFC_ERR_BAD_DIMS;
    if (dtype != PAFC_F32 && dtype != PAFC_BF16) return PAFC_ERR_DTYPE;
    const long rows = (long)B * T;
    const long nblk = (rows + 3) / 4;
    if (nblk > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_BF16)
        hipLaunchKernelGGL((pafc::ctc_argmax_kernel<pafc::bf16_t>), dim3((unsigned)nblk), dim3(256), 0, s, rows, T, V,
                           (const pafc::bf16_t *)scores, lens, blank_id, best);
    else
        hipLaunchKernelGGL((pafc::ctc_argmax_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, s, rows, T, V,
                           (const float *)scores, lens, blank_id, best);
    hipLaunchKernelGGL(pafc::ctc_collapse_kernel, dim3(B), dim3(256), 0, s, T, blank_id, best, tokens, ntok, frames);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_log_softmax_rows(int dtype, long rows, int V, const void *x, void *out, pafc_stream_t stream) {
    if (!x || !out) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || V <= 0) return PAFC_ERR_BAD_DIMS;
    if (dtype != PAFC_F32 && dtype != PAFC_BF16) return PAFC_ERR_DTYPE;
    const long nblk = (rows + 3) / 4;
    if (nblk > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_BF16)
        hipLaunchKernelGGL((pafc::log_softmax_kernel<pafc::bf16_t, 16>), dim3((unsigned)nblk), dim3(256), 0, s, rows, V,
                           (const pafc::bf16_t *)x, (pafc::bf16_t *)out);
    else
        // fp32 rows hold 4 values per 16-byte chunk: 20 chunks per lane keep a V <= 5120 row (the 5000-token vocabulary) in
        // registers -- with 16 the fp32 model's CTC rows took the three-sweep path (984 us per 30-minute file, 1.8 TB/s)
        hipLaunchKernelGGL((pafc::log_softmax_kernel<float, 20>), dim3((unsigned)nblk), dim3(256), 0, s, rows, V,
                           (const float *)x, (float *)out);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
