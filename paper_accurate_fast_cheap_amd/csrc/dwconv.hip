// Channels-last depthwise conv1d (+ optional fused GLU input and length mask) for gfx950.
// See include/pafc_encoder_ops.h: pafc_dwconv1d_cl.
//
// HBM-bound: one read of x, one write of y.  A block is 256 threads = 4 waves = 512 channels... in general
// C/2 lanes wide in chunks of 128 channels per wave; each lane owns 2 adjacent channels (one 4-byte bf16x2 or
// 8-byte f32x2 access, so a wave touches 256/512 contiguous bytes of a row) and TO consecutive output frames.
// The K taps of its two channels live in registers; every input row of the tile (+halo) is loaded once and
// scattered into the <= K output accumulators it feeds, with all indices compile-time (full unroll).
#include <stdlib.h>
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int KMAX = 31;
constexpr int TO = 16;  // output frames per lane

template <typename ET> struct Pair;
template <> struct Pair<float> {
    __device__ static __forceinline__ float2 load(const float *p) { return *reinterpret_cast<const float2 *>(p); }
    __device__ static __forceinline__ void store(float *p, float2 v) { *reinterpret_cast<float2 *>(p) = v; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct Pair<bf16_t> {
    __device__ static __forceinline__ float2 load(const bf16_t *p) {
        const uint32_t q = *reinterpret_cast<const uint32_t *>(p);
        return make_float2(bf16_bits_to_f32(q & 0xffffu), __uint_as_float(q & 0xffff0000u));
    }
    __device__ static __forceinline__ void store(bf16_t *p, float2 v) {
        *reinterpret_cast<uint32_t *>(p) = f32_to_bf16_bits(v.x) | (f32_to_bf16_bits(v.y) << 16);
    }
    __device__ static __forceinline__ float round(float v) { return round_bf16(v); }
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// The conv module's LayerNorm + SiLU (convolution.py:134-138: `self.activation(self.norm(x))`) applied by the convolution's own
// epilogue: a block already owns ALL the channels of its 16 output frames (C = 512 = 256 lanes x 2), so the rows' statistics are
// two block reductions and the normalised rows never make a round trip through memory as the convolution's raw output.
struct LnTail {
    const void *gamma, *beta;   // (C), element type
    float eps;
};

// ACT: 0 plain, 1 GLU on the input (x is (.., 2C)), 2 SiLU on the output (Mamba-2's conv1d + SiLU),
//      3 = y = SiLU(LayerNorm_C(conv output)), every intermediate rounded where the module chain rounds it (C == 512 only)
template <typename ET, int K, int ACT>
__global__ __launch_bounds__(256) void dwconv_kernel(int T_in, int C, int left_pad, int T_out, const ET *__restrict__ x,
                                                     long ldx, const ET *__restrict__ w, const ET *__restrict__ bias,
                                                     ET *__restrict__ y, const int32_t *__restrict__ lens, const LnTail ln) {
    constexpr bool GLU = ACT == 1;
    const int c = (blockIdx.y * 256 + threadIdx.x) * 2;
    if (c >= C) return;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * TO;
    const long xc = ldx;             // row stride of x in elements (2C for GLU, C, or a wider row the input is a slice of)
    const ET *xb = x + (size_t)b * T_in * xc;
    const int valid = lens ? min(T_in, lens[b]) : T_in;

    float w0[K], w1[K];
    if constexpr (sizeof(ET) == 2) {
        // the 2 K taps of channels c, c + 1 are K consecutive dwords (c is even): K loads instead of 2 K two-byte ones --
        // the tap fetch was the larger half of this kernel's load instructions
        const uint32_t *wd = reinterpret_cast<const uint32_t *>(w + (size_t)c * K);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t d = wd[j];
            const float lo = bf16_bits_to_f32(d & 0xffffu), hi = __uint_as_float(d & 0xffff0000u);
            if (2 * j < K) w0[2 * j] = lo; else w1[2 * j - K] = lo;
            if (2 * j + 1 < K) w0[2 * j + 1] = hi; else w1[2 * j + 1 - K] = hi;
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            w0[k] = Elem<ET>::load(w + (size_t)c * K + k);
            w1[k] = Elem<ET>::load(w + (size_t)(c + 1) * K + k);
        }
    }
    float2 bv = make_float2(0.f, 0.f);
    if (bias) bv = Pair<ET>::load(bias + c);
    float a0[TO], a1[TO];
#pragma unroll
    for (int o = 0; o < TO; ++o) { a0[o] = bv.x; a1[o] = bv.y; }

    // input rows s = t0 - left_pad + q, q in [0, TO + K - 1); row q feeds output o with tap k = q - o.
    // All rows are fetched first, branch-free (clamped address + select), so that the loads are in flight
    // together instead of one exposed HBM round trip per row; then the taps are applied from registers.
    constexpr int NR = TO + K - 1;
    float2 in[NR];
    float2 gate[GLU ? NR : 1];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int s = t0 - left_pad + q;
        const int sc = min(max(s, 0), T_in - 1);
        in[q] = Pair<ET>::load(xb + (size_t)sc * xc + c);
        if (GLU) gate[q] = Pair<ET>::load(xb + (size_t)sc * xc + C + c);
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int s = t0 - left_pad + q;
        float2 v = in[q];
        if (GLU) {
            v.x = Pair<ET>::round(v.x * sigmoidf_(gate[q].x));  // torch computes glu in float, one rounding
            v.y = Pair<ET>::round(v.y * sigmoidf_(gate[q].y));
        }
        if (!(s >= 0 && s < valid)) v = make_float2(0.f, 0.f);
#pragma unroll
        for (int o = 0; o < TO; ++o) {
            const int k = q - o;
            if (k >= 0 && k < K) {
                a0[o] = fmaf(w0[k], v.x, a0[o]);
                a1[o] = fmaf(w1[k], v.y, a1[o]);
            }
        }
    }
    ET *yb = y + (size_t)b * T_out * C + c;
    if constexpr (ACT == 3) {
        // rows = my 16 output frames, all 512 channels in this block: u = the convolution output as the module chain stores it
        // (rounded), LayerNorm with the two-pass variance of add_layernorm_kernel, rounded, SiLU, rounded.
        __shared__ __attribute__((aligned(16))) float red[256 * 20];     // [lane of the block][16 rows (+ 4 pad)]
        __shared__ __attribute__((aligned(16))) float tot[TO];
        const int tid = threadIdx.x, rr = tid >> 4, jj = tid & 15;
        auto block_sums = [&](const float (&v)[TO]) {     // tot[r] = sum over the block's 256 lanes of v[r]
#pragma unroll
            for (int q = 0; q < TO / 4; ++q)
                *reinterpret_cast<float4 *>(&red[tid * 20 + 4 * q]) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            __syncthreads();
            float p = 0.f;                               // lane (row rr, part jj): lanes 16 jj .. 16 jj + 15 of row rr
#pragma unroll
            for (int i = 0; i < 16; ++i) p += red[(16 * jj + i) * 20 + rr];
            p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 8, 64);
            if (jj == 0) tot[rr] = p;
            __syncthreads();
        };
        float u0[TO], u1[TO], sv[TO];
#pragma unroll
        for (int o = 0; o < TO; ++o) { u0[o] = Pair<ET>::round(a0[o]); u1[o] = Pair<ET>::round(a1[o]); sv[o] = u0[o] + u1[o]; }
        block_sums(sv);
        const float inv_c = 1.f / (float)C;
        float mean[TO];
#pragma unroll
        for (int o = 0; o < TO; ++o) {
            mean[o] = tot[o] * inv_c;
            const float d0 = u0[o] - mean[o], d1 = u1[o] - mean[o];
            sv[o] = fmaf(d1, d1, d0 * d0);
        }
        __syncthreads();                                 // every lane has read tot before the second reduction overwrites it
        block_sums(sv);
        const float2 gv = Pair<ET>::load((const ET *)ln.gamma + c), bt = Pair<ET>::load((const ET *)ln.beta + c);
#pragma unroll
        for (int o = 0; o < TO; ++o) {
            const float rstd = rsqrtf(tot[o] * inv_c + ln.eps);
            float o0 = Pair<ET>::round(fmaf((u0[o] - mean[o]) * rstd, gv.x, bt.x));
            float o1 = Pair<ET>::round(fmaf((u1[o] - mean[o]) * rstd, gv.y, bt.y));
            o0 = o0 * sigmoidf_(o0);
            o1 = o1 * sigmoidf_(o1);
            if (t0 + o < T_out) Pair<ET>::store(yb + (size_t)(t0 + o) * C, make_float2(o0, o1));
        }
        return;
    }
#pragma unroll
    for (int o = 0; o < TO; ++o) {
        if (ACT == 2) {   // SiLU of the rounded convolution output, as the framework's two ops compute it
            const float u0 = Pair<ET>::round(a0[o]), u1 = Pair<ET>::round(a1[o]);
            a0[o] = u0 * sigmoidf_(u0);
            a1[o] = u1 * sigmoidf_(u1);
        }
        if (t0 + o < T_out) Pair<ET>::store(yb + (size_t)(t0 + o) * C, make_float2(a0[o], a1[o]));
    }
}

template <typename ET, int K>
int launch(int B, int T_in, int C, int left_pad, int T_out, const void *x, long ldx, const void *w, const void *bias,
           void *y, int act, const int32_t *lens, const LnTail &ln, hipStream_t s) {
    dim3 grid((T_out + TO - 1) / TO, (C / 2 + 255) / 256, B), block(256);
    if (act == 1)
        hipLaunchKernelGGL((dwconv_kernel<ET, K, 1>), grid, block, 0, s, T_in, C, left_pad, T_out, (const ET *)x, ldx,
                           (const ET *)w, (const ET *)bias, (ET *)y, lens, ln);
    else if (act == 2)
        hipLaunchKernelGGL((dwconv_kernel<ET, K, 2>), grid, block, 0, s, T_in, C, left_pad, T_out, (const ET *)x, ldx,
                           (const ET *)w, (const ET *)bias, (ET *)y, lens, ln);
    else if (act == 3) {
        if constexpr (K == 31 || K == 15) {        // the conv module's kernel sizes (conf/rwkv/*.yaml: cnn_module_kernel)
            hipLaunchKernelGGL((dwconv_kernel<ET, K, 3>), grid, block, 0, s, T_in, C, left_pad, T_out, (const ET *)x, ldx,
                               (const ET *)w, (const ET *)bias, (ET *)y, lens, ln);
        } else {
            return PAFC_ERR_UNSUPPORTED;
        }
    } else
        hipLaunchKernelGGL((dwconv_kernel<ET, K, 0>), grid, block, 0, s, T_in, C, left_pad, T_out, (const ET *)x, ldx,
                           (const ET *)w, (const ET *)bias, (ET *)y, lens, ln);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

template <typename ET>
int dispatch_k(int K, int B, int T_in, int C, int left_pad, int T_out, const void *x, long ldx, const void *w,
               const void *bias, void *y, int act, const int32_t *lens, const LnTail &ln, hipStream_t s) {
    switch (K) {
        case 31: return launch<ET, 31>(B, T_in, C, left_pad, T_out, x, ldx, w, bias, y, act, lens, ln, s);
        case 15: return launch<ET, 15>(B, T_in, C, left_pad, T_out, x, ldx, w, bias, y, act, lens, ln, s);
        case 7: return launch<ET, 7>(B, T_in, C, left_pad, T_out, x, ldx, w, bias, y, act, lens, ln, s);
        case 4: return launch<ET, 4>(B, T_in, C, left_pad, T_out, x, ldx, w, bias, y, act, lens, ln, s);
        case 3: return launch<ET, 3>(B, T_in, C, left_pad, T_out, x, ldx, w, bias, y, act, lens, ln, s);
        default: return PAFC_ERR_UNSUPPORTED;
    }
}


// Weight / bias gradient of the same convolution (the training step, config c4):
//   dw[c][k] = sum_{b,t} dy[b][t][c] * x[b][t + k - left_pad][c],   db[c] = sum_{b,t} dy[b][t][c]
// A lane owns 2 channels and keeps their 2 (K + 1) sums in registers while its block walks WG_SPAN output frames in
// tiles of TO (same register tiling as the forward: every x row of a tile is loaded once and meets the <= K dy rows
// it was multiplied into); a block leaves one fp32 partial per (k, c), a second kernel adds the partials in a fixed
// order (deterministic: no atomics).
constexpr int WG_SPAN = 8 * TO;

template <typename ET, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(int T_in, int C, int left_pad, int T_out,
                                                           const ET *__restrict__ x, long ldx,
                                                           const ET *__restrict__ dy, float *__restrict__ part) {
    const int c = (blockIdx.y * 256 + threadIdx.x) * 2;
    if (c >= C) return;
    const int b = blockIdx.z;
    const ET *xb = x + (size_t)b * T_in * ldx;
    const ET *db = dy + (size_t)b * T_out * C;
    float a0[K], a1[K], s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) a0[k] = a1[k] = 0.f;
    constexpr int NR = TO + K - 1;
    for (int t0 = blockIdx.x * WG_SPAN; t0 < min(T_out, (int)(blockIdx.x + 1) * WG_SPAN); t0 += TO) {
        float2 g[TO], in[NR];
#pragma unroll
        for (int o = 0; o < TO; ++o) {
            const int t = min(t0 + o, T_out - 1);
            g[o] = Pair<ET>::load(db + (size_t)t * C + c);
            if (t0 + o >= T_out) g[o] = make_float2(0.f, 0.f);
            s0 += g[o].x;
            s1 += g[o].y;
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int s = t0 - left_pad + q;
            in[q] = Pair<ET>::load(xb + (size_t)min(max(s, 0), T_in - 1) * ldx + c);
            if (!(s >= 0 && s < T_in)) in[q] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q)
#pragma unroll
            for (int o = 0; o < TO; ++o) {
                const int k = q - o;
                if (k >= 0 && k < K) {
                    a0[k] = fmaf(g[o].x, in[q].x, a0[k]);
                    a1[k] = fmaf(g[o].y, in[q].y, a1[k]);
                }
            }
    }
    float *pp = part + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * (K + 1) * C + c;
#pragma unroll
    for (int k = 0; k < K; ++k) *reinterpret_cast<float2 *>(pp + (size_t)k * C) = make_float2(a0[k], a1[k]);
    *reinterpret_cast<float2 *>(pp + (size_t)K * C) = make_float2(s0, s1);
}

// dw (C, K) and db (C) in fp32 from nblk partials of (K + 1, C)
__global__ __launch_bounds__(256) void dwconv_wgrad_reduce_kernel(int C, int K, int nblk, const float *__restrict__ part,
                                                                  float *__restrict__ dw, float *__restrict__ dbias) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // over (K + 1) * C, c fastest: coalesced reads
    if (i >= (K + 1) * C) return;
    const int k = i / C, c = i - k * C;
    float s = 0.f;
    for (int p = 0; p < nblk; ++p) s += part[(size_t)p * (K + 1) * C + i];
    if (k < K)
        dw[(size_t)c * K + k] = s;
    else if (dbias)
        dbias[c] = s;
}

template <typename ET, int K>
int launch_wgrad(int B, int T_in, int C, int left_pad, int T_out, const void *x, long ldx, const void *dy, float *part,
                 float *dw, float *dbias, hipStream_t s) {
    const int nx = (T_out + WG_SPAN - 1) / WG_SPAN;
    dim3 grid(nx, (C / 2 + 255) / 256, B), block(256);
    hipLaunchKernelGGL((dwconv_wgrad_kernel<ET, K>), grid, block, 0, s, T_in, C, left_pad, T_out, (const ET *)x, ldx,
                       (const ET *)dy, part);
    hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel, dim3(((K + 1) * C + 255) / 256), block, 0, s, C, K, nx * B, part, dw,
                       dbias);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

template <typename ET>
int dispatch_wgrad(int K, int B, int T_in, int C, int left_pad, int T_out, const void *x, long ldx, const void *dy,
                   float *part, float *dw, float *dbias, hipStream_t s) {
    switch (K) {
        case 31: return launch_wgrad<ET, 31>(B, T_in, C, left_pad, T_out, x, ldx, dy, part, dw, dbias, s);
        case 15: return launch_wgrad<ET, 15>(B, T_in, C, left_pad, T_out, x, ldx, dy, part, dw, dbias, s);
        case 7: return launch_wgrad<ET, 7>(B, T_in, C, left_pad, T_out, x, ldx, dy, part, dw, dbias, s);
        case 4: return launch_wgrad<ET, 4>(B, T_in, C, left_pad, T_out, x, ldx, dy, part, dw, dbias, s);
        case 3: return launch_wgrad<ET, 3>(B, T_in, C, left_pad, T_out, x, ldx, dy, part, dw, dbias, s);
        default: return PAFC_ERR_UNSUPPORTED;
    }
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_dwconv1d_cl(int dtype, int B, int T_in, int C, int K, int left_pad, int T_out, const void *x,
                                const void *w, const void *bias, void *y, int glu, const int32_t *lens,
                                pafc_stream_t stream) {
    if (!x || !w || !y) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T_in <= 0 || T_out <= 0 || C <= 0 || (C % 2) || K <= 0 || K > pafc::KMAX || left_pad < 0 ||
        B > 65535)
        return PAFC_ERR_BAD_DIMS;
    return pafc_dwconv1d_cl_ex(dtype, B, T_in, C, K, left_pad, T_out, x, glu ? 2L * C : (long)C, w, bias, y, glu ? 1 : 0, lens,
                               stream);
}

extern "C" int pafc_dwconv1d_cl_ex(int dtype, int B, int T_in, int C, int K, int left_pad, int T_out, const void *x, long ldx,
                                   const void *w, const void *bias, void *y, int act, const int32_t *lens,
                                   pafc_stream_t stream) {
    if (!x || !w || !y) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T_in <= 0 || T_out <= 0 || C <= 0 || (C % 2) || K <= 0 || K > pafc::KMAX || left_pad < 0 ||
        B > 65535 || act < 0 || act > 2 || ldx < (act == 1 ? 2L * C : (long)C) || (ldx % 2))
        return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    const pafc::LnTail none{nullptr, nullptr, 0.f};
    if (dtype == PAFC_BF16)
        return pafc::dispatch_k<pafc::bf16_t>(K, B, T_in, C, left_pad, T_out, x, ldx, w, bias, y, act, lens, none, s);
    if (dtype == PAFC_F32) return pafc::dispatch_k<float>(K, B, T_in, C, left_pad, T_out, x, ldx, w, bias, y, act, lens, none, s);
    return PAFC_ERR_DTYPE;
}

extern "C" int pafc_dwconv1d_cl_ln_silu(int dtype, int B, int T_in, int C, int K, int left_pad, int T_out, const void *x, long ldx,
                                        const void *w, const void *bias, const void *gamma, const void *beta, float eps, void *y,
                                        const int32_t *lens, pafc_stream_t stream) {
    if (!x || !w || !y || !gamma || !beta) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T_in <= 0 || T_out <= 0 || left_pad < 0 || B > 65535 || ldx < C || (ldx % 2)) return PAFC_ERR_BAD_DIMS;
    if (C != 512 || (K != 31 && K != 15)) return PAFC_ERR_UNSUPPORTED;      // one block = all the channels of its frames
    if (dtype != PAFC_BF16) return PAFC_ERR_DTYPE;
    const pafc::LnTail ln{gamma, beta, eps};
    return pafc::dispatch_k<pafc::bf16_t>(K, B, T_in, C, left_pad, T_out, x, ldx, w, bias, y, 3, lens, ln, (hipStream_t)stream);
}

extern "C" size_t pafc_dwconv1d_cl_wgrad_workspace_bytes(int B, int T_out, int C, int K) {
    if (B <= 0 || T_out <= 0 || C <= 0 || K <= 0) return 0;
    return (size_t)B * ((T_out + pafc::WG_SPAN - 1) / pafc::WG_SPAN) * (K + 1) * C * sizeof(float);
}

extern "C" int pafc_dwconv1d_cl_wgrad(int dtype, int B, int T_in, int C, int K, int left_pad, int T_out, const void *x,
                                      long ldx, const void *dy, float *dw, float *dbias, void *workspace,
                                      size_t workspace_bytes, pafc_stream_t stream) {
    if (!x || !dy || !dw || !workspace) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T_in <= 0 || T_out <= 0 || C <= 0 || (C % 2) || K <= 0 || K > pafc::KMAX || left_pad < 0 ||
        B > 65535 || ldx < C || (ldx % 2))
        return PAFC_ERR_BAD_DIMS;
    if (workspace_bytes < pafc_dwconv1d_cl_wgrad_workspace_bytes(B, T_out, C, K)) return PAFC_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)workspace;
    if (dtype == PAFC_BF16)
        return pafc::dispatch_wgrad<pafc::bf16_t>(K, B, T_in, C, left_pad, T_out, x, ldx, dy, part, dw, dbias, s);
    if (dtype == PAFC_F32) return pafc::dispatch_wgrad<float>(K, B, T_in, C, left_pad, T_out, x, ldx, dy, part, dw, dbias, s);
    return PAFC_ERR_DTYPE;
}
