// LayerNorm backward for the training step (config c4), gfx950.  C ABI: include/pafc_encoder_ops.h: pafc_layernorm_bwd.
//
// The layer has seven LayerNorms (encoder_layer.py:201-259, convolution.py:136, src/model.py:323); under autograd the
// framework runs three kernels per norm in the backward (input gradient, partial gamma/beta sums, their reduction) on
// fp32 tensors, 8 % of the c4 step.  Here: one pass over (x, dy) with one wave per row -- the same ownership as the
// forward kernel (add_layernorm_kernel), mean / rstd recomputed from x in registers rather than stored -- that writes
// dx and keeps the per-channel sums of dy * xhat and dy in registers across the ROWS_PER_BLOCK rows of its block; a
// block leaves one fp32 partial per channel, a second kernel adds the partials in a fixed order (deterministic).
//   xhat = (x - mean) * rstd,  g = dy * gamma
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  dgamma = sum_rows dy * xhat,  dbeta = sum_rows dy
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int LVEC = 8;        // channels per lane per iteration
constexpr int LMAXIT = 2;      // => C <= 64 * 8 * 2 = 1024, as the forward kernel
constexpr int ROWS_PER_BLOCK = 16;   // 4 rows per wave: ~4 blocks per CU in flight at the c4 shape (16 000 rows)

template <typename ET> __device__ __forceinline__ void ld8(const ET *p, float *f);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t *p, float *f) {
    Elem<bf16_t>::unpack(*reinterpret_cast<const uint4 *>(p), f);
}
template <> __device__ __forceinline__ void ld8<float>(const float *p, float *f) {
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename ET> __device__ __forceinline__ void st8(ET *p, const float *f);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t *p, const float *f) {
    uint4 q;
    q.x = f32_to_bf16_bits(f[0]) | (f32_to_bf16_bits(f[1]) << 16);
    q.y = f32_to_bf16_bits(f[2]) | (f32_to_bf16_bits(f[3]) << 16);
    q.z = f32_to_bf16_bits(f[4]) | (f32_to_bf16_bits(f[5]) << 16);
    q.w = f32_to_bf16_bits(f[6]) | (f32_to_bf16_bits(f[7]) << 16);
    *reinterpret_cast<uint4 *>(p) = q;
}
template <> __device__ __forceinline__ void st8<float>(float *p, const float *f) {
    reinterpret_cast<float4 *>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4 *>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}

// EX: dtype of x, gamma and dx; ED: dtype of dy
template <typename EX, typename ED>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(long rows, int C, const EX *__restrict__ x,
                                                            const ED *__restrict__ dy, const EX *__restrict__ gamma,
                                                            float eps, EX *__restrict__ dx, float *__restrict__ part,
                                                            const EX *__restrict__ dx_add) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.f / (float)C;
    float gm[LMAXIT][LVEC], ag[LMAXIT][LVEC], ab[LMAXIT][LVEC];
#pragma unroll
    for (int it = 0; it < LMAXIT; ++it) {
        const int c = (it * 64 + lane) * LVEC;
#pragma unroll
        for (int e = 0; e < LVEC; ++e) { gm[it][e] = 0.f; ag[it][e] = 0.f; ab[it][e] = 0.f; }
        if (c < C) ld8<EX>(gamma + c, gm[it]);
    }
    const long r_end = min(rows, ((long)blockIdx.x + 1) * ROWS_PER_BLOCK);
    for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < r_end; row += 4) {
        float xv[LMAXIT][LVEC], gv[LMAXIT][LVEC];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
                ld8<EX>(x + (size_t)row * C + c, xv[it]);
                ld8<ED>(dy + (size_t)row * C + c, gv[it]);
#pragma unroll
                for (int e = 0; e < LVEC; ++e) sum += xv[it][e];
            }
        }
        const float mean = wave_sum(sum) * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < LVEC; ++e) { const float d = xv[it][e] - mean; sq = fmaf(d, d, sq); }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) * inv_c + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < LVEC; ++e) {
                    const float xh = (xv[it][e] - mean) * rstd;
                    const float d = gv[it][e];
                    ag[it][e] = fmaf(d, xh, ag[it][e]);
                    ab[it][e] += d;
                    const float g = d * gm[it][e];
                    s1 += g;
                    s2 = fmaf(g, xh, s2);
                    xv[it][e] = xh;
                    gv[it][e] = g;
                }
            }
        }
        s1 = wave_sum(s1) * inv_c;
        s2 = wave_sum(s2) * inv_c;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
                float o[LVEC];
#pragma unroll
                for (int e = 0; e < LVEC; ++e) o[e] = rstd * (gv[it][e] - s1 - xv[it][e] * s2);
                if (dx_add != nullptr) {     // the gradient that reaches x past the norm (the residual path of a pre-norm branch)
                    float av[LVEC];
                    ld8<EX>(dx_add + (size_t)row * C + c, av);
#pragma unroll
                    for (int e = 0; e < LVEC; ++e) o[e] += av[e];
                }
                st8<EX>(dx + (size_t)row * C + c, o);
            }
        }
    }
    // the four waves' sums -> one partial per channel for this block: [block][0: dgamma, 1: dbeta][C]
    __shared__ float s_red[2][4][64 * LVEC * LMAXIT];
#pragma unroll
    for (int it = 0; it < LMAXIT; ++it)
#pragma unroll
        for (int e = 0; e < LVEC; ++e) {
            s_red[0][wave][(it * 64 + lane) * LVEC + e] = ag[it][e];
            s_red[1][wave][(it * 64 + lane) * LVEC + e] = ab[it][e];
        }
    __syncthreads();
    float *po = part + (size_t)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int k = i / C, c = i - k * C;
        po[i] = s_red[k][0][c] + s_red[k][1][c] + s_red[k][2][c] + s_red[k][3][c];
    }
}

// out[i] = sum over the nblk partials, i over 2 * C (dgamma then dbeta).  A block = 16 entries x 64 slices of the partial
// list (a thread adds ~nblk / 64 values); the slices are combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void layernorm_bwd_reduce_kernel(int n, int nblk, const float *__restrict__ part,
                                                                    float *__restrict__ out) {
    __shared__ float s_sum[64][16];
    const int ch = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + ch;
    float s = 0.f;
    if (i < n)
        for (int b = slice; b < nblk; b += 64) s += part[(size_t)b * n + i];
    s_sum[slice][ch] = s;
    __syncthreads();
    if (slice == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += s_sum[k][ch];
        out[i] = t;
    }
}

// ---- LayerNorm + SiLU of the conv module in the training step (convolution.py:136-138: `activation(norm(x))` between the
// depthwise convolution and pointwise_conv2).  Under bf16 autocast the framework runs this as cast (bf16 -> fp32), LayerNorm,
// SiLU, cast (fp32 -> bf16 at the projection) forward and the same four again backward, all over (B, T, C); here one kernel each
// way: x and y / dy and dx in the convolution's dtype (bf16), the arithmetic in fp32 against the norm's own (fp32 or bf16)
// parameters EG, the SAME values as the chain (one rounding, at the end).
//   z = xhat gamma + beta,  y = z sigmoid(z);   dz = dy sigmoid(z) (1 + z (1 - sigmoid(z))),  then LayerNorm's backward on dz.
template <typename EX, typename EG>
__global__ __launch_bounds__(256) void ln_silu_fwd_kernel(long rows, int C, const EX *__restrict__ x, const EG *__restrict__ gamma,
                                                          const EG *__restrict__ beta, float eps, EX *__restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.f / (float)C;
    float gm[LMAXIT][LVEC], bt[LMAXIT][LVEC];
#pragma unroll
    for (int it = 0; it < LMAXIT; ++it) {
        const int c = (it * 64 + lane) * LVEC;
#pragma unroll
        for (int e = 0; e < LVEC; ++e) { gm[it][e] = 0.f; bt[it][e] = 0.f; }
        if (c < C) { ld8<EG>(gamma + c, gm[it]); ld8<EG>(beta + c, bt[it]); }
    }
    const long r_end = min(rows, ((long)blockIdx.x + 1) * ROWS_PER_BLOCK);
    for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < r_end; row += 4) {
        float xv[LMAXIT][LVEC];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
                ld8<EX>(x + (size_t)row * C + c, xv[it]);
#pragma unroll
                for (int e = 0; e < LVEC; ++e) sum += xv[it][e];
            }
        }
        const float mean = wave_sum(sum) * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < LVEC; ++e) { const float d = xv[it][e] - mean; sq = fmaf(d, d, sq); }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) * inv_c + eps);
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
                float o[LVEC];
#pragma unroll
                for (int e = 0; e < LVEC; ++e) {
                    const float z = fmaf((xv[it][e] - mean) * rstd, gm[it][e], bt[it][e]);
                    o[e] = z / (1.f + __expf(-z));
                }
                st8<EX>(y + (size_t)row * C + c, o);
            }
        }
    }
}

template <typename EX, typename EG>
__global__ __launch_bounds__(256) void ln_silu_bwd_kernel(long rows, int C, const EX *__restrict__ x, const EX *__restrict__ dy,
                                                          const EG *__restrict__ gamma, const EG *__restrict__ beta, float eps,
                                                          EX *__restrict__ dx, float *__restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.f / (float)C;
    float gm[LMAXIT][LVEC], bt[LMAXIT][LVEC], ag[LMAXIT][LVEC], ab[LMAXIT][LVEC];
#pragma unroll
    for (int it = 0; it < LMAXIT; ++it) {
        const int c = (it * 64 + lane) * LVEC;
#pragma unroll
        for (int e = 0; e < LVEC; ++e) { gm[it][e] = 0.f; bt[it][e] = 0.f; ag[it][e] = 0.f; ab[it][e] = 0.f; }
        if (c < C) { ld8<EG>(gamma + c, gm[it]); ld8<EG>(beta + c, bt[it]); }
    }
    const long r_end = min(rows, ((long)blockIdx.x + 1) * ROWS_PER_BLOCK);
    for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < r_end; row += 4) {
        float xv[LMAXIT][LVEC], gv[LMAXIT][LVEC];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
                ld8<EX>(x + (size_t)row * C + c, xv[it]);
                ld8<EX>(dy + (size_t)row * C + c, gv[it]);
#pragma unroll
                for (int e = 0; e < LVEC; ++e) sum += xv[it][e];
            }
        }
        const float mean = wave_sum(sum) * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < LVEC; ++e) { const float d = xv[it][e] - mean; sq = fmaf(d, d, sq); }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) * inv_c + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < LVEC; ++e) {
                    const float xh = (xv[it][e] - mean) * rstd;
                    const float z = fmaf(xh, gm[it][e], bt[it][e]);
                    const float sg = 1.f / (1.f + __expf(-z));
                    const float d = gv[it][e] * (sg * fmaf(z, 1.f - sg, 1.f));     // dL/dz
                    ag[it][e] = fmaf(d, xh, ag[it][e]);
                    ab[it][e] += d;
                    const float g = d * gm[it][e];
                    s1 += g;
                    s2 = fmaf(g, xh, s2);
                    xv[it][e] = xh;
                    gv[it][e] = g;
                }
            }
        }
        s1 = wave_sum(s1) * inv_c;
        s2 = wave_sum(s2) * inv_c;
#pragma unroll
        for (int it = 0; it < LMAXIT; ++it) {
            const int c = (it * 64 + lane) * LVEC;
            if (c < C) {
                float o[LVEC];
#pragma unroll
                for (int e = 0; e < LVEC; ++e) o[e] = rstd * (gv[it][e] - s1 - xv[it][e] * s2);
                st8<EX>(dx + (size_t)row * C + c, o);
            }
        }
    }
    __shared__ float s_red[2][4][64 * LVEC * LMAXIT];
#pragma unroll
    for (int it = 0; it < LMAXIT; ++it)
#pragma unroll
        for (int e = 0; e < LVEC; ++e) {
            s_red[0][wave][(it * 64 + lane) * LVEC + e] = ag[it][e];
            s_red[1][wave][(it * 64 + lane) * LVEC + e] = ab[it][e];
        }
    __syncthreads();
    float *po = part + (size_t)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int k = i / C, c = i - k * C;
        po[i] = s_red[k][0][c] + s_red[k][1][c] + s_red[k][2][c] + s_red[k][3][c];
    }
}

template <typename EX, typename ED>
int launch_ln_bwd(long rows, int C, const void *x, const void *dy, const void *gamma, float eps, void *dx, float *dgb,
                  float *part, const void *dx_add, hipStream_t s) {
    const int nblk = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    hipLaunchKernelGGL((layernorm_bwd_kernel<EX, ED>), dim3(nblk), dim3(256), 0, s, rows, C, (const EX *)x, (const ED *)dy,
                       (const EX *)gamma, eps, (EX *)dx, part, (const EX *)dx_add);
    hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * C + 15) / 16), dim3(1024), 0, s, 2 * C, nblk, part, dgb);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // namespace
}  // namespace pafc

extern "C" size_t pafc_layernorm_bwd_workspace_bytes(long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return (size_t)((rows + pafc::ROWS_PER_BLOCK - 1) / pafc::ROWS_PER_BLOCK) * 2 * C * sizeof(float);
}

// ... + dx_add (dtype_x, or null): dx = LayerNorm's input gradient + dx_add -- the gradient that reaches x past the norm (the residual
// path of a pre-norm branch, encoder_layer.py:201-256), which autograd would add in a pass of its own
extern "C" int pafc_layernorm_bwd_add(int dtype_x, int dtype_dy, long rows, int C, const void *x, const void *dy,
                                      const void *gamma, float eps, const void *dx_add, void *dx, float *dgamma_dbeta,
                                      void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    if (!x || !dy || !gamma || !dx || !dgamma_dbeta || !workspace) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || C <= 0 || C % pafc::LVEC || C > 64 * pafc::LVEC * pafc::LMAXIT || rows > 0x7fffffffL * pafc::ROWS_PER_BLOCK)
        return PAFC_ERR_BAD_DIMS;
    if (workspace_bytes < pafc_layernorm_bwd_workspace_bytes(rows, C)) return PAFC_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)workspace;
    using pafc::bf16_t;
    if (dtype_x == PAFC_F32 && dtype_dy == PAFC_F32)
        return pafc::launch_ln_bwd<float, float>(rows, C, x, dy, gamma, eps, dx, dgamma_dbeta, part, dx_add, s);
    if (dtype_x == PAFC_F32 && dtype_dy == PAFC_BF16)
        return pafc::launch_ln_bwd<float, bf16_t>(rows, C, x, dy, gamma, eps, dx, dgamma_dbeta, part, dx_add, s);
    if (dtype_x == PAFC_BF16 && dtype_dy == PAFC_BF16)
        return pafc::launch_ln_bwd<bf16_t, bf16_t>(rows, C, x, dy, gamma, eps, dx, dgamma_dbeta, part, dx_add, s);
    if (dtype_x == PAFC_BF16 && dtype_dy == PAFC_F32)
        return pafc::launch_ln_bwd<bf16_t, float>(rows, C, x, dy, gamma, eps, dx, dgamma_dbeta, part, dx_add, s);
    return PAFC_ERR_DTYPE;
}

extern "C" int pafc_layernorm_bwd(int dtype_x, int dtype_dy, long rows, int C, const void *x, const void *dy,
                                  const void *gamma, float eps, void *dx, float *dgamma_dbeta, void *workspace,
                                  size_t workspace_bytes, pafc_stream_t stream) {
    return pafc_layernorm_bwd_add(dtype_x, dtype_dy, rows, C, x, dy, gamma, eps, nullptr, dx, dgamma_dbeta, workspace, workspace_bytes,
                                  stream);
}

/* LayerNorm + SiLU of the conv module, training step: see the kernels.  dtype_x: x, y, dy, dx; dtype_g: gamma, beta. */
extern "C" int pafc_layernorm_silu_fwd(int dtype_x, int dtype_g, long rows, int C, const void *x, const void *gamma, const void *beta,
                                       float eps, void *y, pafc_stream_t stream) {
    if (!x || !gamma || !beta || !y) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || C <= 0 || C % pafc::LVEC || C > 64 * pafc::LVEC * pafc::LMAXIT || rows > 0x7fffffffL * pafc::ROWS_PER_BLOCK)
        return PAFC_ERR_BAD_DIMS;
    const int nblk = (int)((rows + pafc::ROWS_PER_BLOCK - 1) / pafc::ROWS_PER_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    using pafc::bf16_t;
#define PAFC_LNS_F(EX, EG)                                                                                                 \
    hipLaunchKernelGGL((pafc::ln_silu_fwd_kernel<EX, EG>), dim3(nblk), dim3(256), 0, s, rows, C, (const EX *)x, (const EG *)gamma, \
                       (const EG *)beta, eps, (EX *)y)
    if (dtype_x == PAFC_BF16 && dtype_g == PAFC_F32) PAFC_LNS_F(bf16_t, float);
    else if (dtype_x == PAFC_BF16 && dtype_g == PAFC_BF16) PAFC_LNS_F(bf16_t, bf16_t);
    else if (dtype_x == PAFC_F32 && dtype_g == PAFC_F32) PAFC_LNS_F(float, float);
    else return PAFC_ERR_DTYPE;
#undef PAFC_LNS_F
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_layernorm_silu_bwd(int dtype_x, int dtype_g, long rows, int C, const void *x, const void *dy, const void *gamma,
                                       const void *beta, float eps, void *dx, float *dgamma_dbeta, void *workspace,
                                       size_t workspace_bytes, pafc_stream_t stream) {
    if (!x || !dy || !gamma || !beta || !dx || !dgamma_dbeta || !workspace) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || C <= 0 || C % pafc::LVEC || C > 64 * pafc::LVEC * pafc::LMAXIT || rows > 0x7fffffffL * pafc::ROWS_PER_BLOCK)
        return PAFC_ERR_BAD_DIMS;
    if (workspace_bytes < pafc_layernorm_bwd_workspace_bytes(rows, C)) return PAFC_ERR_WORKSPACE;
    const int nblk = (int)((rows + pafc::ROWS_PER_BLOCK - 1) / pafc::ROWS_PER_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)workspace;
    using pafc::bf16_t;
#define PAFC_LNS_B(EX, EG)                                                                                                 \
    hipLaunchKernelGGL((pafc::ln_silu_bwd_kernel<EX, EG>), dim3(nblk), dim3(256), 0, s, rows, C, (const EX *)x, (const EX *)dy,    \
                       (const EG *)gamma, (const EG *)beta, eps, (EX *)dx, part)
    if (dtype_x == PAFC_BF16 && dtype_g == PAFC_F32) PAFC_LNS_B(bf16_t, float);
    else if (dtype_x == PAFC_BF16 && dtype_g == PAFC_BF16) PAFC_LNS_B(bf16_t, bf16_t);
    else if (dtype_x == PAFC_F32 && dtype_g == PAFC_F32) PAFC_LNS_B(float, float);
    else return PAFC_ERR_DTYPE;
#undef PAFC_LNS_B
    hipLaunchKernelGGL(pafc::layernorm_bwd_reduce_kernel, dim3((2 * C + 15) / 16), dim3(1024), 0, s, 2 * C, nblk, part, dgamma_dbeta);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
