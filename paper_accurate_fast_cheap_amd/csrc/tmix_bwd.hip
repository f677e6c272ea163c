// Backward of the time-mix block's element-wise chain for the training step (config c4), gfx950.
// C ABI: include/pafc_encoder_ops.h: pafc_tmix_shift_mix_bwd, pafc_tmix_mix4_bwd.
//
// Forward (src/model.py:274-284; kernels tmix_shift_mix_kernel / tmix_mix4_kernel in glue.hip):
//     xx = shift(x) - x          shift: x_{t-1} (zero at t = 0); reversed time: x_{t+1} (zero at T-1)
//     xxx = x + xx * maa_x
//     z_q = x + xx * (maa_q + m_q),  q in {r, k, v, w}
// Under autograd the framework runs this as ~16 element-wise kernels forward and ~40 backward (each a full (B, T, C)
// round trip, five of them row reductions for the maa gradients): 26 ms of the 94 ms c4 step together with the LoRA
// products.  Here each of the two groups is ONE pass: a wave owns a row (lane = 8 channels per 512), walks the
// ROWS rows of its block keeping the per-channel maa sums in registers, and a second kernel adds the blocks' partials in
// a fixed order (deterministic).  The transposed shift -- x_t also fed row t + 1 (t - 1 when reversed) -- is taken by
// reading that neighbour row's incoming gradients as well (L2-hot: the neighbouring wave streams the same row).
//     dxxx -> dx = dxxx (1 - maa_x) + [dxxx maa_x]_{next},              dmaa_x = sum_rows dxxx xx
//     dz_q -> dm_q = dz_q xx,  dmaa_q = sum_rows dz_q xx,  dxx = sum_q dz_q (maa_q + m_q),
//             dx = sum_q dz_q - dxx + [dxx]_{next}
// Sums in fp32, one rounding to the element type on the way out.
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int TV = 8;            // channels per lane per iteration
constexpr int TMAXIT = 2;        // C <= 1024
constexpr int TROWS = 16;        // rows per block (4 per wave)

template <typename ET> __device__ __forceinline__ void tl8(const ET *p, float *f);
template <> __device__ __forceinline__ void tl8<bf16_t>(const bf16_t *p, float *f) {
    Elem<bf16_t>::unpack(*reinterpret_cast<const uint4 *>(p), f);
}
template <> __device__ __forceinline__ void tl8<float>(const float *p, float *f) {
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename ET> __device__ __forceinline__ void ts8(ET *p, const float *f);
template <> __device__ __forceinline__ void ts8<bf16_t>(bf16_t *p, const float *f) {
    uint4 q;
    q.x = f32_to_bf16_bits(f[0]) | (f32_to_bf16_bits(f[1]) << 16);
    q.y = f32_to_bf16_bits(f[2]) | (f32_to_bf16_bits(f[3]) << 16);
    q.z = f32_to_bf16_bits(f[4]) | (f32_to_bf16_bits(f[5]) << 16);
    q.w = f32_to_bf16_bits(f[6]) | (f32_to_bf16_bits(f[7]) << 16);
    *reinterpret_cast<uint4 *>(p) = q;
}
template <> __device__ __forceinline__ void ts8<float>(float *p, const float *f) {
    reinterpret_cast<float4 *>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4 *>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}

struct MixBwdArgs {
    const void *x, *m, *maa;          // x (rows, C); m (NQ, rows, C) or null; maa (NQ, C)
    const void *dz[4];                // NQ incoming gradients (rows, C)
    void *dx, *dm;                    // dx (rows, C); dm (NQ, rows, C) -- or (rows, NQ, C): dm_qs / dm_rs -- or null
    long dm_qs, dm_rs;                // element strides of dm between q and between rows
    float *part;                      // [nblk][NQ][C]
    long rows;
    int T, C, reverse;
};

// NQ = 1: the first lerp (m = 0, no dm);  NQ = 4: the four data-dependent lerps
template <typename ET, int NQ>
__global__ __launch_bounds__(256) void tmix_mix_bwd_kernel(const MixBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C, T = a.T;
    const ET *x = (const ET *)a.x;
    float acc[NQ][TMAXIT][TV];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int it = 0; it < TMAXIT; ++it)
#pragma unroll
            for (int e = 0; e < TV; ++e) acc[q][it][e] = 0.f;

    const long r_end = min(a.rows, ((long)blockIdx.x + 1) * TROWS);
    for (long row = (long)blockIdx.x * TROWS + wave; row < r_end; row += 4) {
        const int t = (int)(row % T);
        const bool has_prev = a.reverse ? (t < T - 1) : (t > 0);      // the row whose x is this row's shift(x)
        const bool has_next = a.reverse ? (t > 0) : (t < T - 1);      // the row that takes this row's x as ITS shift(x)
        const long prow = a.reverse ? row + 1 : row - 1, nrow = a.reverse ? row - 1 : row + 1;
#pragma unroll
        for (int it = 0; it < TMAXIT; ++it) {
            const int c = (it * 64 + lane) * TV;
            if (c < C) {
                float xc[TV], xp[TV], xx[TV], sum_dz[TV], dxx[TV], dxx_n[TV];
                tl8<ET>(x + row * C + c, xc);
                tl8<ET>(x + (has_prev ? prow : row) * C + c, xp);
#pragma unroll
                for (int e = 0; e < TV; ++e) {
                    xx[e] = (has_prev ? xp[e] : 0.f) - xc[e];
                    sum_dz[e] = 0.f; dxx[e] = 0.f; dxx_n[e] = 0.f;
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float g[TV], gn[TV], mq[TV], mn[TV], av[TV];
                    const ET *dzq = (const ET *)a.dz[q];
                    tl8<ET>(dzq + row * C + c, g);
                    tl8<ET>(dzq + (has_next ? nrow : row) * C + c, gn);
                    tl8<ET>((const ET *)a.maa + (size_t)q * C + c, av);
                    if (NQ > 1) {
                        tl8<ET>((const ET *)a.m + ((size_t)q * a.rows + row) * C + c, mq);
                        tl8<ET>((const ET *)a.m + ((size_t)q * a.rows + (has_next ? nrow : row)) * C + c, mn);
                    }
                    float dmq[TV];
#pragma unroll
                    for (int e = 0; e < TV; ++e) {
                        const float w = av[e] + (NQ > 1 ? mq[e] : 0.f);
                        const float wn = av[e] + (NQ > 1 ? mn[e] : 0.f);
                        dmq[e] = g[e] * xx[e];
                        acc[q][it][e] += dmq[e];
                        sum_dz[e] += g[e];
                        dxx[e] = fmaf(g[e], w, dxx[e]);
                        dxx_n[e] = fmaf(has_next ? gn[e] : 0.f, wn, dxx_n[e]);
                    }
                    if (NQ > 1) ts8<ET>((ET *)a.dm + (size_t)q * a.dm_qs + (size_t)row * a.dm_rs + c, dmq);
                }
                float o[TV];
#pragma unroll
                for (int e = 0; e < TV; ++e) o[e] = sum_dz[e] - dxx[e] + dxx_n[e];
                ts8<ET>((ET *)a.dx + row * C + c, o);
            }
        }
    }
    // the four waves' maa sums -> one partial per (q, channel) for this block
    __shared__ float s_red[4][TV * 64 * TMAXIT];
    float *po = a.part + (size_t)blockIdx.x * NQ * C;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < TMAXIT; ++it)
#pragma unroll
            for (int e = 0; e < TV; ++e) s_red[wave][(it * 64 + lane) * TV + e] = acc[q][it][e];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256)
            po[(size_t)q * C + c] = s_red[0][c] + s_red[1][c] + s_red[2][c] + s_red[3][c];
    }
}

// out[i] = sum over the nblk partials (i over n entries): 16 entries x 64 slices per block, fixed order
__global__ __launch_bounds__(1024) void tmix_bwd_reduce_kernel(int n, int nblk, const float *__restrict__ part,
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

template <int NQ>
int launch_mix_bwd(int dtype, const MixBwdArgs &a, float *dmaa, hipStream_t s) {
    const int nblk = (int)((a.rows + TROWS - 1) / TROWS);
    if (dtype == PAFC_BF16)
        hipLaunchKernelGGL((tmix_mix_bwd_kernel<bf16_t, NQ>), dim3(nblk), dim3(256), 0, s, a);
    else if (dtype == PAFC_F32)
        hipLaunchKernelGGL((tmix_mix_bwd_kernel<float, NQ>), dim3(nblk), dim3(256), 0, s, a);
    else
        return PAFC_ERR_DTYPE;
    hipLaunchKernelGGL(tmix_bwd_reduce_kernel, dim3((NQ * a.C + 15) / 16), dim3(1024), 0, s, NQ * a.C, nblk, a.part, dmaa);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

bool mix_dims_ok(int B, int T, int C) {
    return B > 0 && T > 0 && C > 0 && C % TV == 0 && C <= 64 * TV * TMAXIT && (long)B * T <= 0x7fffffffL * (long)TROWS;
}

}  // namespace
}  // namespace pafc

extern "C" size_t pafc_tmix_bwd_workspace_bytes(long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return (size_t)((rows + pafc::TROWS - 1) / pafc::TROWS) * 4 * C * sizeof(float);
}

extern "C" int pafc_tmix_shift_mix_bwd(int dtype, int B, int T, int C, int reverse, const void *x, const void *maa_x,
                                       const void *dxxx, void *dx, float *dmaa_x, void *workspace, size_t workspace_bytes,
                                       pafc_stream_t stream) {
    if (!x || !maa_x || !dxxx || !dx || !dmaa_x || !workspace) return PAFC_ERR_NULL_POINTER;
    if (!pafc::mix_dims_ok(B, T, C)) return PAFC_ERR_BAD_DIMS;
    const long rows = (long)B * T;
    if (workspace_bytes < pafc_tmix_bwd_workspace_bytes(rows, C)) return PAFC_ERR_WORKSPACE;
    pafc::MixBwdArgs a{};
    a.x = x; a.m = nullptr; a.maa = maa_x; a.dz[0] = dxxx; a.dx = dx; a.dm = nullptr; a.part = (float *)workspace;
    a.rows = rows; a.T = T; a.C = C; a.reverse = reverse ? 1 : 0;
    return pafc::launch_mix_bwd<1>(dtype, a, dmaa_x, (hipStream_t)stream);
}

static int mix4_bwd_launch(int dtype, int B, int T, int C, int reverse, const void *x, const void *m, const void *maa,
                           const void *dz_r, const void *dz_k, const void *dz_v, const void *dz_w, void *dx, void *dm, bool dm_by_rows,
                           float *dmaa, void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    if (!x || !m || !maa || !dz_r || !dz_k || !dz_v || !dz_w || !dx || !dm || !dmaa || !workspace) return PAFC_ERR_NULL_POINTER;
    if (!pafc::mix_dims_ok(B, T, C)) return PAFC_ERR_BAD_DIMS;
    const long rows = (long)B * T;
    if (workspace_bytes < pafc_tmix_bwd_workspace_bytes(rows, C)) return PAFC_ERR_WORKSPACE;
    pafc::MixBwdArgs a{};
    a.x = x; a.m = m; a.maa = maa; a.dz[0] = dz_r; a.dz[1] = dz_k; a.dz[2] = dz_v; a.dz[3] = dz_w; a.dx = dx; a.dm = dm;
    a.dm_qs = dm_by_rows ? C : rows * C;
    a.dm_rs = dm_by_rows ? 4L * C : C;
    a.part = (float *)workspace; a.rows = rows; a.T = T; a.C = C; a.reverse = reverse ? 1 : 0;
    return pafc::launch_mix_bwd<4>(dtype, a, dmaa, (hipStream_t)stream);
}

extern "C" int pafc_tmix_mix4_bwd(int dtype, int B, int T, int C, int reverse, const void *x, const void *m, const void *maa,
                                  const void *dz_r, const void *dz_k, const void *dz_v, const void *dz_w, void *dx, void *dm,
                                  float *dmaa, void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    return mix4_bwd_launch(dtype, B, T, C, reverse, x, m, maa, dz_r, dz_k, dz_v, dz_w, dx, dm, false, dmaa, workspace, workspace_bytes,
                           stream);
}

// the same with dm laid out (B*T, 4, C): the LoRA-up matrices' gradients then read it as ONE (B*T, 4 C) operand
extern "C" int pafc_tmix_mix4_bwd_rows(int dtype, int B, int T, int C, int reverse, const void *x, const void *m, const void *maa,
                                       const void *dz_r, const void *dz_k, const void *dz_v, const void *dz_w, void *dx, void *dm,
                                       float *dmaa, void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    return mix4_bwd_launch(dtype, B, T, C, reverse, x, m, maa, dz_r, dz_k, dz_v, dz_w, dx, dm, true, dmaa, workspace, workspace_bytes,
                           stream);
}
