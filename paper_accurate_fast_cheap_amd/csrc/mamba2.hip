// Glue of the Mamba-2 block around the chunked scan (C ABI: include/pafc_encoder_ops.h: pafc_mamba2_prep / _finish).
//
// PARITY UNPINNED: the reference only wraps the third-party mamba_ssm.modules.mamba2.Mamba2 (mamba_att_wrapper.py:24-35,
// mamba2_bidirectional.py:12-36); these kernels fuse the restatement in transformer/mamba2.py (its module docstring has
// the algebra), which is also what the tests compare them with.
//
//   prep:   from the conv + SiLU output xBC = [x (H*64) | B (128) | C (128)] and the raw dt columns of in_proj's output:
//           dt = softplus(dt_raw + dt_bias), a_{t+1} = exp(dt_{t+1} A), A = -exp(A_log)   (per head and step)
//           -> the six fp32 operand planes of the two scans (d_state 128 = two 64-wide halves):
//              r_half = C_half (same for every head), k_half = a_{t+1} B_half, v = dt x, w = log(-log a_{t+1})
//           (~15 framework kernels and as many (L, 1024) fp32 temporaries in the op-by-op version)
//   finish: y = y_0 + y_1 + ((B . C) dt + D) x;  y * SiLU(z);  RMSNorm over the d_inner channels * weight   (one pass)
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

__device__ __forceinline__ float softplus_(float x) { return x > 20.f ? x : log1pf(__expf(x)); }   // torch: beta 1, threshold 20

template <typename ET>
__global__ __launch_bounds__(256) void mamba2_prep_kernel(int L, int d_inner, const ET *xbc, const ET *dt_raw, long ld_dt,
                                                          const float *dt_bias, const float *A_log, float *r0, float *r1,
                                                          float *k0, float *k1, float *v, float *w) {
    const long row = blockIdx.x;                 // b * L + t
    const int t = (int)(row % L);
    const int ldx = d_inner + 256;
    const ET *xr = xbc + row * ldx;
    const ET *Bp = xr + d_inner, *Cp = Bp + 128;
    for (int idx = threadIdx.x * 4; idx < d_inner; idx += 1024) {
        const int h = idx >> 6, j = idx & 63;
        const float A = -__expf(A_log[h]);
        const float dt = softplus_(Elem<ET>::load(dt_raw + row * ld_dt + h) + dt_bias[h]);
        float nxt = 0.f;                          // log a_{t+1}; the last step's value never reaches an output
        if (t + 1 < L) nxt = softplus_(Elem<ET>::load(dt_raw + (row + 1) * ld_dt + h) + dt_bias[h]) * A;
        const float a_next = __expf(nxt);
        const float wl = __logf(fmaxf(-nxt, 1e-30f));
        float4 o_r0, o_r1, o_k0, o_k1, o_v;
        float *pr0 = &o_r0.x, *pr1 = &o_r1.x, *pk0 = &o_k0.x, *pk1 = &o_k1.x, *pv = &o_v.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b0 = Elem<ET>::load(Bp + j + e), b1 = Elem<ET>::load(Bp + 64 + j + e);
            pr0[e] = Elem<ET>::load(Cp + j + e);
            pr1[e] = Elem<ET>::load(Cp + 64 + j + e);
            pk0[e] = a_next * b0;
            pk1[e] = a_next * b1;
            pv[e] = Elem<ET>::load(xr + idx + e) * dt;
        }
        const long o = row * d_inner + idx;
        *reinterpret_cast<float4 *>(r0 + o) = o_r0;
        *reinterpret_cast<float4 *>(r1 + o) = o_r1;
        *reinterpret_cast<float4 *>(k0 + o) = o_k0;
        *reinterpret_cast<float4 *>(k1 + o) = o_k1;
        *reinterpret_cast<float4 *>(v + o) = o_v;
        *reinterpret_cast<float4 *>(w + o) = make_float4(wl, wl, wl, wl);
    }
}

__device__ __forceinline__ float block_sum(float x, float *red) {
    x = wave_sum(x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename ET>
__global__ __launch_bounds__(256) void mamba2_finish_kernel(int L, int d_inner, const float *y0, const float *y1,
                                                            const ET *xbc, const ET *dt_raw, long ld_dt, const ET *z,
                                                            long ld_z, const float *dt_bias, const float *Dp,
                                                            const ET *norm_w, float eps, int diag, ET *out) {
    __shared__ float red[4];
    using E = Elem<ET>;
    const long row = blockIdx.x;
    const int ldx = d_inner + 256;
    const ET *xr = xbc + row * ldx;
    const ET *Bp = xr + d_inner, *Cp = Bp + 128;
    float bc = 0.f;
    if (diag && threadIdx.x < 128) bc = E::load(Bp + threadIdx.x) * E::load(Cp + threadIdx.x);
    bc = block_sum(bc, red);                      // (B . C) over the 128 state dimensions (0 when the scan had the s = t term)
    float g[4];                                   // gated values of this thread (d_inner <= 1024: one pass of 4 each)
    float ss = 0.f;
    const int idx = threadIdx.x * 4;
    const bool on = idx < d_inner;
    if (on) {
        const int h = idx >> 6;
        const float dt = softplus_(E::load(dt_raw + row * ld_dt + h) + dt_bias[h]);
        const float4 a = *reinterpret_cast<const float4 *>(y0 + row * d_inner + idx);
        const float4 b = y1 ? *reinterpret_cast<const float4 *>(y1 + row * d_inner + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float ya[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = E::load(xr + idx + e);
            const float y = E::round(ya[e] + bc * (x * dt) + x * Dp[h]);       // y.to(z.dtype)
            const float zz = E::load(z + row * ld_z + idx + e);
            const float s = E::round(zz * __builtin_amdgcn_rcpf(1.f + __expf(-zz)));   // F.silu(z) in the model dtype
            g[e] = E::round(y * s);                                             // (x * silu(z)), then .float()
            ss += g[e] * g[e];
        }
    }
    ss = block_sum(ss, red);
    const float rs = rsqrtf(ss / (float)d_inner + eps);
    if (on) {
#pragma unroll
        for (int e = 0; e < 4; ++e) E::store(out + row * d_inner + idx + e, g[e] * rs * E::load(norm_w + idx + e));
    }
}


// out = RMSNorm(y * silu(z)) * w for a scan output that already carries the skip term (pafc_mamba2_scan_skip_bf16): one
// wave per row, the row (d_inner <= 1024 = 16 values per lane) stays in registers between the two passes.
template <typename ET>
__global__ __launch_bounds__(256) void mamba2_gate_norm_kernel(long rows, int d_inner, const ET *__restrict__ y,
                                                               const ET *__restrict__ z, long ld_z,
                                                               const ET *__restrict__ norm_w, float eps, ET *__restrict__ out) {
    using E = Elem<ET>;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float g[2][8];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * 64 + lane) * 8;
        if (c < d_inner) {
            float yv[8], zv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { yv[e] = E::load(y + row * d_inner + c + e); zv[e] = E::load(z + row * ld_z + c + e); }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = E::round(zv[e] * __builtin_amdgcn_rcpf(1.f + __expf(-zv[e])));   // F.silu(z) in the model dtype
                g[it][e] = E::round(yv[e] * s);
                ss = fmaf(g[it][e], g[it][e], ss);
            }
        }
    }
    const float rs = rsqrtf(wave_sum(ss) / (float)d_inner + eps);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * 64 + lane) * 8;
        if (c < d_inner) {
#pragma unroll
            for (int e = 0; e < 8; ++e) E::store(out + row * d_inner + c + e, g[it][e] * rs * E::load(norm_w + c + e));
        }
    }
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_mamba2_prep(int dtype, int B, int L, int d_inner, const void *xbc, const void *dt_raw, long ld_dt,
                                const float *dt_bias, const float *A_log, float *r0, float *r1, float *k0, float *k1,
                                float *v, float *w, pafc_stream_t stream) {
    if (!xbc || !dt_raw || !dt_bias || !A_log || !r0 || !r1 || !k0 || !k1 || !v || !w) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || L <= 0 || d_inner <= 0 || d_inner % 64 || ld_dt < d_inner / 64) return PAFC_ERR_BAD_DIMS;
    const long rows = (long)B * L;
    if (rows > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_BF16)
        hipLaunchKernelGGL(pafc::mamba2_prep_kernel<pafc::bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, L, d_inner,
                           (const pafc::bf16_t *)xbc, (const pafc::bf16_t *)dt_raw, ld_dt, dt_bias, A_log, r0, r1, k0, k1, v, w);
    else if (dtype == PAFC_F32)
        hipLaunchKernelGGL(pafc::mamba2_prep_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, L, d_inner,
                           (const float *)xbc, (const float *)dt_raw, ld_dt, dt_bias, A_log, r0, r1, k0, k1, v, w);
    else
        return PAFC_ERR_DTYPE;
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_mamba2_finish(int dtype, int B, int L, int d_inner, const float *y0, const float *y1, const void *xbc,
                                  const void *dt_raw, long ld_dt, const void *z, long ld_z, const float *dt_bias,
                                  const float *D, const void *norm_weight, float eps, int diag, void *out,
                                  pafc_stream_t stream) {
    if (!y0 || !xbc || !dt_raw || !z || !dt_bias || !D || !norm_weight || !out) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || L <= 0 || d_inner <= 0 || d_inner % 64 || d_inner > 1024 || ld_dt < d_inner / 64 || ld_z < d_inner)
        return PAFC_ERR_BAD_DIMS;
    const long rows = (long)B * L;
    if (rows > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_BF16)
        hipLaunchKernelGGL(pafc::mamba2_finish_kernel<pafc::bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, L, d_inner, y0, y1,
                           (const pafc::bf16_t *)xbc, (const pafc::bf16_t *)dt_raw, ld_dt, (const pafc::bf16_t *)z, ld_z, dt_bias,
                           D, (const pafc::bf16_t *)norm_weight, eps, diag, (pafc::bf16_t *)out);
    else if (dtype == PAFC_F32)
        hipLaunchKernelGGL(pafc::mamba2_finish_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, L, d_inner, y0, y1,
                           (const float *)xbc, (const float *)dt_raw, ld_dt, (const float *)z, ld_z, dt_bias, D,
                           (const float *)norm_weight, eps, diag, (float *)out);
    else
        return PAFC_ERR_DTYPE;
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_mamba2_gate_norm(int dtype, long rows, int d_inner, const void *y, const void *z, long ld_z,
                                     const void *norm_weight, float eps, void *out, pafc_stream_t stream) {
    if (!y || !z || !norm_weight || !out) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || d_inner <= 0 || d_inner % 8 || d_inner > 1024 || ld_z < d_inner || rows > 4L * 0x7fffffffL)
        return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == PAFC_BF16)
        hipLaunchKernelGGL(pafc::mamba2_gate_norm_kernel<pafc::bf16_t>, grid, block, 0, s, rows, d_inner, (const pafc::bf16_t *)y,
                           (const pafc::bf16_t *)z, ld_z, (const pafc::bf16_t *)norm_weight, eps, (pafc::bf16_t *)out);
    else if (dtype == PAFC_F32)
        hipLaunchKernelGGL(pafc::mamba2_gate_norm_kernel<float>, grid, block, 0, s, rows, d_inner, (const float *)y,
                           (const float *)z, ld_z, (const float *)norm_weight, eps, (float *)out);
    else
        return PAFC_ERR_DTYPE;
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
