// Element-wise groups of the TRAINING step as one kernel each (gfx950).  C ABI: include/pafc_encoder_ops.h
// (pafc_residual_dropout_fwd / _bwd, pafc_silu_dropout_fwd / _bwd).
//
// The layer's four residual branches are `x = x + dropout(branch)` / `x + ff_scale * dropout(branch)`
// (wenet/transformer/encoder_layer.py:205-206,232,247,254-255) and its two feed-forward modules apply
// `w_2(dropout(activation(w_1 x)))` (positionwise_feed_forward.py:47-55).  Under bf16 autocast the branch is bf16 and the
// residual stream fp32, so the framework runs dropout, a scale, a cast and an add as separate kernels forward and a masked scale, a
// scale and casts backward: ~500 activation-sized launches per c4 step.  Here each group is ONE pass forward and ONE backward:
//
//   out = x + (scale / (1 - p)) * keep * y          keep = [u >= p], u = uniform(seed, offset, element index)
//   dy  = (scale / (1 - p)) * keep * dout           (dx = dout: no kernel)
//   out = silu(h) * keep / (1 - p),   dh = dout * keep / (1 - p) * silu'(h)
//
// The keep mask is never stored: it is a counter-based generator (Philox-4x32 with 7 rounds: 128 random bits per
// four-element group) of (seed, offset, group index), recomputed in the backward pass.  seed = torch's CPU seed, offset = a
// per-call counter kept by the host wrapper: a step is reproducible under torch.manual_seed.  Rounding: the forward rounds once
// to the output dtype; the reference's op chain (dropout in bf16, scale in bf16, add in fp32) rounds twice more -- dropout
// rates and scales here (0.1, 0.5) make those exact or sub-ulp; the eval path (p = 0 or not training) does not come here.
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

__device__ __forceinline__ void philox4x32_7(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                             unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep flags of the 8 elements [8 g, 8 g + 8): two Philox calls' worth would be wasteful -- 16 bits per element from ONE call
__device__ __forceinline__ void keep8(unsigned long long seed, unsigned long long offset, unsigned long long g, unsigned thr16,
                                      bool (&keep)[8]) {
    unsigned r[4];
    philox4x32_7((unsigned)g, (unsigned)(g >> 32), (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed,
                 (unsigned)(seed >> 32), r);
#pragma unroll
    for (int e = 0; e < 8; ++e) keep[e] = ((r[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= thr16;
}

template <typename T> __device__ __forceinline__ void ld8(const T *p, float *f);
template <> __device__ __forceinline__ void ld8<float>(const float *p, float *f) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t *p, float *f) {
    Elem<bf16_t>::unpack(*reinterpret_cast<const uint4 *>(p), f);
}
template <typename T> __device__ __forceinline__ void st8(T *p, const float *f);
template <> __device__ __forceinline__ void st8<float>(float *p, const float *f) {
    *reinterpret_cast<float4 *>(p) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t *p, const float *f) {
    uint4 w;
    w.x = f32_to_bf16_bits(f[0]) | (f32_to_bf16_bits(f[1]) << 16);
    w.y = f32_to_bf16_bits(f[2]) | (f32_to_bf16_bits(f[3]) << 16);
    w.z = f32_to_bf16_bits(f[4]) | (f32_to_bf16_bits(f[5]) << 16);
    w.w = f32_to_bf16_bits(f[6]) | (f32_to_bf16_bits(f[7]) << 16);
    *reinterpret_cast<uint4 *>(p) = w;
}

// MODE 0: out = x + s * keep * y      (x: TX, y: TY, out: TX)
// MODE 1: dy  = s * keep * dout       (dout: TX -> a, dy: TY -> out)
template <typename TX, typename TY, int MODE>
__global__ __launch_bounds__(256) void residual_dropout_kernel(long n8, const TX *x, const TY *y, void *out, float s, unsigned thr16,
                                                               unsigned long long seed, unsigned long long offset) {
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n8; g += (long)gridDim.x * 256) {
        bool keep[8];
        if (thr16) keep8(seed, offset, (unsigned long long)g, thr16, keep);
        float a[8], o[8];
        if constexpr (MODE == 0) {
            float b[8];
            ld8<TX>(x + 8 * g, a);
            ld8<TY>(y + 8 * g, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (!thr16 || keep[e]) ? fmaf(s, b[e], a[e]) : a[e];
            st8<TX>((TX *)out + 8 * g, o);
        } else {
            ld8<TX>(x + 8 * g, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (!thr16 || keep[e]) ? s * a[e] : 0.f;
            st8<TY>((TY *)out + 8 * g, o);
        }
    }
}

// MODE 0: out = silu(h) * keep * s;  MODE 1: dh = dout * keep * s * silu'(h)      (all T)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void silu_dropout_kernel(long n8, const T *h, const T *dout, T *out, float s, unsigned thr16,
                                                           unsigned long long seed, unsigned long long offset) {
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n8; g += (long)gridDim.x * 256) {
        bool keep[8];
        if (thr16) keep8(seed, offset, (unsigned long long)g, thr16, keep);
        float a[8], o[8];
        ld8<T>(h + 8 * g, a);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // the reference rounds silu(h) to the activation dtype before the dropout scales it
                const float sv = Elem<T>::round(a[e] * __builtin_amdgcn_rcpf(1.f + __expf(-a[e])));
                o[e] = (!thr16 || keep[e]) ? s * sv : 0.f;
            }
        } else {
            float d[8];
            ld8<T>(dout + 8 * g, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-a[e]));
                const float ds = sg * (1.f + a[e] * (1.f - sg));                     // d silu / dh
                o[e] = (!thr16 || keep[e]) ? s * d[e] * ds : 0.f;
            }
        }
        st8<T>(out + 8 * g, o);
    }
}

inline unsigned grid_for(long n8) {
    const long blocks = (n8 + 255) / 256;
    const long cap = (long)device_cus() * 16;
    return (unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}
inline unsigned thr_of(float p) {   // keep iff u16 >= thr: P(drop) = thr / 65536
    const float t = p * 65536.f + 0.5f;
    return p <= 0.f ? 0u : (unsigned)(t > 65535.f ? 65535.f : t);
}

}  // namespace

// ---- multi-tensor transpose + cast: the training step's bf16 weight copies in (K, N) layout ---------------------------------
// One launch for ALL weights: dX = dY W is the forward GEMM against W^T, and W changes every step.  The table (device memory)
// holds one descriptor per tensor; a block of 256 threads moves one 64 x 64 tile through LDS (reads along the source rows,
// writes along the destination rows: both sides in whole 128 / 256-byte segments) and finds its tensor by a binary search over
// the tile offsets.
struct TrDesc {
    const void *src;      // (rows, cols) row-major, fp32 or bf16
    void *dst;            // (cols, rows) row-major bf16
    int rows, cols;
    int src_f32;          // 1: fp32 source, 0: bf16
    int tile0;            // first tile of this tensor in the grid
};

__global__ __launch_bounds__(256) void multi_transpose_kernel(const TrDesc *__restrict__ tab, int n) {
    __shared__ unsigned short tile[64][66];
    int lo = 0, hi = n - 1;
    const int t = blockIdx.x;
    while (lo < hi) {                       // last descriptor with tile0 <= t (uniform over the block)
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile0 <= t) lo = mid; else hi = mid - 1;
    }
    const TrDesc d = tab[lo];
    const int tc = (d.cols + 63) >> 6;
    const int lt = t - d.tile0, r0 = (lt / tc) * 64, c0 = (lt % tc) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        unsigned short v = 0;
        if (r < d.rows && c < d.cols) {
            const size_t o = (size_t)r * d.cols + c;
            v = d.src_f32 ? (unsigned short)f32_to_bf16_bits(reinterpret_cast<const float *>(d.src)[o])
                          : reinterpret_cast<const unsigned short *>(d.src)[o];
        }
        tile[ty * 16 + i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;          // destination row c, column r
        if (c < d.cols && r < d.rows) reinterpret_cast<unsigned short *>(d.dst)[(size_t)c * d.rows + r] = tile[tx][ty * 16 + i];
    }
}

// The same table, no transposition: dst (bf16) = src (fp32 or bf16) element for element -- the bf16 copies of the fp32 master weights
// that train_shadows() refreshes once per step.  (torch._foreach_copy_ from fp32 to bf16 lists is one launch per tensor on this
// build: 108 of the step's launches.)  A block moves one chunk of 4 096 elements; tile0 counts chunks; rows x cols = the element count.
constexpr int CAST_CHUNK = 4096;
__global__ __launch_bounds__(256) void multi_cast_kernel(const TrDesc *__restrict__ tab, int n) {
    int lo = 0, hi = n - 1;
    const int t = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile0 <= t) lo = mid; else hi = mid - 1;
    }
    const TrDesc d = tab[lo];
    const long numel = (long)d.rows * d.cols;
    const long base = (long)(t - d.tile0) * CAST_CHUNK;
#pragma unroll
    for (int i = 0; i < CAST_CHUNK / 256; ++i) {
        const long idx = base + i * 256 + threadIdx.x;
        if (idx < numel)
            reinterpret_cast<unsigned short *>(d.dst)[idx] =
                d.src_f32 ? (unsigned short)f32_to_bf16_bits(reinterpret_cast<const float *>(d.src)[idx])
                          : reinterpret_cast<const unsigned short *>(d.src)[idx];
    }
}

}  // namespace pafc

using pafc::bf16_t;

extern "C" int pafc_residual_dropout(int backward, int dtype_x, int dtype_y, long n, const void *x, const void *y, void *out,
                                     float scale, float p, unsigned long long seed, unsigned long long offset, pafc_stream_t stream) {
    if (!x || !out || (!backward && !y)) return PAFC_ERR_NULL_POINTER;
    if (n <= 0 || n % 8) return PAFC_ERR_BAD_DIMS;
    if (!(p >= 0.f && p < 1.f)) return PAFC_ERR_BAD_DIMS;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)out) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    const long n8 = n / 8;
    const unsigned thr = pafc::thr_of(p);
    const float s = scale / (1.f - (float)thr / 65536.f);       // the rate the 16-bit threshold realises
    const dim3 grid(pafc::grid_for(n8)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define PAFC_RD(TX, TY)                                                                                                         \
    do {                                                                                                                         \
        if (backward) hipLaunchKernelGGL((pafc::residual_dropout_kernel<TX, TY, 1>), grid, block, 0, st, n8, (const TX *)x,      \
                                         (const TY *)nullptr, out, s, thr, seed, offset);                                        \
        else hipLaunchKernelGGL((pafc::residual_dropout_kernel<TX, TY, 0>), grid, block, 0, st, n8, (const TX *)x, (const TY *)y, \
                                out, s, thr, seed, offset);                                                                      \
    } while (0)
    if (dtype_x == PAFC_F32 && dtype_y == PAFC_BF16) PAFC_RD(float, bf16_t);
    else if (dtype_x == PAFC_F32 && dtype_y == PAFC_F32) PAFC_RD(float, float);
    else if (dtype_x == PAFC_BF16 && dtype_y == PAFC_BF16) PAFC_RD(bf16_t, bf16_t);
    else return PAFC_ERR_UNSUPPORTED;
#undef PAFC_RD
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_silu_dropout(int backward, int dtype, long n, const void *h, const void *dout, void *out, float p,
                                 unsigned long long seed, unsigned long long offset, pafc_stream_t stream) {
    if (!h || !out || (backward && !dout)) return PAFC_ERR_NULL_POINTER;
    if (n <= 0 || n % 8) return PAFC_ERR_BAD_DIMS;
    if (!(p >= 0.f && p < 1.f)) return PAFC_ERR_BAD_DIMS;
    if ((((uintptr_t)h | (uintptr_t)dout | (uintptr_t)out) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    const long n8 = n / 8;
    const unsigned thr = pafc::thr_of(p);
    const float s = 1.f / (1.f - (float)thr / 65536.f);
    const dim3 grid(pafc::grid_for(n8)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PAFC_BF16) {
        if (backward) hipLaunchKernelGGL((pafc::silu_dropout_kernel<bf16_t, 1>), grid, block, 0, st, n8, (const bf16_t *)h, (const bf16_t *)dout, (bf16_t *)out, s, thr, seed, offset);
        else hipLaunchKernelGGL((pafc::silu_dropout_kernel<bf16_t, 0>), grid, block, 0, st, n8, (const bf16_t *)h, (const bf16_t *)nullptr, (bf16_t *)out, s, thr, seed, offset);
    } else if (dtype == PAFC_F32) {
        if (backward) hipLaunchKernelGGL((pafc::silu_dropout_kernel<float, 1>), grid, block, 0, st, n8, (const float *)h, (const float *)dout, (float *)out, s, thr, seed, offset);
        else hipLaunchKernelGGL((pafc::silu_dropout_kernel<float, 0>), grid, block, 0, st, n8, (const float *)h, (const float *)nullptr, (float *)out, s, thr, seed, offset);
    } else {
        return PAFC_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

// bf16 transposed copies of many matrices in ONE launch (include/pafc_encoder_ops.h).  `table`: n descriptors of 32 bytes in
// DEVICE memory -- { const void *src; void *dst; int rows, cols, src_f32, tile0; } with tile0 = the running sum of
// ceil(rows / 64) * ceil(cols / 64) over the earlier tensors; total_tiles = that sum over all n.
extern "C" int pafc_multi_cast_bf16(const void *table, int n, int total_chunks, pafc_stream_t stream) {
    if (!table) return PAFC_ERR_NULL_POINTER;
    if (n <= 0 || total_chunks <= 0) return PAFC_ERR_BAD_DIMS;
    hipLaunchKernelGGL(pafc::multi_cast_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream,
                       (const pafc::TrDesc *)table, n);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_multi_transpose_bf16(const void *table, int n, int total_tiles, pafc_stream_t stream) {
    if (!table) return PAFC_ERR_NULL_POINTER;
    if (n <= 0 || total_tiles <= 0) return PAFC_ERR_BAD_DIMS;
    if (((uintptr_t)table & 7) != 0) return PAFC_ERR_ALIGNMENT;
    static_assert(sizeof(pafc::TrDesc) == 32, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(pafc::multi_transpose_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const pafc::TrDesc *)table, n);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
