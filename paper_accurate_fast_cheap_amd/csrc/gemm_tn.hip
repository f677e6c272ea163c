// Weight gradient of the dense projections (training step, config c4) as a hand-written bf16 GEMM for gfx950.
// C ABI: include/pafc_encoder_ops.h: pafc_gemm_tn_bf16.
//
//   dW[m][n] = sum_r dY[r][m] * X[r][n]            dY: (R, M), X: (R, N), both row-major (the contraction index r =
//                                                  batch x time is the SLOW axis of both operands)
// i.e. what autograd computes for nn.Linear's weight (grad_output^T @ input), R = 16 000 rows against a 512 x 2048
// output at the c4 shape.  The library's pick for this layout is a 64 x 64 tile walking all of R serially
// (95 us = 130 TFLOP/s for every such product, 18 % of the training step); here
//   * the R axis is split over S blocks per output tile (grid.y) so that ~2 blocks per CU are in flight; each leaves an
//     fp32 partial tile, a second kernel adds the S partials in a fixed order (deterministic: no atomics) and writes
//     dW directly in fp32 (or bf16) -- the bf16 rounding of the library path is skipped for fp32 master weights;
//   * both operand tiles ([64 r][128 columns], 256-byte rows) go global -> LDS by LDS-DMA exactly as they lie in
//     memory, and the MFMA operands (8 consecutive r per lane) are gathered by the transposing LDS read
//     ds_read_b64_tr_b16 (2 per 16 x 32 operand); the 16-byte chunks of a row are XOR-swizzled on the SOURCE side of
//     the DMA with the pattern that keeps those reads conflict-free:
//         chunk' = chunk ^ (((r & 3) << 2) | ((r >> 2) & 3))
//   * rows past the end of a block's R range are fetched from the last valid row (in bounds) and the dY tile's rows
//     are zeroed in LDS, so ragged R costs one extra pass over <= 63 LDS rows in one block per tile.
// Block = 256 threads (2 x 2 waves), tile 128 x 128 x 64, wave tile 64 x 64 = 4 x 4 MFMA 16x16x32 bf16, 2 LDS stages.
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int TBM = 128, TBN = 128, TBK = 64;
typedef float f32x4t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));
typedef short s16x4t __attribute__((ext_vector_type(4)));
typedef short s16x8t __attribute__((ext_vector_type(8)));

struct TnParams {
    const bf16_t *A, *B;     // dY (R, M) with row stride lda; X (R, N) with row stride ldb
    float *part;             // (S, M, N) fp32
    float *bias_part;        // (S, M) fp32 column sums of dY (the bias gradient), or null
    long R, lda, ldb, rows_per_split;
    int M, N, mtiles, ntiles, S;
};

__device__ __forceinline__ void tdma16(const bf16_t *src, bf16_t *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

__device__ __forceinline__ s16x4t tr_read(const bf16_t *lds, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4t *)((const char *)lds + byte_off));
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];   // [2 stages][A 64x128 | B 64x128] (64 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int mt = blockIdx.x / p.ntiles, nt = blockIdx.x % p.ntiles;
    const int m0 = mt * TBM, n0 = nt * TBN;
    const long k_begin = (long)blockIdx.y * p.rows_per_split;
    const long k_end = min(p.R, k_begin + p.rows_per_split);
    const int iters = k_end > k_begin ? (int)((k_end - k_begin + TBK - 1) / TBK) : 0;

    // DMA: instruction j of wave w brings rows 4 (4 j + w) .. + 3 (1 KiB); lane -> (row, swizzled chunk)
    int d_row[4], a_col[4], b_col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 4 + wave) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        d_row[j] = row;
        a_col[j] = min(m0 + ch * 8, p.M - 8);     // column tail: clamped chunks feed outputs that are never stored
        b_col[j] = min(n0 + ch * 8, p.N - 8);
    }
    auto issue = [&](int it, int buf) {
        bf16_t *At = lds + buf * (2 * TBK * TBM);
        bf16_t *Bt = At + TBK * TBM;
        const long kb = k_begin + (long)it * TBK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long r = min(kb + d_row[j], k_end - 1);
            const int base = (j * 4 + wave) * 4 * TBM;
            tdma16(p.A + r * p.lda + a_col[j], At + base);
            tdma16(p.B + r * p.ldb + b_col[j], Bt + base);
        }
    };

    f32x4t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4t{0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of dY: one more MFMA per dY operand against a block of ones, in the blocks of the
    // first N-tile only (wave column 0), instead of a separate reduction pass over dY
    const bool want_bias = p.bias_part != nullptr && nt == 0 && wn == 0;   // wave-uniform
    f32x4t accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4t{0.f, 0.f, 0.f, 0.f};
    const s16x8t ones_bits = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    const bf16x8t ones = __builtin_bit_cast(bf16x8t, ones_bits);

    // transposing reads: lane 4 q + pp of 16-lane group g supplies the address of row (8 g + 4 h + q), columns
    // 16 blk + 4 pp .. + 3 and receives column (lane & 15) of the four rows
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    int rd_base[2], rd_swz[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + q;
        rd_base[h] = 256 * row + 8 * (pp & 1);
        rd_swz[h] = ((row & 3) << 2) | ((row >> 2) & 3);
    }

    if (iters > 0) issue(0, 0);
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 1 < iters) issue(it + 1, (it + 1) & 1);
        bf16_t *At = lds + (it & 1) * (2 * TBK * TBM);
        const bf16_t *Bt = At + TBK * TBM;
        const long valid = k_end - (k_begin + (long)it * TBK);
        if (valid < TBK) {   // block-uniform: the ragged end of this block's R range
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            for (int c = tid; c < (TBK - (int)valid) * 16; c += 256)
                *reinterpret_cast<uint4 *>((char *)At + valid * 256 + (long)c * 16) = z;
            __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ca = wm * 8 + 2 * i, cb = wn * 8 + 2 * i;
                s16x4t a0 = tr_read(At, ks * 8192 + rd_base[0] + 16 * ((ca | (pp >> 1)) ^ rd_swz[0]));
                s16x4t a1 = tr_read(At, ks * 8192 + rd_base[1] + 16 * ((ca | (pp >> 1)) ^ rd_swz[1]));
                s16x4t b0 = tr_read(Bt, ks * 8192 + rd_base[0] + 16 * ((cb | (pp >> 1)) ^ rd_swz[0]));
                s16x4t b1 = tr_read(Bt, ks * 8192 + rd_base[1] + 16 * ((cb | (pp >> 1)) ^ rd_swz[1]));
                af[i] = __builtin_bit_cast(bf16x8t, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                bfr[i] = __builtin_bit_cast(bf16x8t, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (want_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
            }
        }
    }
    if (want_bias && (lane & 15) == 0) {   // all 16 columns of accb hold the same sums
        float *bo = p.bias_part + (size_t)blockIdx.y * p.M;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + 4 * g + r;
                if (m < p.M) bo[m] = accb[i][r];
            }
    }

    // C/D layout: column = lane & 15 (n), row = 4 (lane >> 4) + reg (m); 16 lanes write 64 contiguous bytes
    float *out = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 64 + i * 16 + 4 * g + r;
            if (m < p.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + (lane & 15);
                    if (n < p.N) out[(size_t)m * p.N + n] = acc[i][j][r];
                }
            }
        }
}

// dW = sum of the S partials, in order; out_f32 or out_bf16 (exactly one is non-null).  The bias gradient's partials ride in
// the same launch: indices [n4, n4 + nb4) reduce bias_part (stride nb4) into the bias outputs.
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(long n4, int S, long stride4, const float4 *__restrict__ part,
                                                             float4 *__restrict__ out_f32, uint2 *__restrict__ out_bf16,
                                                             long nb4, const float4 *__restrict__ bias_part,
                                                             float4 *__restrict__ bias_f32, uint2 *__restrict__ bias_bf16) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4 + nb4) return;
    if (i >= n4) {                                   // (block-uniform except in one block)
        i -= n4; part = bias_part; stride4 = nb4; out_f32 = bias_f32; out_bf16 = bias_bf16;
    }
    float4 a = part[i];
    for (int s = 1; s < S; ++s) {
        const float4 b = part[(long)s * stride4 + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (out_f32) out_f32[i] = a;
    else
        out_bf16[i] = make_uint2(f32_to_bf16_bits(a.x) | (f32_to_bf16_bits(a.y) << 16),
                                 f32_to_bf16_bits(a.z) | (f32_to_bf16_bits(a.w) << 16));
}

void plan(long R, int M, int N, int *S, long *rows_per_split) {
    const long tiles = (long)((M + TBM - 1) / TBM) * ((N + TBN - 1) / TBN);
    long s = (512 + tiles - 1) / tiles;                 // ~2 blocks per CU
    const long max_s = (R + 2 * TBK - 1) / (2 * TBK);    // at least two K steps per block
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    long rps = ((R + s - 1) / s + TBK - 1) / TBK * TBK;
    *rows_per_split = rps;
    *S = (int)((R + rps - 1) / rps);
}

}  // namespace
}  // namespace pafc

extern "C" size_t pafc_gemm_tn_workspace_bytes(long R, int M, int N) {
    if (R <= 0 || M <= 0 || N <= 0) return 0;
    int S; long rps;
    pafc::plan(R, M, N, &S, &rps);
    return (size_t)S * M * (N + 1) * sizeof(float);      // partial tiles + partial column sums
}

extern "C" int pafc_gemm_tn_bf16(long R, int M, int N, const void *dy, long lda, const void *x, long ldb, void *dw,
                                 void *dbias, int dw_dtype, void *workspace, size_t workspace_bytes,
                                 pafc_stream_t stream) {
    if (!dy || !x || !dw || !workspace) return PAFC_ERR_NULL_POINTER;
    if (R <= 0 || M < 8 || N < 8 || (M % 8) || (N % 8) || lda < M || ldb < N || (lda % 8) || (ldb % 8))
        return PAFC_ERR_BAD_DIMS;
    if (dw_dtype != PAFC_F32 && dw_dtype != PAFC_BF16) return PAFC_ERR_DTYPE;
    if ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw | (uintptr_t)dbias | (uintptr_t)workspace) & 15) != 0)
        return PAFC_ERR_ALIGNMENT;
    pafc::TnParams p{};
    pafc::plan(R, M, N, &p.S, &p.rows_per_split);
    if (workspace_bytes < (size_t)p.S * M * (N + 1) * sizeof(float)) return PAFC_ERR_WORKSPACE;
    p.A = (const pafc::bf16_t *)dy; p.B = (const pafc::bf16_t *)x; p.part = (float *)workspace;
    p.bias_part = dbias ? p.part + (size_t)p.S * M * N : nullptr;
    p.R = R; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N;
    p.mtiles = (M + pafc::TBM - 1) / pafc::TBM; p.ntiles = (N + pafc::TBN - 1) / pafc::TBN;
    if ((long)p.mtiles * p.ntiles > 2147483647L || p.S > 65535) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 2 * 2 * pafc::TBK * pafc::TBM * sizeof(pafc::bf16_t);
    if (hipFuncSetAttribute((const void *)pafc::gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return PAFC_ERR_LAUNCH;
    hipLaunchKernelGGL(pafc::gemm_tn_kernel, dim3(p.mtiles * p.ntiles, p.S), dim3(256), lds, s, p);
    const long n4 = (long)M * N / 4, nb4 = dbias ? M / 4 : 0;
    hipLaunchKernelGGL(pafc::gemm_tn_reduce_kernel, dim3((unsigned)((n4 + nb4 + 255) / 256)), dim3(256), 0, s, n4, p.S, n4,
                       (const float4 *)workspace, dw_dtype == PAFC_F32 ? (float4 *)dw : nullptr,
                       dw_dtype == PAFC_BF16 ? (uint2 *)dw : nullptr, nb4, (const float4 *)p.bias_part,
                       dw_dtype == PAFC_F32 ? (float4 *)dbias : nullptr, dw_dtype == PAFC_BF16 ? (uint2 *)dbias : nullptr);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
