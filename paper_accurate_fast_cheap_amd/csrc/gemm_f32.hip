// fp32 GEMM with a fused epilogue on the fp32 matrix cores of gfx950 (C ABI: include/pafc_encoder_ops.h: pafc_gemm_f32).
//
//   out (M, N) = act(alpha * A (M, K) . W (N, K)^T + bias (N) + residual (M, N)),   batched over grid.z
//
// Exact fp32 products, fp32 accumulation (v_mfma_f32_32x32x2_f32): the arithmetic of the reference's fp32 nn.Linear /
// 1 x 1 Conv1d / torch.bmm call sites (wenet/transformer/positionwise_feed_forward.py:47-55, convolution.py:118-141,
// rwkv_v6/src/model.py:277-324, subsampling.py:218-224, ctc.py:106-114, encoder_layer.py:201-259) for a model WITHOUT the
// bf16 slot (rwkv_do_bfloat16: False -- BASELINE configs[0], the 1e-3 parity configuration) and for the few-rows fp32
// products of the other precision modes.  It replaces the library GEMM (hipBLASLt plans / the framework's F.linear) on every
// inference path: those kernels stall when two HIP streams issue them concurrently ("the c2 stall":
// tools/repro_plan_churn.py, tools/micro/two_stream_linear.py), and decode batches run two or three streams deep.
//
// Block = 256 threads = 2 x 2 waves, block tile (64 TM) x (64 TN), wave tile (32 TM) x (32 TN), K-step 32 / 64, two LDS stages:
//   * global -> registers -> LDS: a thread fetches one 16-byte quad of TM + TN ... rows per K-step (8 consecutive lanes cover
//     128 contiguous bytes of a row), the NEXT step's quads are requested before the current step's MFMAs and stored behind
//     them; rows beyond M / N are clamped (their results are never stored), quads beyond K are zero.
//   * LDS rows are 36 dwords apart: ds_write_b128 of 8 lanes = one row's 128 bytes; ds_read_b128 of a 16-lane group hits 16
//     distinct 4-bank runs of the 64 banks (36 m mod 64, m in the group's rows).
//   * K order inside a step is permuted identically for both operands so that a lane reads CONTIGUOUS k: lane half h of a wave
//     owns k = (KS / 2) h .. (KS / 2) (h + 1) - 1 of the step, and MFMA j of the step multiplies k pair {j, KS / 2 + j} -- two
//     ds_read_b128 per operand tile feed eight MFMAs (a dot product does not care in which order its terms are added).
//   * accumulator layout of the 32 x 32 MFMA: lane l, register r holds row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32 -- a
//     store instruction writes 128 contiguous bytes of two output rows.
#include <cstdlib>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // (HIP's float4 is a struct: selects and array elements of it go through memory)

// K-step KS (32 for the 128 x 128 tiles, 64 for the 64 x 64 ones: a block of few-rows problems has the CU to itself, and every
// K-step exposes a global-load round trip of ~1 us behind only 16 MFMAs per wave -- half as many, twice as long steps);
// LDS row stride KS + 4 dwords: 36 m or 68 m mod 64 hits 16 distinct 4-bank runs over the 16 rows of a ds_read_b128 group

struct GemmF32Params {
    const float *A, *W, *bias, *res;
    float *out;
    long M, lda, ldw, ldr, ldo;
    long sA, sW, sB, sR, sO;    // batch strides in elements
    int N, K, act, tiles_n;
    float alpha;
};

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case 1: return v / (1.f + __expf(-v));                     // SiLU
        case 2: return tanhf(v);
        case 3: return v > 0.f ? v : 0.f;
        default: return v;
    }
}

template <int TM, int TN, int KS>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int LDT = KS + 4;                                    // LDS row stride in dwords
    constexpr int QR = KS / 4;                                     // quads per row and K-step: 8 or 16 consecutive lanes cover a row's KS floats
    constexpr int RS = 256 / QR;                                   // rows one pass of the block's 256 threads covers
    constexpr int RA = BM / RS, RB = BN / RS;                      // quads per thread and K-step (rows RS apart)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                                               // [2][BM][LDT]
    float *Bs = lds + 2 * BM * LDT;                                // [2][BN][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int tn = blockIdx.x % p.tiles_n;
    const long tm = blockIdx.x / p.tiles_n;
    const int z = blockIdx.z;
    const float *A = p.A + z * p.sA, *W = p.W + z * p.sW;
    const long m0 = tm * BM;
    const int n0 = tn * BN;
    const int q = tid % QR, r0 = tid / QR;                         // this thread's quad of the K-step and first row
    const float *ga[RA], *gb[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        long r = m0 + r0 + RS * i;
        ga[i] = A + (r < p.M ? r : p.M - 1) * p.lda + 4 * q;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        int r = n0 + r0 + RS * i;
        gb[i] = W + (long)(r < p.N ? r : p.N - 1) * p.ldw + 4 * q;
    }
    const int ksteps = (p.K + KS - 1) / KS;
    f32x4 pa[RA], pb[RB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // (macros, not lambdas: arrays captured by reference end up in scratch memory)
#define PAFC_F32_FETCH(ks_)                                                                                           \
    {                                                                                                                 \
        const bool in_ = (ks_) * KS + 4 * q < p.K; /* K % 4 == 0: a quad is inside or outside as a whole */            \
        _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                                \
            pa[i] = in_ ? *reinterpret_cast<const f32x4 *>(ga[i] + (long)(ks_) * KS) : zero4;                        \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                                \
            pb[i] = in_ ? *reinterpret_cast<const f32x4 *>(gb[i] + (long)(ks_) * KS) : zero4;                        \
    }
#define PAFC_F32_STASH(buf_)                                                                                          \
    {                                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                                \
            *reinterpret_cast<f32x4 *>(As + ((buf_) * BM + r0 + RS * i) * LDT + 4 * q) = pa[i];                      \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                                \
            *reinterpret_cast<f32x4 *>(Bs + ((buf_) * BN + r0 + RS * i) * LDT + 4 * q) = pb[i];                      \
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    PAFC_F32_FETCH(0)
    PAFC_F32_STASH(0)
    __syncthreads();
    const int lrow = lane & 31, half = lane >> 5;
    for (int ks = 0; ks < ksteps; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < ksteps) PAFC_F32_FETCH(ks + 1)                // in flight under this step's MFMAs
        const float *ab = As + (buf * BM + wm * 32 * TM + lrow) * LDT + (KS / 2) * half;
        const float *bb = Bs + (buf * BN + wn * 32 * TN + lrow) * LDT + (KS / 2) * half;
#pragma unroll
        for (int hh = 0; hh < KS / 16; ++hh) {
            f32x4 a[TM][2], b[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a[i][0] = *reinterpret_cast<const f32x4 *>(ab + i * 32 * LDT + 8 * hh);
                a[i][1] = *reinterpret_cast<const f32x4 *>(ab + i * 32 * LDT + 8 * hh + 4);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                b[j][0] = *reinterpret_cast<const f32x4 *>(bb + j * 32 * LDT + 8 * hh);
                b[j][1] = *reinterpret_cast<const f32x4 *>(bb + j * 32 * LDT + 8 * hh + 4);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = a[i][e >> 2][e & 3];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = b[j][e >> 2][e & 3];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (ks + 1 < ksteps) {
            PAFC_F32_STASH(buf ^ 1)                                // the other stage: last read one barrier ago
            __syncthreads();
        }
    }
    // ---- epilogue: alpha, bias, residual, activation on the accumulator; 128-byte runs per store --------------------
    const float *bias = p.bias ? p.bias + z * p.sB : nullptr;
    const float *res = p.res ? p.res + z * p.sR : nullptr;
    float *out = p.out + z * p.sO;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + j * 32 + lrow;
        if (col >= p.N) continue;
        const float bj = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + wm * 32 * TM + i * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                if (row >= p.M) continue;
                float v = p.alpha * acc[i][j][r] + bj;
                if (res) v += res[row * p.ldr + col];
                out[row * p.ldo + col] = act_apply(v, p.act);
            }
        }
    }
}

// ---- few rows: no operand staging at all ------------------------------------------------------------------------------------
// A decode batch of short windows is a few hundred rows: 128 x 128 or 64 x 64 tiles leave most CUs idle and every K-step of the
// staged kernel exposes a barrier and a global-load round trip (16 of them at K = 512, 64 at K = 2048).  Here a block owns a
// 32 x (32 TN) output tile and its four waves SPLIT K: wave w takes the 64-wide K-steps w, w + 4, ... -- two steps per wave at
// K = 512 -- straight from global memory (L2: the operands of a few-rows product are small) into MFMA operand registers: lane
// (m = lane % 32, h = lane / 32) reads the 32 contiguous k of its half of the step from row m (eight 16-byte quads per operand
// tile), which is exactly the operand the 32 x 32 x 2 MFMA wants from it under the K permutation of the staged kernel (MFMA j of
// the step multiplies the k pair {j, 32 + j}).  The next step's quads are requested before the current step's MFMAs.  The four
// partial accumulators meet in LDS once, at the end; wave w then finishes rows 8 w + 4 h + (0 .. 3) of every 8-row group.
template <int TN>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const GemmF32Params p) {
    constexpr int KS2 = 64;
    __shared__ float red[4][TN][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lrow = lane & 31, half = lane >> 5;
    const int tn = blockIdx.x % p.tiles_n;
    const long tm = blockIdx.x / p.tiles_n;
    const int z = blockIdx.z;
    const long m0 = tm * 32;
    const int n0 = tn * 32 * TN;
    const long ra = m0 + lrow;
    const float *ga = p.A + z * p.sA + (ra < p.M ? ra : p.M - 1) * p.lda + 32 * half;
    const float *gb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rb = n0 + j * 32 + lrow;
        gb[j] = p.W + z * p.sW + (long)(rb < p.N ? rb : p.N - 1) * p.ldw + 32 * half;
    }
    const int ksteps = (p.K + KS2 - 1) / KS2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][8], b[2][TN][8];
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#define PAFC_F32_SMALL_FETCH(buf_, ks_)                                                                               \
    {                                                                                                                 \
        const int k0_ = (ks_) * KS2 + 32 * half;                                                                      \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                               \
            const bool in_ = k0_ + 4 * e < p.K;                                                                       \
            a[buf_][e] = in_ ? *reinterpret_cast<const f32x4 *>(ga + (long)(ks_) * KS2 + 4 * e) : zero4;               \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                            \
                b[buf_][j][e] = in_ ? *reinterpret_cast<const f32x4 *>(gb[j] + (long)(ks_) * KS2 + 4 * e) : zero4;     \
        }                                                                                                             \
    }
#define PAFC_F32_SMALL_MMA(buf_)                                                                                      \
    {                                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                                 \
            _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                             \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                        \
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf_][e][c], b[buf_][j][e][c], acc[j], 0, 0, 0);  \
    }
    int ks = wv;
    if (ks < ksteps) PAFC_F32_SMALL_FETCH(0, ks)
    for (; ks < ksteps; ks += 8) {                  // two of this wave's steps per trip: register sets 0 and 1 by name
        if (ks + 4 < ksteps) PAFC_F32_SMALL_FETCH(1, ks + 4)
        PAFC_F32_SMALL_MMA(0)
        if (ks + 4 < ksteps) {
            if (ks + 8 < ksteps) PAFC_F32_SMALL_FETCH(0, ks + 8)
            PAFC_F32_SMALL_MMA(1)
        }
    }
#undef PAFC_F32_SMALL_FETCH
#undef PAFC_F32_SMALL_MMA
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv][j][r][lane] = acc[j][r];
    __syncthreads();
    const float *bias = p.bias ? p.bias + z * p.sB : nullptr;
    const float *res = p.res ? p.res + z * p.sR : nullptr;
    float *out = p.out + z * p.sO;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + j * 32 + lrow;
        if (col >= p.N) continue;
        const float bj = bias ? bias[col] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {               // register r = 4 g + wv of every wave's partial: row 8 g + 4 half + wv
            const int r = 4 * g + wv;
            const long row = m0 + 8 * g + 4 * half + wv;
            if (row >= p.M) continue;
            const float sum = (red[0][j][r][lane] + red[1][j][r][lane]) + (red[2][j][r][lane] + red[3][j][r][lane]);
            float v = p.alpha * sum + bj;
            if (res) v += res[row * p.ldr + col];
            out[row * p.ldo + col] = act_apply(v, p.act);
        }
    }
}

template <int TN>
int launch_small(const GemmF32Params &p, int batch, hipStream_t s) {
    GemmF32Params q = p;
    q.tiles_n = (p.N + 32 * TN - 1) / (32 * TN);
    const long blocks = ((p.M + 31) / 32) * q.tiles_n;
    if (blocks > 0x7fffffffL || batch > 65535) return PAFC_ERR_BAD_DIMS;
    hipLaunchKernelGGL(gemm_f32_small_kernel<TN>, dim3((unsigned)blocks, 1, (unsigned)batch), dim3(256), 0, s, q);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

#undef PAFC_F32_FETCH
#undef PAFC_F32_STASH

template <int TM, int TN, int KS>
int launch(const GemmF32Params &p, int batch, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN, LDT = KS + 4;
    GemmF32Params q = p;
    q.tiles_n = (p.N + BN - 1) / BN;
    const long tiles_m = (p.M + BM - 1) / BM;
    const long blocks = tiles_m * q.tiles_n;
    if (blocks > 0x7fffffffL || batch > 65535) return PAFC_ERR_BAD_DIMS;
    const size_t lds = (size_t)2 * (BM + BN) * LDT * sizeof(float);
    auto kern = gemm_f32_kernel<TM, TN, KS>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PAFC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, 1, (unsigned)batch), dim3(256), lds, s, q);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_gemm_f32(long M, int N, int K, int batch, const float *A, long lda, long strideA, const float *W, long ldw,
                             long strideW, const float *bias, long strideBias, const float *residual, long ldr, long strideR,
                             float *out, long ldo, long strideO, float alpha, int act, pafc_stream_t stream) {
    if (!A || !W || !out) return PAFC_ERR_NULL_POINTER;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return PAFC_ERR_BAD_DIMS;
    if (act < 0 || act > 3) return PAFC_ERR_UNSUPPORTED;
    // 16-byte quads along K: K, the leading dimensions and batch strides of A and W multiples of 4 floats, bases 16-byte aligned
    if (K % 4 || lda % 4 || ldw % 4 || strideA % 4 || strideW % 4 || lda < K || ldw < K) return PAFC_ERR_UNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)W) & 15) return PAFC_ERR_ALIGNMENT;
    if (ldo < N || (residual && ldr < N)) return PAFC_ERR_BAD_DIMS;
    pafc::GemmF32Params p;
    p.A = A; p.W = W; p.bias = bias; p.res = residual; p.out = out;
    p.M = M; p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldo = ldo;
    p.sA = strideA; p.sW = strideW; p.sB = strideBias; p.sR = strideR; p.sO = strideO;
    p.N = N; p.K = K; p.act = act; p.tiles_n = 0; p.alpha = alpha;
    hipStream_t s = (hipStream_t)stream;
    // 128 x 128 tiles once they give every CU a block or more (two fit a CU); smaller problems take 64 x 64 tiles, four times the
    // blocks (a decode batch of a few hundred rows x N = 512 is 16 big tiles on 256 CUs)
    const long big = ((M + 127) / 128) * ((N + 127) / 128) * batch;
    const int cus = pafc::device_cus();
    static const int force = getenv("PAFC_GEMM_F32_KERNEL") ? atoi(getenv("PAFC_GEMM_F32_KERNEL")) : 0;   // A/B: 1 big, 2 mid, 3 few-rows
    if (force == 1 || (!force && big >= cus)) return pafc::launch<2, 2, 32>(p, batch, s);
    // so few 64 x 64 tiles that three quarters of the CUs would idle (a single short window: 499 rows x N = 512 is 64 tiles): the
    // few-rows kernel -- 32 x 32 tiles, K split over the four waves of a block, operands straight from L2 to registers.  Its
    // operand traffic grows with the tile count (each tile re-reads its 64 rows of K floats), so wider / taller problems keep the
    // staged tiles (tools/bench_gemm_f32.py, profiles/r06h_gemm_f32_variants.txt)
    const long mid = ((M + 63) / 64) * ((N + 63) / 64) * batch;
    if (force == 3 || (!force && 4 * mid <= cus)) return pafc::launch_small<1>(p, batch, s);
    return pafc::launch<1, 1, 64>(p, batch, s);
}
