// 3x3 stride-2 convolution of the subsampling front end as an implicit GEMM on the bf16 matrix cores (gfx950).
// C ABI: include/pafc_encoder_ops.h: pafc_conv3x3s2_nhwc_bf16.
//
// Replaces the second Conv2d(C, C, 3, 2) + ReLU of Conv2dSubsampling4 (wenet/transformer/subsampling.py:187-192):
// 31 % of all encoder MACs and the largest single kernel of the pass.  Through the framework it is
// a library convolution whose algorithm is picked per input shape by a search that costs seconds per new shape
// (and, without the search, falls back to a kernel ~4x slower): a ragged decode set of ~90 batch shapes
// spends most of its time there.  Here it is one kernel with shape-independent behaviour.
//
//   out[b][t2][f2][co] = relu(bias[co] + sum_{kh,kw,ci} w[co][ci][kh][kw] * in[b][2 t2 + kh][2 f2 + kw][ci])
// as a GEMM  M = B*T2*F2 output positions,  N = Co,  K = 9 * Ci, NHWC in and out (so every A row segment of one tap is
// 64 contiguous channels = 128 B, and the output needs no transpose before the Linear that follows).
//
// Block = 256 threads (4 waves, 2 x 2), tile 128 (M) x 128 (N) x 64 (K); each wave owns 64 x 64 = 4 x 4 MFMA
// 16x16x32 tiles (64 accumulator VGPRs).  A and B tiles go global -> LDS by LDS-DMA (global_load_lds, 16 B per lane,
// no VGPRs): the LDS image is lane-linear, so the XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free is applied to the per-lane SOURCE address (chunk p of row r holds source chunk p ^ (r & 7)).  Two LDS
// buffers: the loads of K-step i+1 are in flight while step i is multiplied.  Epilogue: bias + ReLU on the fp32
// accumulators, one rounding to bf16, staged through LDS and stored as whole 256-byte rows.
// Blocks are numbered so that the N-tiles of one M-tile run on one XCD (they share the A tile in that XCD's L2).
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int BM = 128, BN = 128, BK = 64;
typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));

struct ConvParams {
    const bf16_t *in;    // (B, T1, F1, Ci)
    const bf16_t *wt;    // (9, Co, Ci): tap-major, K contiguous
    const bf16_t *bias;  // (Co) or null
    bf16_t *out;         // (B, T2, F2, Co)
    int B, T1, F1, Ci, Co, T2, F2;
    long M;              // B * T2 * F2
    int relu;
    int mtiles, ntiles;
};

__device__ __forceinline__ void dma16(const bf16_t *src, bf16_t *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void conv3x3s2_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];   // [2 buffers][A 128x64 | B 128x64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: consecutive ids on one XCD walk (m-tile major, n-tile minor)
    const long nblk = (long)p.mtiles * p.ntiles;
    long bid = blockIdx.x;
    const long per = nblk / 8;
    if (bid < per * 8) bid = (bid % 8) * per + bid / 8;   // bijective on the first 8*per ids; the tail keeps its id
    const int mt0 = (int)(bid / p.ntiles), nt0 = (int)(bid % p.ntiles);
    const long m0 = (long)mt0 * BM;
    const int n0 = nt0 * BN;

    // ---- per-lane DMA sources: wave w fills rows [32 w, 32 w + 32) of A and of B, 4 instructions of 8 rows each ----
    const int sub = lane >> 3, pch = lane & 7;             // row within the 8-row group, LDS chunk position
    const bf16_t *a_src[4];
    const bf16_t *b_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + sub;
        const int c = pch ^ (row & 7);                      // source chunk that lands at position pch
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;                          // clamp: the row is computed but never stored
        const int f2 = (int)(m % p.F2);
        const long bt = m / p.F2;
        const int t2 = (int)(bt % p.T2);
        const int b = (int)(bt / p.T2);
        a_src[j] = p.in + (((long)b * p.T1 + 2 * t2) * p.F1 + 2 * f2) * p.Ci + 8 * c;
        b_src[j] = p.wt + (long)(n0 + row) * p.Ci + 8 * c;
    }
    const int kblocks = p.Ci / BK;
    const int iters = 9 * kblocks;
    const long tap_stride_b = (long)p.Co * p.Ci;

    auto issue = [&](int it, int buf) {
        const int tap = it / kblocks, kb = it - tap * kblocks;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const long aoff = ((long)kh * p.F1 + kw) * p.Ci + kb * BK;
        const long boff = tap * tap_stride_b + kb * BK;
        bf16_t *A = lds + buf * (2 * BM * BK);
        bf16_t *Bt = A + BM * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rowbase = (wave * 32 + j * 8) * BK;   // wave-uniform LDS base of this 8-row group (1 KiB)
            dma16(a_src[j] + aoff, A + rowbase);
            dma16(b_src[j] + boff, Bt + rowbase);
        }
    };

    f32x4c acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4c{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, kq = lane >> 4;
    issue(0, 0);
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // buffer it&1 landed; everyone left buffer (it+1)&1
        if (it + 1 < iters) issue(it + 1, (it + 1) & 1);
        const bf16_t *A = lds + (it & 1) * (2 * BM * BK);
        const bf16_t *Bt = A + BM * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8c af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                const int pos = (ks * 4 + kq) ^ (row & 7);
                af[i] = *reinterpret_cast<const bf16x8c *>(A + row * BK + pos * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + fr;
                const int pos = (ks * 4 + kq) ^ (row & 7);
                bfr[j] = *reinterpret_cast<const bf16x8c *>(Bt + row * BK + pos * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();   // all waves done with the operand buffers: reuse them to stage the output tile

    // ---- epilogue: C/D layout col = lane & 15 (n), row = 4 (lane >> 4) + reg (m) --------------------------------
    constexpr int LDO = BN + 8;
    bf16_t *O = lds;   // [128][136] bf16 = 34 KiB <= 64 KiB
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = wn * 64 + j * 16 + fr;
        const float bv = p.bias ? bf16_bits_to_f32(p.bias[n0 + col]) : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = acc[i][j][g] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                O[(wm * 64 + i * 16 + 4 * kq + g) * LDO + col] = (bf16_t)f32_to_bf16_bits(v);
            }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = q * 16 + (tid >> 4), c8 = (tid & 15) * 8;
        const long m = m0 + row;
        if (m < p.M)
            *reinterpret_cast<uint4 *>(p.out + m * p.Co + n0 + c8) = *reinterpret_cast<const uint4 *>(O + row * LDO + c8);
    }
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_conv3x3s2_nhwc_bf16(int B, int T1, int F1, int Ci, int Co, const void *in, const void *w_tap_co_ci,
                                        const void *bias, void *out, int relu, pafc_stream_t stream) {
    if (!in || !w_tap_co_ci || !out) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T1 < 3 || F1 < 3 || Ci <= 0 || Co <= 0 || Ci % pafc::BK || Co % pafc::BN) return PAFC_ERR_BAD_DIMS;
    pafc::ConvParams p{};
    p.in = (const pafc::bf16_t *)in; p.wt = (const pafc::bf16_t *)w_tap_co_ci; p.bias = (const pafc::bf16_t *)bias;
    p.out = (pafc::bf16_t *)out;
    p.B = B; p.T1 = T1; p.F1 = F1; p.Ci = Ci; p.Co = Co;
    p.T2 = (T1 - 3) / 2 + 1; p.F2 = (F1 - 3) / 2 + 1;
    p.M = (long)B * p.T2 * p.F2;
    p.relu = relu;
    p.mtiles = (int)((p.M + pafc::BM - 1) / pafc::BM);
    p.ntiles = Co / pafc::BN;
    const size_t lds = 2 * 2 * pafc::BM * pafc::BK * sizeof(pafc::bf16_t);   // 64 KiB
    if (hipFuncSetAttribute((const void *)pafc::conv3x3s2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return PAFC_ERR_LAUNCH;
    const long nblk = (long)p.mtiles * p.ntiles;
    hipLaunchKernelGGL(pafc::conv3x3s2_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
