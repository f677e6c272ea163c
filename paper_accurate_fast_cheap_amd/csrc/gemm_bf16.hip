// Dense projections of the encoder as a hand-written bf16 GEMM with fused epilogues (gfx950).
// C ABI: include/pafc_encoder_ops.h: pafc_gemm_bf16.
//
//   out[z][m][n] = act(alpha * sum_k A[z][m][k] * W[z][n][k] + bias[z][n] + residual[z][m][n])
// i.e. nn.Linear (weight stored (N, K), K contiguous) with the bias, the activation (SiLU / tanh / ReLU), the
// ff_scale and the residual add applied to the fp32 accumulator before the single rounding to bf16.  Replaces the
// FFN / 1x1-conv / r,k,v / output projections of ConformerEncoderLayer, ConvolutionModule and RWKV_Tmix_x060c
// (wenet/transformer/positionwise_feed_forward.py:47-55, convolution.py:118-141, rwkv_v6/src/model.py:286-324,
// encoder_layer.py:201-259) -- 55 % of the encoder pass.  z = batch (grid.z) for the stacked projections of the two
// directions.
//
// Same skeleton as the subsampling convolution (conv_sub.hip): block = 256 threads (2 x 2 waves), tile 128 x 128 x 64,
// each wave 4 x 4 MFMA 16x16x32 tiles; A and W tiles go global -> LDS by LDS-DMA with the XOR swizzle on the source
// side, two LDS stages; tiles are numbered so that the N-tiles of one M-tile run on one XCD (they share the A tile
// in that XCD's L2).  Epilogue: the accumulators (+ bias, activation) are staged through LDS -- as fp32 when a
// residual has to be added (the residual is then read with full 16-byte rows and added before the rounding), as
// bf16 otherwise -- and leave as whole 256-byte rows.
// (Tried and measured, not kept: a persistent tile loop that prefetches the next tile's first K-step across the
// epilogue -- 177 vs 165 us on the FFN shape: the wait for the prefetch also waits for the tile's output stores.)
#include <stdio.h>
#include <stdlib.h>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int GBM = 128, GBN = 128, GBK = 64;
typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));

struct GemmParams {
    const bf16_t *A, *W, *bias, *res;
    bf16_t *out;
    long M;
    int N, K;
    long lda, ldw, ldo, ldr;          // row strides (elements)
    long sA, sW, sO, sB, sR;          // batch strides (elements); sB = 0 shares the bias
    float alpha;
    int act;                          // 0 none, 1 SiLU, 2 tanh, 3 ReLU
    int mtiles, ntiles;
    // fp32-output forms (EPI 3 / 4, pafc_gemm_bf16_f32out): bias / residual are fp32 (the bf16_t pointers above are reinterpreted),
    // out fp32 or bf16 planes hi | lo (lo at column offset lo_off); split-operand A (SPL): A = planes [hi K1 | lo K1] of an fp32
    // activation, W = [hi | hi | lo] (N x 3 K1), K = 3 K1 columns walked as hi, lo, hi of A -- nk1 = K1 / 64 K-steps per segment
    int nk1, K1;
    int pb_sh;                        // > 0: A's planes alternate in blocks of 1 << pb_sh columns ([hi PB | lo PB] ...), else [hi K1 | lo K1]
    long lo_off;
    // K split over blockIdx.y (ksplit > 1, pafc_gemm_bf16_f32out with a workspace): block z walks its share of the K-steps of the
    // SAME operands and leaves a raw fp32 partial (EPI 3 with alpha 1, no bias, no residual) at out + z * sO
    int ksplit;
};

__device__ __forceinline__ void gdma16(const bf16_t *src, bf16_t *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    if (act == 2) {   // tanh(v) = 1 - 2 / (exp(2v) + 1); saturates correctly at +-inf
        return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
    }
    if (act == 3) return fmaxf(v, 0.f);
    return v;
}

// EPI: 0 = plain (bf16 staging), 1 = residual (fp32 staging), 2 = GLU (the tile's columns [0, 64) are values, [64, 128)
// the gates of the same 64 output channels; the output has N / 2 columns)
// BM x BN = the block's tile: 128 x 128 (two blocks per CU), or -- for the few-thousand-row problems whose 128 x 128 tiles
// would leave half the CUs idle (a batch of 2 000-frame windows, 16-64 concurrent streams, a c2 decode batch) -- 128 x 64 or
// 64 x 64: same loop, the wave's share of the tile shrinks (MI x NI MFMA tiles of 16 x 16), more tiles fill the chip.
// NST = LDS stages: 2 for 128 x 128 (two blocks of 64 KiB per CU), 3 for the smaller tiles -- a few-thousand-row problem
// with a long K (w_2: 32 K-steps) is bound by the latency of each K-step's operands, one more step in flight hides it.
// EPI 3 / 4 (round 6): fp32 results -- 3 = fp32 out (+ fp32 bias, + fp32 residual), 4 = the fp32 result as bf16 planes hi | lo --
// and SPL = a split-operand A: the fp32 projections of a model with the bf16 slot at a FEW HUNDRED to a few thousand rows (a
// single 2 000-frame window is 499), where the 256-wide phase-pipelined tiles cannot fill the chip and exact fp32 products on the
// fp32 matrix cores cost four times the matrix time of three bf16 products.
template <int EPI, int BM = GBM, int BN = GBN, int NST = 2, bool SPL = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
    constexpr bool HAS_RES = EPI == 1;
    constexpr int MI = BM / 32, NI = BN / 32;      // 16 x 16 MFMA tiles per wave along m / n (2 x 2 waves)
    static_assert(EPI != 2 || (BM == 128 && BN == 128), "GLU: value | gate blocks of 64 columns need the 128 x 128 tile");
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];   // [2 stages][A BM x 64 | W BN x 64] (64 KiB at 128 x 128)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.y;

    const long nblk = (long)p.mtiles * p.ntiles;
    long bid = blockIdx.x;
    const long per = nblk / 8;
    if (bid < per * 8) bid = (bid % 8) * per + bid / 8;   // XCD-aware order; the tail keeps its id
    const int mt0 = (int)(bid / p.ntiles), nt0 = (int)(bid % p.ntiles);
    const long m0 = (long)mt0 * BM;
    const int n0 = nt0 * BN;

    const bf16_t *Az = p.A + z * p.sA, *Wz = p.W + z * p.sW;
    const int sub = lane >> 3, pch = lane & 7;
    const bf16_t *a_src[MI];                               // a wave stages BM / 4 rows of A and BN / 4 rows of W, 8 per instruction
    const bf16_t *w_src[NI];
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int row = wave * (BM / 4) + j * 8 + sub;
        const int c = pch ^ (row & 7);
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;                          // clamp: the row is computed but never stored
        a_src[j] = Az + m * p.lda + 8 * c;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int row = wave * (BN / 4) + j * 8 + sub;
        const int c = pch ^ (row & 7);
        const int n = min(n0 + row, p.N - 1);               // N tail: clamped rows feed columns that are never stored
        w_src[j] = Wz + (long)n * p.ldw + 8 * c;
    }
    int iters = p.K / GBK, it0 = 0;
    if (p.ksplit > 1) {                                     // this block's share of the K-steps
        const int per = (iters + p.ksplit - 1) / p.ksplit;
        it0 = z * per;
        iters = max(0, min(iters, it0 + per) - it0);
    }
    constexpr int STAGE = (BM + BN) * GBK;                  // elements per stage

    auto issue = [&](int it, int buf) {
        bf16_t *A = lds + buf * STAGE;
        bf16_t *Wt = A + BM * GBK;
        const int git = it0 + it;                            // K-step of the whole product
        const int koff = git * GBK;
        int koffA = koff;
        if constexpr (SPL) {                                 // A walks hi, lo, hi while W walks [hi | hi | lo] straight through
            const int seg = (git >= p.nk1) + (git >= 2 * p.nk1);
            const int k0 = (git - seg * p.nk1) * GBK;
            if (p.pb_sh > 0)     // (the subsampling convolution's plane output: [hi C | lo C] per frequency bin)
                koffA = ((k0 >> p.pb_sh) << (p.pb_sh + 1)) + (k0 & ((1 << p.pb_sh) - 1)) + (seg == 1 ? (1 << p.pb_sh) : 0);
            else
                koffA = (seg == 1 ? p.K1 : 0) + k0;
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) gdma16(a_src[j] + koffA, A + (wave * (BM / 4) + j * 8) * GBK);
#pragma unroll
        for (int j = 0; j < NI; ++j) gdma16(w_src[j] + koff, Wt + (wave * (BN / 4) + j * 8) * GBK);
    };

    f32x4g acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4g{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, kq = lane >> 4;
    issue(0, 0);
    if constexpr (NST >= 3) {
#pragma unroll
        for (int s_ = 1; s_ < NST - 1; ++s_)
            if (iters > s_) issue(s_, s_);
    }
    for (int it = 0; it < iters; ++it) {
        if constexpr (NST >= 3) {
            // a ring of NST stages with NST - 1 in flight: stage `it` has landed when at most the NST - 2 stages issued after it
            // are outstanding (loads complete in order; fewer near the end of K).  NST = 3 for the few-thousand-row problems;
            // NST = 6 (round 6) for the few-HUNDRED-row ones, whose single block per CU walks up to 96 K-steps of a split-operand
            // K = 2048 with nothing but its own ring to hide the ~1 us L2 -> LDS round trip of each
            const int newer = min(NST - 2, iters - 1 - it);
            switch (newer) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * (MI + NI)) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST >= 4 ? 2 : 1) * (MI + NI)) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST >= 5 ? 3 : 1) * (MI + NI)) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST >= 6 ? 4 : 1) * (MI + NI)) : "memory"); break;
            }
            // a RAW barrier: __syncthreads() compiles to s_waitcnt vmcnt(0) + s_barrier, which drains every stage in flight and
            // turns the ring into one exposed L2 -> LDS round trip per K-step (measured in round 6: 0.54 us per K-step whatever the
            // ring's depth).  Each wave has waited for ITS OWN loads of stage `it` above; behind the barrier every wave's have landed.
            // Everyone is also done reading stage it - 1 (its fragments are in registers: the MFMAs that consumed them were issued
            // before this point in program order) = the buffer stage it + NST - 1 is written to.
            asm volatile("s_barrier" ::: "memory");
            if (it + NST - 1 < iters) issue(it + NST - 1, (it + NST - 1) % NST);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it + 1 < iters) issue(it + 1, (it + 1) & 1);
        }
        const bf16_t *A = lds + (NST >= 3 ? it % NST : (it & 1)) * STAGE;
        const bf16_t *Wt = A + BM * GBK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8g af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * (BM / 2) + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8g *>(A + row * GBK + (((ks * 4 + kq) ^ (row & 7)) * 8));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int row = wn * (BN / 2) + j * 16 + fr;
                wf[j] = *reinterpret_cast<const bf16x8g *>(Wt + row * GBK + (((ks * 4 + kq) ^ (row & 7)) * 8));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();   // operand buffers are free: reuse them to stage the output tile

    // ---- epilogue: C/D layout col = lane & 15 (n), row = 4 (lane >> 4) + reg (m) --------------------------------
    const bf16_t *bz = p.bias ? p.bias + z * p.sB : nullptr;
    bf16_t *Oz = p.out + z * p.sO;
    if constexpr (EPI == 3 || EPI == 4) {
        constexpr int LDF = BN + 4;   // fp32 staging [BM][BN + 4]
        float *O = reinterpret_cast<float *>(lds);
        const float *bf = p.bias ? reinterpret_cast<const float *>(p.bias) + z * p.sB : nullptr;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn * (BN / 2) + j * 16 + fr;
            const float bv = bf ? bf[min(n0 + col, p.N - 1)] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    O[(wm * (BM / 2) + i * 16 + 4 * kq + g) * LDF + col] = apply_act(fmaf(acc[i][j][g], p.alpha, bv), p.act);
        }
        __syncthreads();
        if constexpr (EPI == 3) {
            const float *Rf = p.res ? reinterpret_cast<const float *>(p.res) + z * p.sR : nullptr;
            float *Of = reinterpret_cast<float *>(p.out) + z * p.sO;
            constexpr int CPR = BN / 4, RPP = 256 / CPR;   // 16-byte chunks (4 floats) per row, rows per pass
#pragma unroll
            for (int q = 0; q < BM / RPP; ++q) {
                const int row = q * RPP + tid / CPR, c4 = (tid % CPR) * 4;
                const long m = m0 + row;
                if (m < p.M && n0 + c4 < p.N) {
                    float4 o = *reinterpret_cast<const float4 *>(O + row * LDF + c4);
                    if (Rf) {
                        const float4 r = *reinterpret_cast<const float4 *>(Rf + m * p.ldr + n0 + c4);
                        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                    }
                    *reinterpret_cast<float4 *>(Of + m * p.ldo + n0 + c4) = o;
                }
            }
        } else {
            constexpr int CPR = BN / 8, RPP = 256 / CPR;
#pragma unroll
            for (int q = 0; q < BM / RPP; ++q) {
                const int row = q * RPP + tid / CPR, c8 = (tid % CPR) * 8;
                const long m = m0 + row;
                if (m < p.M && n0 + c8 < p.N) {
                    const float4 o0 = *reinterpret_cast<const float4 *>(O + row * LDF + c8);
                    const float4 o1 = *reinterpret_cast<const float4 *>(O + row * LDF + c8 + 4);
                    const float o[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                    uint32_t h[8], l[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        h[e] = f32_to_bf16_bits(o[e]);
                        l[e] = f32_to_bf16_bits(o[e] - bf16_bits_to_f32(h[e]));
                    }
                    uint4 wh, wl;
                    wh.x = h[0] | (h[1] << 16); wh.y = h[2] | (h[3] << 16); wh.z = h[4] | (h[5] << 16); wh.w = h[6] | (h[7] << 16);
                    wl.x = l[0] | (l[1] << 16); wl.y = l[2] | (l[3] << 16); wl.z = l[4] | (l[5] << 16); wl.w = l[6] | (l[7] << 16);
                    *reinterpret_cast<uint4 *>(Oz + m * p.ldo + n0 + c8) = wh;
                    *reinterpret_cast<uint4 *>(Oz + m * p.ldo + p.lo_off + n0 + c8) = wl;
                }
            }
        }
    } else if constexpr (HAS_RES) {
        constexpr int LDF = BN + 4;   // fp32 staging [BM][BN + 4] (66 KiB at 128 x 128)
        float *O = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn * (BN / 2) + j * 16 + fr;
            const float bv = bz ? bf16_bits_to_f32(bz[min(n0 + col, p.N - 1)]) : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    O[(wm * (BM / 2) + i * 16 + 4 * kq + g) * LDF + col] = apply_act(fmaf(acc[i][j][g], p.alpha, bv), p.act);
        }
        __syncthreads();
        const bf16_t *Rz = p.res + z * p.sR;
        constexpr int CPR = BN / 8, RPP = 256 / CPR;       // 16-byte chunks per row, rows per pass of the 256 threads
#pragma unroll
        for (int q = 0; q < BM / RPP; ++q) {
            const int row = q * RPP + tid / CPR, c8 = (tid % CPR) * 8;
            const long m = m0 + row;
            if (m < p.M && n0 + c8 < p.N) {
                const uint4 rq = *reinterpret_cast<const uint4 *>(Rz + m * p.ldr + n0 + c8);
                float r[8];
                Elem<bf16_t>::unpack(rq, r);
                const float4 o0 = *reinterpret_cast<const float4 *>(O + row * LDF + c8);
                const float4 o1 = *reinterpret_cast<const float4 *>(O + row * LDF + c8 + 4);
                uint4 w;
                w.x = f32_to_bf16_bits(o0.x + r[0]) | (f32_to_bf16_bits(o0.y + r[1]) << 16);
                w.y = f32_to_bf16_bits(o0.z + r[2]) | (f32_to_bf16_bits(o0.w + r[3]) << 16);
                w.z = f32_to_bf16_bits(o1.x + r[4]) | (f32_to_bf16_bits(o1.y + r[5]) << 16);
                w.w = f32_to_bf16_bits(o1.z + r[6]) | (f32_to_bf16_bits(o1.w + r[7]) << 16);
                *reinterpret_cast<uint4 *>(Oz + m * p.ldo + n0 + c8) = w;
            }
        }
    } else if constexpr (EPI == 2) {
        // waves wn = 1 hold the gates of exactly the elements waves wn = 0 hold (same lane, same register): the gate
        // passes through LDS as fp32 in the accumulator layout, the product a * sigmoid(b) is rounded once
        constexpr int LDG = 64 + 4;
        float *G = reinterpret_cast<float *>(lds);               // [128][68] fp32 = 34 KiB
        constexpr int LDH = 64 + 8;
        bf16_t *O = lds + 128 * LDG * 2 + 512;                    // behind the gates: [128][72] bf16 = 18 KiB
        if (wn == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = j * 16 + fr;
                const float bv = bz ? bf16_bits_to_f32(bz[n0 + 64 + col]) : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float b = fmaf(acc[i][j][g], p.alpha, bv);
                        G[(wm * 64 + i * 16 + 4 * kq + g) * LDG + col] = __builtin_amdgcn_rcpf(1.f + __expf(-b));
                    }
            }
        }
        __syncthreads();
        if (wn == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = j * 16 + fr;
                const float bv = bz ? bf16_bits_to_f32(bz[n0 + col]) : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = wm * 64 + i * 16 + 4 * kq + g;
                        const float a = fmaf(acc[i][j][g], p.alpha, bv);
                        O[row * LDH + col] = (bf16_t)f32_to_bf16_bits(a * G[row * LDG + col]);
                    }
            }
        }
        __syncthreads();
        const int oc0 = n0 / 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = q * 32 + (tid >> 3), c8 = (tid & 7) * 8;
            const long m = m0 + row;
            if (m < p.M)
                *reinterpret_cast<uint4 *>(Oz + m * p.ldo + oc0 + c8) = *reinterpret_cast<const uint4 *>(O + row * LDH + c8);
        }
    } else {
        constexpr int LDO = BN + 8;
        bf16_t *O = lds;   // [BM][BN + 8] bf16 (34 KiB at 128 x 128)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn * (BN / 2) + j * 16 + fr;
            const float bv = bz ? bf16_bits_to_f32(bz[min(n0 + col, p.N - 1)]) : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    O[(wm * (BM / 2) + i * 16 + 4 * kq + g) * LDO + col] =
                        (bf16_t)f32_to_bf16_bits(apply_act(fmaf(acc[i][j][g], p.alpha, bv), p.act));
        }
        __syncthreads();
        constexpr int CPR = BN / 8, RPP = 256 / CPR;
#pragma unroll
        for (int q = 0; q < BM / RPP; ++q) {
            const int row = q * RPP + tid / CPR, c8 = (tid % CPR) * 8;
            const long m = m0 + row;
            if (m < p.M && n0 + c8 < p.N)
                *reinterpret_cast<uint4 *>(Oz + m * p.ldo + n0 + c8) = *reinterpret_cast<const uint4 *>(O + row * LDO + c8);
        }
    }
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_gemm_bf16_ph(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W, long ldw,
                                 long strideW, const void *bias, long strideBias, const void *residual, long ldr, long strideR,
                                 void *out, long ldo, long strideO, float alpha, int act, int tile_n, int tile_m,
                                 pafc_stream_t stream);

// Which kernel takes a problem: the phase-pipelined 256-wide one (gemm_ph.hip; returns its rows per tile, 256 or 192) when its
// big tiles still fill the chip, else 0 = the 128 x 128 kernel below.  Rows per tile: the count that needs the fewest
// rounds of one-tile-per-CU work, weighted by the rows a round costs plus a fixed per-tile part (prologue, epilogue) worth
// about 64 rows (measured at the 30-minute shapes: tools/bench_gemm_tiles.py).
static int ph_tile_m(long M, int N, int K, int batch, int act, bool has_residual) {
    if (N % 8 || K % 128 || N < 256) return 0;              // (an even number of 64-deep K-steps: gemm_ph.hip's tile loop)
    if (act == 4 && N % 256) return 0;
    if (has_residual && act != 0) return 0;
    const int cus = pafc::device_cus();
    const long nt = (N + 255) / 256;
    // Which family: measured at 3 992 - 32 000 rows on every layer shape (profiles/r04s_gemm_tile_choice_by_rows.txt), one round
    // of the 128-wide kernel (two co-resident 128 x 128 tiles per CU) takes ~19 us at K = 512 where one round of 256-wide tiles takes
    // 23-28 us whatever its fill, and a second round of the small kernel brings it to 29-33 us.  So: the 128-wide kernel while its
    // tiles fit ONE round (<= 2 per CU), the 256-wide one beyond.  (Round 3's rule -- 256-wide tiles must cover 75 % of the CUs
    // -- sent N = 512 products of 18 000-24 000 rows, a batch of 9 000-frame windows or a long decode batch, to two rounds of
    // the small kernel: 28.9 instead of 23.7 us.)  PAFC_PH_MIN_FILL=<percent> brings that rule back for A/B runs.
    static const long min_fill = [] { const char *e = getenv("PAFC_PH_MIN_FILL"); return e ? atol(e) : 0L; }();
    if (min_fill > 0) {
        if (((M + 255) / 256) * nt * batch * 100 < (long)cus * min_fill) return 0;
    } else if (((M + 127) / 128) * ((N + 127) / 128) * batch <= 2L * cus) {
        return 0;
    }
    long best_cost = -1;
    int best = 0;
    for (int tm = 256; tm >= 192; tm -= 64) {
        const long tiles = ((M + tm - 1) / tm) * nt * batch;
        const long cost = ((tiles + cus - 1) / cus) * (tm + 64);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = tm; }
    }
    return best;
}

// GLU row order the dispatcher wants for this problem: 64 (the 128 x 128 kernel's blocks) or 32 (the phase-pipelined one's).
extern "C" int pafc_gemm_bf16_glu_half(long M, int N, int K, int batch) { return ph_tile_m(M, N, K, batch, 4, false) ? 32 : 64; }

extern "C" int pafc_gemm_bf16(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W,
                              long ldw, long strideW, const void *bias, long strideBias, const void *residual, long ldr,
                              long strideR, void *out, long ldo, long strideO, float alpha, int act,
                              pafc_stream_t stream) {
    if (!A || !W || !out) return PAFC_ERR_NULL_POINTER;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535) return PAFC_ERR_BAD_DIMS;
    if (N % 8 || K % pafc::GBK) return PAFC_ERR_UNSUPPORTED;
    if (act < 0 || act > 4) return PAFC_ERR_UNSUPPORTED;
    if (const int tm = ph_tile_m(M, N, K, batch, act, residual != nullptr)) {
        const int rc = pafc_gemm_bf16_ph(M, N, K, batch, A, lda, strideA, W, ldw, strideW, bias, strideBias, residual, ldr, strideR,
                                         out, ldo, strideO, alpha, act, 256, tm, stream);
        if (rc != PAFC_ERR_UNSUPPORTED) return rc;        // (extents beyond its 31-bit offsets fall through)
        // ... except GLU: the caller laid the weight rows out in the order pafc_gemm_bf16_glu_half announced for THIS kernel
        // (blocks of 32), the 128 x 128 kernel below reads blocks of 64 -- refuse rather than compute something else
        if (act == 4) return PAFC_ERR_UNSUPPORTED;
    }
    const bool glu = act == 4;
    if (glu && (N % pafc::GBN || residual)) return PAFC_ERR_UNSUPPORTED;
    if (lda < K || ldw < K || ldo < (glu ? N / 2 : N) || (residual && ldr < N)) return PAFC_ERR_BAD_DIMS;
    // 16-byte row segments everywhere (LDS-DMA sources, vector stores)
    if ((lda | ldw | ldo | strideA | strideW | strideO) % 8 || (residual && ((ldr | strideR) % 8))) return PAFC_ERR_ALIGNMENT;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)residual) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    pafc::GemmParams p{};
    p.A = (const pafc::bf16_t *)A; p.W = (const pafc::bf16_t *)W; p.bias = (const pafc::bf16_t *)bias;
    p.res = (const pafc::bf16_t *)residual; p.out = (pafc::bf16_t *)out;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr;
    p.sA = strideA; p.sW = strideW; p.sO = strideO; p.sB = strideBias; p.sR = strideR;
    p.alpha = alpha; p.act = act;
    // tile: 128 x 128 while its tiles give every CU one; below that 128 x 64, then 64 x 64 (a few thousand rows: a batch of
    // 2 000-frame windows, 16-64 streams, a short c2 batch) -- more, smaller tiles instead of idle CUs.  GLU keeps 128 x 128.
    const long cus = pafc::device_cus();
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * (long)batch; };
    // the largest tile that still gives two tiles per CU (two blocks are co-resident and cover each other's barriers); measured
    // at 1 024 - 8 192 rows on every layer shape: profiles/r04d_gemm_mid_rows_own_tiles_vs_library.txt.  128 x 128 from 1.4 tiles
    // per CU on (round 6: the training step's 15 392 rows x 512 columns = 484 tiles went to 128 x 64, which moves 1.5 x the bytes
    // through L2 -> LDS per flop -- 47.1 vs 43.7 us at K = 2048, 28.1 vs 24.2 at K = 1024; at 10 000 rows = 316 tiles the small
    // tiles still tie: profiles/r06x_gemm_train_tiles_by_rows.txt)
    int bm = 128, bn = 128;
    if (!glu && tiles(128, 128) * 5 < 7 * cus) {
        if (tiles(128, 64) >= 2 * cus) bn = 64;
        else { bm = 64; bn = 64; }
    }
    if (const char *e = getenv("PAFC_GEMM_TILE")) {          // A/B runs: "128x128", "128x64", "64x64"
        if (!glu && sscanf(e, "%dx%d", &bm, &bn) == 2 && !((bm == 128 && (bn == 128 || bn == 64)) || (bm == 64 && bn == 64))) {
            bm = 128; bn = 128;
        }
    }
    p.mtiles = (int)((M + bm - 1) / bm);
    p.ntiles = (N + bn - 1) / bn;
    const long nblk = (long)p.mtiles * p.ntiles;
    if (nblk > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    const dim3 grid((unsigned)nblk, (unsigned)batch);
    hipStream_t s = (hipStream_t)stream;
    const int nst = (bm == 128 && bn == 128) ? 2 : 3;
    const size_t stage_bytes = (size_t)nst * (bm + bn) * pafc::GBK * sizeof(pafc::bf16_t);        // 64 KiB at 128 x 128
    const size_t res_bytes = (size_t)bm * (bn + 4) * sizeof(float);                                // fp32 staging of the epilogue
    const size_t lds = residual ? (res_bytes > stage_bytes ? res_bytes : stage_bytes) : stage_bytes;
    typedef void (*kern_t)(const pafc::GemmParams);
    kern_t kern;
    if (glu) kern = pafc::gemm_bf16_kernel<2>;
    else if (residual) kern = bm == 64 ? pafc::gemm_bf16_kernel<1, 64, 64, 3> : bn == 64 ? pafc::gemm_bf16_kernel<1, 128, 64, 3> : pafc::gemm_bf16_kernel<1>;
    else kern = bm == 64 ? pafc::gemm_bf16_kernel<0, 64, 64, 3> : bn == 64 ? pafc::gemm_bf16_kernel<0, 128, 64, 3> : pafc::gemm_bf16_kernel<0>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PAFC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}


namespace pafc {
namespace {

// second pass of a K-split product: out = act(alpha * (sum of the S partials, in order) + bias) + residual, fp32 or planes
__global__ __launch_bounds__(256) void gemm_ksplit_reduce_kernel(long M, int N, int S, const float *part, const float *bias, float alpha,
                                                                 int act, const float *res, long ldr, void *out, int out_kind, long ldo,
                                                                 long lo_off) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;           // one float4 of the (M, N) result
    const int n4 = N / 4;
    if (q >= M * n4) return;
    const long m = q / n4;
    const int c = (int)(q % n4) * 4;
    const long MN = M * (long)N;
    float4 a = *reinterpret_cast<const float4 *>(part + m * N + c);
    for (int z = 1; z < S; ++z) {
        const float4 b = *reinterpret_cast<const float4 *>(part + z * MN + m * N + c);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float o[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = apply_act(fmaf(o[e], alpha, bias ? bias[c + e] : 0.f), act);
    if (res) {
        const float4 r = *reinterpret_cast<const float4 *>(res + m * ldr + c);
        o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    if (out_kind == 1) {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + m * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        uint32_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = f32_to_bf16_bits(o[e]);
            l[e] = f32_to_bf16_bits(o[e] - bf16_bits_to_f32(h[e]));
        }
        bf16_t *ob = reinterpret_cast<bf16_t *>(out) + m * ldo + c;
        *reinterpret_cast<uint2 *>(ob) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        *reinterpret_cast<uint2 *>(ob + lo_off) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    }
}

// Tile and K split of a product on the small tiles (pafc_gemm_bf16_f32out).  Tile: the largest that still gives two per CU.  K
// split (S blocks per tile walk K / S each, a second launch adds the partials in order: deterministic) where K is long (>= 48
// K-steps: a split-operand w_2 has 96, the subsampling Linear 456):
//   * a few hundred rows -- the 64 x 64 tiles cover less than half of the CUs: S = CUs / tiles (<= 4), 64 x 64 tiles, deep ring
//     (w_2 at 499 rows: 42 us as one chain -> 18 us);
//   * ~1 000 - 4 000 rows -- every CU has its 64 x 64 tiles, and the loop runs at the L2 -> LDS ceiling on them (774 MB per
//     launch at 3 992 rows x 512 columns): 128 x 128 tiles move half the bytes per flop, four K shares per tile keep the chip
//     full -- min(4, two blocks per CU) of them (w_2: 43.7 -> 30.6 us at 1 996 rows, 47.1 -> 41.0 at 3 992, and against the
//     256-wide kernel 76.9 -> 49.3 at 5 000 rows (3 shares), 82.5 -> 66.0 at 7 984 (2); products of 8 K-steps lose with any
//     split: profiles/r06z_split_small_tile_variants.txt).
struct F32outPlan { int bm, bn, S; };
F32outPlan f32out_plan(long M, int N, int Kw, bool may_split = true) {
    const long cus = device_cus();
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    F32outPlan pl{128, 128, 1};
    if (tiles(128, 128) < 2 * cus) {
        if (tiles(128, 64) >= 2 * cus) pl.bn = 64;
        else { pl.bm = 64; pl.bn = 64; }
    }
    const int iters = Kw / GBK;
    if (iters >= 48 && may_split) {
        const long nblk = tiles(64, 64);
        if (nblk * 2 <= cus) {
            long s = cus / nblk;
            if (s > 4) s = 4;
            if (s > iters / 12) s = iters / 12;
            if (s >= 2) { pl.bm = 64; pl.bn = 64; pl.S = (int)s; }
        } else {
            long s = 2 * cus / tiles(128, 128);          // K shares that keep <= two 128 x 128 blocks per CU
            if (s > 4) s = 4;
            if (s >= 2) { pl.bm = 128; pl.bn = 128; pl.S = (int)s; }
        }
    }
    if (const char *e = getenv("PAFC_F32OUT_TILE")) {        // A/B runs: "128x128", "128x64", "64x64"
        int fm = 0, fn = 0;
        if (sscanf(e, "%dx%d", &fm, &fn) == 2 && ((fm == 128 && (fn == 128 || fn == 64)) || (fm == 64 && fn == 64))) { pl.bm = fm; pl.bn = fn; }
    }
    if (const char *e = getenv("PAFC_F32OUT_KSPLIT")) {      // A/B runs: force the factor (1-8)
        const int v = atoi(e);
        if (v >= 1 && v <= 8 && iters >= v && may_split) pl.S = v;
    }
    return pl;
}

}  // namespace
}  // namespace pafc

extern "C" size_t pafc_gemm_bf16_f32out_workspace_bytes(long M, int N, int K, int a_split) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 8 || K % pafc::GBK) return 0;
    const int S = pafc::f32out_plan(M, N, a_split ? 3 * K : K).S;
    return S > 1 ? (size_t)S * M * N * sizeof(float) : 0;
}

// fp32 results from bf16 operands on the 128 x 128 / 128 x 64 / 64 x 64 tiles (include/pafc_encoder_ops.h): the split-operand
// projections of an fp32 model with the bf16 slot at few rows, and the slot's bf16 output projection into the fp32 residual stream.
extern "C" int pafc_gemm_bf16_f32out(long M, int N, int K, const void *A, long lda, int a_split, const void *W, long ldw,
                                     const float *bias, const float *residual, long ldr, void *out, int out_kind, long ldo,
                                     long lo_off, float alpha, int act, void *workspace, size_t workspace_bytes,
                                     pafc_stream_t stream) {
    return pafc_gemm_bf16_f32out_pb(M, N, K, A, lda, a_split, 0, W, ldw, bias, residual, ldr, out, out_kind, ldo, lo_off, alpha, act,
                                    workspace, workspace_bytes, stream);
}

// ... with a_plane_block as pafc_gemm_ph_ex2 takes it (a split A whose planes alternate in blocks of that many columns: the
// subsampling convolution's output feeding Linear(F' C, odim) -- at a few hundred to a few thousand rows that product is 2-8 tiles
// of the 256-wide kernel against K = 9 728 x 3: 400 us at 3 992 rows, 205 us at 499)
extern "C" int pafc_gemm_bf16_f32out_pb(long M, int N, int K, const void *A, long lda, int a_split, int a_plane_block, const void *W,
                                        long ldw, const float *bias, const float *residual, long ldr, void *out, int out_kind,
                                        long ldo, long lo_off, float alpha, int act, void *workspace, size_t workspace_bytes,
                                        pafc_stream_t stream) {
    if (!A || !W || !out) return PAFC_ERR_NULL_POINTER;
    int pb_sh = 0;
    if (a_plane_block) {
        if (!a_split) return PAFC_ERR_UNSUPPORTED;
        pb_sh = 6;
        while ((1 << pb_sh) < a_plane_block) ++pb_sh;
        if ((1 << pb_sh) != a_plane_block || K % a_plane_block) return PAFC_ERR_UNSUPPORTED;
    }
    if (M <= 0 || N <= 0 || K <= 0) return PAFC_ERR_BAD_DIMS;
    if (N % 8 || K % pafc::GBK) return PAFC_ERR_UNSUPPORTED;
    if ((out_kind != 1 && out_kind != 2) || act < 0 || act > 3) return PAFC_ERR_UNSUPPORTED;
    if (out_kind == 2 && residual) return PAFC_ERR_UNSUPPORTED;
    const int Kw = a_split ? 3 * K : K;
    if (lda < (a_split ? 2 * K : K) || ldw < Kw || (residual && ldr < N)) return PAFC_ERR_BAD_DIMS;
    if (out_kind == 2 ? (lo_off < N || ldo < lo_off + N) : ldo < N) return PAFC_ERR_BAD_DIMS;
    if ((lda | ldw) % 8 || (out_kind == 1 ? ldo % 4 : (ldo | lo_off) % 8) || (residual && ldr % 4)) return PAFC_ERR_ALIGNMENT;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)residual) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    pafc::GemmParams p{};
    p.A = (const pafc::bf16_t *)A; p.W = (const pafc::bf16_t *)W; p.bias = (const pafc::bf16_t *)bias;
    p.res = (const pafc::bf16_t *)residual; p.out = (pafc::bf16_t *)out;
    p.M = M; p.N = N; p.K = Kw; p.K1 = K; p.nk1 = K / pafc::GBK; p.pb_sh = pb_sh;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr; p.lo_off = lo_off;
    p.alpha = alpha; p.act = act; p.ksplit = 1;
    const long cus = pafc::device_cus();
    pafc::F32outPlan pl = pafc::f32out_plan(M, N, Kw);
    if (pl.S > 1 && !(workspace && workspace_bytes >= (size_t)pl.S * M * N * sizeof(float) && N % 4 == 0))
        pl = pafc::f32out_plan(M, N, Kw, false);         // no workspace from the caller: the tile of the unsplit product
    const int bm = pl.bm, bn = pl.bn, S = pl.S;
    p.mtiles = (int)((M + bm - 1) / bm);
    p.ntiles = (N + bn - 1) / bn;
    const long nblk = (long)p.mtiles * p.ntiles;
    if (nblk > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    // K split over blocks when the caller brought the workspace for it (pafc_gemm_bf16_f32out_workspace_bytes > 0)
    const bool split_k = S > 1;
    const int out_kind_final = out_kind;
    if (split_k) {          // pass 1: raw partials (alpha 1, no bias / activation / residual) into the workspace
        p.ksplit = S;
        p.out = (pafc::bf16_t *)workspace; p.ldo = N; p.sO = (long)M * N;
        p.bias = nullptr; p.res = nullptr; p.alpha = 1.f; p.act = 0;
        out_kind = 1;
    }
    // a deep ring when the grid is at most one 64 x 64 block per CU (few hundred rows): see the kernel's K loop
    const bool deep = bm == 64 && nblk * p.ksplit <= cus;
    const int nst = (bm == 128 && bn == 128) ? 2 : deep ? 6 : 3;
    const size_t stage_bytes = (size_t)nst * (bm + bn) * pafc::GBK * sizeof(pafc::bf16_t);
    const size_t out_bytes = (size_t)bm * (bn + 4) * sizeof(float);
    const size_t lds = out_bytes > stage_bytes ? out_bytes : stage_bytes;
    typedef void (*kern_t)(const pafc::GemmParams);
    kern_t kern;
#define PAFC_PICK(E, S)                                                                                                   \
    (deep ? (kern_t)pafc::gemm_bf16_kernel<E, 64, 64, 6, S>                                                                   \
          : bm == 64 ? (kern_t)pafc::gemm_bf16_kernel<E, 64, 64, 3, S> : bn == 64 ? (kern_t)pafc::gemm_bf16_kernel<E, 128, 64, 3, S> \
                                                                                : (kern_t)pafc::gemm_bf16_kernel<E, 128, 128, 2, S>)
    if (out_kind == 1) kern = a_split ? PAFC_PICK(3, true) : PAFC_PICK(3, false);
    else kern = a_split ? PAFC_PICK(4, true) : PAFC_PICK(4, false);
#undef PAFC_PICK
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PAFC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)p.ksplit), dim3(256), lds, (hipStream_t)stream, p);
    if (split_k) {
        const long q = M * (long)(N / 4);
        hipLaunchKernelGGL(pafc::gemm_ksplit_reduce_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, N, S,
                           (const float *)workspace, bias, alpha, act, residual, ldr, out, out_kind_final, ldo, lo_off);
    }
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
