// Skinny bf16 GEMM for the streaming chunk step: out = act(alpha * A W^T + bias [+ residual]) with FEW rows (a 2.56-s chunk of
// one stream is 64-80 rows) -- the projections of a Conformer layer while it serves chunks with recurrent-state carry
// (positionwise_feed_forward.py:47-55, convolution.py:118-141, src/model.py:273-324, encoder_layer.py:201-259 in the
// reference; its forward_chunk runs them through the framework's GEMM).
//
// At this size nothing is bound by bandwidth or by the matrix cores: a step is ~250 dependent launches of ~5 us, so what
// counts is the length of ONE launch's critical path and how many launches there are.  Hence:
//   * a block owns 16 output columns (GLU: 16 value + the 16 matching gate columns) of a group of MT 16-row tiles and walks the
//     whole K; the grid is (N / 16) x row groups (x batch) -- when N / 16 alone would leave most CUs idle the row tiles are
//     spread over blocks too (MT = 1) -- and nothing is exchanged between blocks;
//   * its four (long K: eight) waves split K; operands go straight from L2 to registers as MFMA fragments (the product is
//     formed transposed, W rows as the A operand, activation rows as the B operand: both are "row r, 8 consecutive k" 16-byte
//     loads, and a lane ends up with 4 consecutive output columns of its row), a whole batch of K-steps in flight before the
//     first MFMA of the batch;
//   * the partial tiles meet in LDS once; bias / activation / GLU / residual / rounding ride in the same launch, and so can
//       - the LayerNorm in FRONT of the projection, folded as in gemm_ph.hip: out = rstd (x W'^T - mean csum) + b', the row
//         statistics either summed from partials a producing launch left or formed here from the operand fragments
//         themselves (a block sees every row whole);
//       - the token shift + first lerp of the time-mix as the operand's producer (src/model.py:274-276: the fragment of
//         x + (x_prev - x) * maa_x is formed in registers from the row, its predecessor -- or the frame carried over from the
//         previous chunk -- and maa_x), which makes the LoRA down-projection of a chunk one launch;
//       - LayerNorm + SiLU as the operand's producer (the conv module's norm + activation in front of pointwise_conv2,
//         convolution.py:136-139): row statistics in a first pass over the block's rows, the fragments normalised in registers;
//       - the partial statistics of the rows it writes (for a folded LayerNorm downstream).
// Same arithmetic forms as gemm_ph.hip (fp32 accumulation, one rounding; SiLU / tanh / sigmoid through exp + rcp).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));

struct SkParams {
    const bf16_t *A, *W, *bias, *res;
    bf16_t *out;
    long M, lda, ldw, ldr, ldo, sA, sW, sR, sO, sBias;
    int N, K;              // N = rows of W (GLU: value rows [0, N/2), gate rows [N/2, N)); K % 32 == 0
    float alpha;
    int act;               // 0 none, 1 SiLU, 2 tanh, 3 ReLU, 4 GLU (the codes of pafc_gemm_bf16)
    int round_first;       // out = bf16(alpha * acc) + bias rounded again: where an op chain `x @ W` then `+ b` rounds (model.py:289)
    const float *st_in;    // LayerNorm folded in front: float2 [M][parts_in] partial (sum, sum of squares) of the A rows, or null
    int parts_in;
    int ln_self;           // ... or: the statistics are formed here from the A fragments
    const float *csum;     // [N] column sums of the folded weight
    float eps, inv_c;
    float *st_out;         // float2 [M][N_out / 16]: partial statistics of the rows written, or null
    const bf16_t *nrm_g, *nrm_b;   // the operand is silu(LayerNorm(a)) (convolution.py:136-137 before pointwise_conv2): gamma, beta [K], or null
    float nrm_eps;
    const bf16_t *mix_maa; // token shift + lerp as the operand's producer: maa_x [K], or null
    const bf16_t *mix_prev;// [M / T][K] the frame before each sequence (streaming carry), or null = zero
    int T;                 // rows per sequence (mix)
    const bf16_t *lora_x;  // PROD 3: the operand is bf16(tanh(x W1^T)) (K = 64 wide), formed here from x [M][K1] (row stride ldx) ...
    const bf16_t *lora_w1; // ... and W1 [K][K1] (K1 innermost): the decay LoRA's two products in one launch (src/model.py:286-289)
    long ldx;
    int K1;
};

__device__ __forceinline__ float sk_act(float v, int act) {
    if (act == 1) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    if (act == 2) return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
    if (act == 3) return fmaxf(v, 0.f);
    return v;
}
__device__ __forceinline__ void sk_unpack(const uint4 q, float (&f)[8]) {
    f[0] = bf16_bits_to_f32(q.x & 0xffffu); f[1] = __uint_as_float(q.x & 0xffff0000u);
    f[2] = bf16_bits_to_f32(q.y & 0xffffu); f[3] = __uint_as_float(q.y & 0xffff0000u);
    f[4] = bf16_bits_to_f32(q.z & 0xffffu); f[5] = __uint_as_float(q.z & 0xffff0000u);
    f[6] = bf16_bits_to_f32(q.w & 0xffffu); f[7] = __uint_as_float(q.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned sk_pack(float lo, float hi) {   // both already bf16 values
    return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}

// MT: 16-row tiles per block (rows beyond M are clamped on load and skipped on store); GLU doubles the weight fragments and
// accumulators; NWV waves split K; MIX: the A fragments are formed from x, its predecessor row and maa_x.
template <int MT, bool GLU, int NWV, int PROD>
__global__ __launch_bounds__(NWV * 64) void gemm_skinny_kernel(const SkParams p) {
    constexpr int NB = GLU ? 2 : 1;
    constexpr bool MIX = PROD == 1, NRM = PROD == 2, LOR = PROD == 3;
    constexpr int KB = (MIX || NRM) ? 2 : (MT <= 2 ? 8 : (MT <= 5 ? 4 : 2));   // K-steps (of 32) whose operands are in flight together
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, qq = lane >> 4;
    const int z = blockIdx.z;
    const int n_out = GLU ? p.N / 2 : p.N;
    const int n0 = blockIdx.x * 16;
    const long m0 = (long)blockIdx.y * (MT * 16);
    const bf16_t *A = p.A + z * p.sA, *W = p.W + z * p.sW;
    const int nks = p.K / 32;                             // K-steps in all; this wave's share:
    const int ks0 = (int)((long)wave * nks / NWV), ks1 = (int)((long)(wave + 1) * nks / NWV);
    __shared__ __attribute__((aligned(16))) float s_part[NWV][MT * NB][4][64];   // [wave][tile][reg][lane]
    __shared__ float s_red[NWV][MT * 16][2];             // per-wave (sum, sum of squares) of the rows' K-shares (ln_self)
    __shared__ float s_stat[MT * 16][2];                 // rstd, -mean * rstd of the block's rows (LayerNorm fold)

    const bf16_t *wrow[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) wrow[g] = W + (long)(n0 + g * n_out + r16) * p.ldw + 8 * qq;
    const bf16_t *arow[MT], *nrow[MIX ? MT : 1];
    bool has_nb[MIX ? MT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        long r = m0 + i * 16 + r16;
        if (r > p.M - 1) r = p.M - 1;
        arow[i] = LOR ? nullptr : A + r * p.lda + 8 * qq;
        if constexpr (MIX) {
            const long b = r / p.T, t = r - b * p.T;
            has_nb[i] = t > 0 || p.mix_prev != nullptr;
            nrow[i] = t > 0 ? arow[i] - p.lda : (p.mix_prev != nullptr ? p.mix_prev + b * p.K + 8 * qq : arow[i]);
        }
    }
    // NRM: LayerNorm + SiLU as the operand's producer.  First pass: each wave sums its K-share of the block's rows (the
    // fragments it will multiply later: they come back from L1 / L2), the shares meet in LDS -> rstd, -mean * rstd per row
    float n_rstd[NRM ? MT : 1], n_nm[NRM ? MT : 1];
    if constexpr (NRM) {
        float q1[MT], q2[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) { q1[i] = 0.f; q2[i] = 0.f; }
        for (int ks = ks0; ks < ks1; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float f[8];
                sk_unpack(*reinterpret_cast<const uint4 *>(arow[i] + ks * 32), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { q1[i] += f[e]; q2[i] = fmaf(f[e], f[e], q2[i]); }
            }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s1 = q1[i], s2 = q2[i];
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (qq == 0) { s_red[wave][i * 16 + r16][0] = s1; s_red[wave][i * 16 + r16][1] = s2; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) { s1 += s_red[w][i * 16 + r16][0]; s2 += s_red[w][i * 16 + r16][1]; }
            const float mean = s1 * p.inv_c;
            const float var = fmaxf(fmaf(-mean, mean, s2 * p.inv_c), 0.f);
            n_rstd[i] = rsqrtf(var + p.nrm_eps);
            n_nm[i] = -mean * n_rstd[i];
        }
        __syncthreads();                                  // (s_red is used again below when ln_self is on)
    }
    // LOR: the 64-wide hidden tile t = bf16(tanh(x W1^T)) of the block's rows, formed here (every column block forms it again:
    // 16 rows x 64 x K1 multiply-adds, nothing next to a second launch) with the K split, the partial sums' order and the
    // roundings of the separate tanh GEMM, and left in LDS as the operand rows of the main product.
    __shared__ __attribute__((aligned(16))) bf16_t s_hid[LOR ? MT * 16 : 1][LOR ? 64 + 8 : 8];
    if constexpr (LOR) {
        static_assert(MT == 1 && !GLU && NWV == 4, "LOR: one row tile per block, four waves (wave tau finishes hidden tile tau)");
        __shared__ float s_lor[NWV][4][4][64];            // [wave][tau][reg][lane]: the waves' K-shares of the hidden tile
        const int nk1 = p.K1 / 32;
        const int k0 = (int)((long)wave * nk1 / NWV), k1 = (int)((long)(wave + 1) * nk1 / NWV);
        long r = m0 + r16;
        if (r > p.M - 1) r = p.M - 1;
        const bf16_t *xrow = p.lora_x + r * p.ldx + 8 * qq;
        f32x4s a1[4] = {f32x4s{0.f, 0.f, 0.f, 0.f}, f32x4s{0.f, 0.f, 0.f, 0.f}, f32x4s{0.f, 0.f, 0.f, 0.f}, f32x4s{0.f, 0.f, 0.f, 0.f}};
        for (int kb = k0; kb < k1; kb += 4) {            // four K-steps' operands in flight together
            uint4 xf[4], wf1[4][4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                if (kb + s_ < k1) {
                    xf[s_] = *reinterpret_cast<const uint4 *>(xrow + (kb + s_) * 32);
#pragma unroll
                    for (int tau = 0; tau < 4; ++tau)
                        wf1[s_][tau] = *reinterpret_cast<const uint4 *>(p.lora_w1 + (long)(16 * tau + r16) * p.K1 + (kb + s_) * 32 + 8 * qq);
                }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                if (kb + s_ < k1) {
#pragma unroll
                    for (int tau = 0; tau < 4; ++tau)
                        a1[tau] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, wf1[s_][tau]),
                                                                          __builtin_bit_cast(bf16x8s, xf[s_]), a1[tau], 0, 0, 0);
                }
        }
#pragma unroll
        for (int tau = 0; tau < 4; ++tau)
#pragma unroll
            for (int e = 0; e < 4; ++e) s_lor[wave][tau][e][lane] = a1[tau][e];
        __syncthreads();
        {   // wave tau finishes hidden columns 16 tau .. + 15: lane (r16, qq) owns columns 16 tau + 4 qq .. + 3 of row r16
            const int tau = wave;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) sum += s_lor[w][tau][e][lane];
                v[e] = round_bf16(sk_act(sum, 2));
            }
            *reinterpret_cast<uint2 *>(&s_hid[r16][16 * tau + 4 * qq]) = uint2{sk_pack(v[0], v[1]), sk_pack(v[2], v[3])};
        }
        __syncthreads();
    }
    f32x4s acc[MT][NB];
    float ls1[MT], ls2[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        ls1[i] = 0.f; ls2[i] = 0.f;
#pragma unroll
        for (int g = 0; g < NB; ++g) acc[i][g] = f32x4s{0.f, 0.f, 0.f, 0.f};
    }
    const bool self_stats = p.ln_self != 0;

    auto step_batch = [&](int ks, auto nsteps) {
        constexpr int NS = decltype(nsteps)::value;
        uint4 wf[NS][NB], af[NS][MT], nf[MIX ? NS : 1][MIX ? MT : 1], mf[(MIX || NRM) ? NS : 1], bfq[NRM ? NS : 1];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = (ks + s) * 32;
#pragma unroll
            for (int g = 0; g < NB; ++g) wf[s][g] = *reinterpret_cast<const uint4 *>(wrow[g] + k);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (LOR) af[s][i] = *reinterpret_cast<const uint4 *>(&s_hid[i * 16 + r16][k + 8 * qq]);
                else af[s][i] = *reinterpret_cast<const uint4 *>(arow[i] + k);
            }
            if constexpr (MIX) {
                mf[s] = *reinterpret_cast<const uint4 *>(p.mix_maa + k + 8 * qq);
#pragma unroll
                for (int i = 0; i < MT; ++i) nf[s][i] = *reinterpret_cast<const uint4 *>(nrow[i] + k);
            }
            if constexpr (NRM) {
                mf[s] = *reinterpret_cast<const uint4 *>(p.nrm_g + k + 8 * qq);
                bfq[s] = *reinterpret_cast<const uint4 *>(p.nrm_b + k + 8 * qq);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                uint4 a = af[s][i];
                if constexpr (MIX) {             // xxx = x + (x_prev - x) * maa_x, each op rounded as the reference's op chain does
                    float xc[8], xn[8], mm[8], o[8];
                    sk_unpack(a, xc); sk_unpack(nf[s][i], xn); sk_unpack(mf[s], mm);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xx = round_bf16((has_nb[i] ? xn[e] : 0.f) - xc[e]);
                        o[e] = round_bf16(xc[e] + round_bf16(xx * mm[e]));
                    }
                    a = uint4{sk_pack(o[0], o[1]), sk_pack(o[2], o[3]), sk_pack(o[4], o[5]), sk_pack(o[6], o[7])};
                }
                if constexpr (NRM) {             // silu(bf16(LN(a))), rounded again: the two kernels' roundings
                    float xc[8], gg[8], bb[8], o[8];
                    sk_unpack(a, xc); sk_unpack(mf[s], gg); sk_unpack(bfq[s], bb);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float y = round_bf16(fmaf(fmaf(xc[e], n_rstd[i], n_nm[i]), gg[e], bb[e]));
                        o[e] = round_bf16(y * __builtin_amdgcn_rcpf(1.f + __expf(-y)));
                    }
                    a = uint4{sk_pack(o[0], o[1]), sk_pack(o[2], o[3]), sk_pack(o[4], o[5]), sk_pack(o[6], o[7])};
                }
                if (self_stats) {
                    float f[8];
                    sk_unpack(a, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ls1[i] += f[e]; ls2[i] = fmaf(f[e], f[e], ls2[i]); }
                }
#pragma unroll
                for (int g = 0; g < NB; ++g)
                    acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, wf[s][g]),
                                                                        __builtin_bit_cast(bf16x8s, a), acc[i][g], 0, 0, 0);
            }
    };
    int ks = ks0;
    for (; ks + KB <= ks1; ks += KB) step_batch(ks, std::integral_constant<int, KB>{});
    for (; ks < ks1; ++ks) step_batch(ks, std::integral_constant<int, 1>{});

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) s_part[wave][i * NB + g][e][lane] = acc[i][g][e];
    if (self_stats) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s1 = ls1[i], s2 = ls2[i];
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (qq == 0) { s_red[wave][i * 16 + r16][0] = s1; s_red[wave][i * 16 + r16][1] = s2; }
        }
    }
    __syncthreads();
    if (p.st_in != nullptr || self_stats) {              // thread t: statistics of the block's row t
        const int t = threadIdx.x;
        if (t < MT * 16) {
            float s1 = 0.f, s2 = 0.f;
            if (self_stats) {
#pragma unroll
                for (int w = 0; w < NWV; ++w) { s1 += s_red[w][t][0]; s2 += s_red[w][t][1]; }
            } else {
                long r = m0 + t;
                if (r > p.M - 1) r = p.M - 1;
                const float2 *sp = reinterpret_cast<const float2 *>(p.st_in) + r * p.parts_in;
                for (int j = 0; j < p.parts_in; ++j) { const float2 v = sp[j]; s1 += v.x; s2 += v.y; }
            }
            const float mean = s1 * p.inv_c;
            const float var = fmaxf(fmaf(-mean, mean, s2 * p.inv_c), 0.f);
            const float rstd = rsqrtf(var + p.eps);
            s_stat[t][0] = rstd;
            s_stat[t][1] = -mean * rstd;
        }
        __syncthreads();
    }
    // epilogue: wave w finishes the row tiles w, w + NWV, ...; lane (r16, qq) owns columns n0 + 4 qq .. + 3 of row r16
    const bool folded = p.st_in != nullptr || self_stats;
    for (int i = wave; i < MT; i += NWV) {
        const long row = m0 + i * 16 + r16;
        const int col = n0 + 4 * qq;
        float v[NB][4];
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) s += s_part[w][i * NB + g][e][lane];
                v[g][e] = s;
            }
        float o[4];
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const int c = col + g * n_out;
            if (folded) {
                const float rstd = s_stat[i * 16 + r16][0], nm = s_stat[i * 16 + r16][1];
                const float4 cs = *reinterpret_cast<const float4 *>(p.csum + c);
                v[g][0] = fmaf(nm, cs.x, rstd * v[g][0]); v[g][1] = fmaf(nm, cs.y, rstd * v[g][1]);
                v[g][2] = fmaf(nm, cs.z, rstd * v[g][2]); v[g][3] = fmaf(nm, cs.w, rstd * v[g][3]);
            }
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr) {
                const uint2 q = *reinterpret_cast<const uint2 *>(p.bias + z * p.sBias + c);
                b[0] = bf16_bits_to_f32(q.x & 0xffffu); b[1] = __uint_as_float(q.x & 0xffff0000u);
                b[2] = bf16_bits_to_f32(q.y & 0xffffu); b[3] = __uint_as_float(q.y & 0xffff0000u);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[g][e] = p.round_first ? b[e] + round_bf16(p.alpha * v[g][e]) : fmaf(p.alpha, v[g][e], b[e]);
        }
        if constexpr (GLU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[0][e] * __builtin_amdgcn_rcpf(1.f + __expf(-v[1][e]));
        } else {
            if (p.res != nullptr && row < p.M) {
                const uint2 q = *reinterpret_cast<const uint2 *>(p.res + z * p.sR + row * p.ldr + col);
                v[0][0] += bf16_bits_to_f32(q.x & 0xffffu); v[0][1] += __uint_as_float(q.x & 0xffff0000u);
                v[0][2] += bf16_bits_to_f32(q.y & 0xffffu); v[0][3] += __uint_as_float(q.y & 0xffff0000u);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = sk_act(v[0][e], p.act);
        }
        const unsigned short h0 = f32_to_bf16_bits(o[0]), h1 = f32_to_bf16_bits(o[1]), h2 = f32_to_bf16_bits(o[2]),
                             h3 = f32_to_bf16_bits(o[3]);
        if (row < p.M)
            *reinterpret_cast<uint2 *>(p.out + z * p.sO + row * p.ldo + col) =
                uint2{(unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16)};
        if (p.st_out != nullptr) {                       // statistics of the values as stored
            const float f0 = bf16_bits_to_f32(h0), f1 = bf16_bits_to_f32(h1), f2 = bf16_bits_to_f32(h2), f3 = bf16_bits_to_f32(h3);
            float s1 = (f0 + f1) + (f2 + f3), s2 = fmaf(f0, f0, f1 * f1) + fmaf(f2, f2, f3 * f3);
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (qq == 0 && row < p.M)
                reinterpret_cast<float2 *>(p.st_out)[(z * p.M + row) * (n_out / 16) + blockIdx.x] = make_float2(s1, s2);
        }
    }
}

template <int MT, bool GLU, int NWV, int PROD>
void launch_sk(const SkParams &p, int batch, hipStream_t s) {
    const int n_out = GLU ? p.N / 2 : p.N;
    const long groups = (p.M + MT * 16 - 1) / (MT * 16);
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, GLU, NWV, PROD>), dim3((unsigned)(n_out / 16), (unsigned)groups, (unsigned)batch),
                       dim3(NWV * 64), 0, s, p);
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_gemm_skinny_bf16_ex(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W,
                                        long ldw, long strideW, const void *bias, long strideBias, const void *residual, long ldr,
                                        long strideR, void *out, long ldo, long strideO, float alpha, int act, int round_first,
                                        const float *ln_stats_in, int ln_parts_in, int ln_self, const float *ln_csum, float ln_eps,
                                        float *ln_stats_out, const void *mix_maa, const void *mix_prev, int mix_T,
                                        const void *norm_gamma, const void *norm_beta, float norm_eps, pafc_stream_t stream) {
    if (!A || !W || !out) return PAFC_ERR_NULL_POINTER;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return PAFC_ERR_BAD_DIMS;
    if (act < 0 || act > 4) return PAFC_ERR_UNSUPPORTED;
    const bool glu = act == 4;
    const int n_out = glu ? N / 2 : N;
    // fragments are 16-byte loads of 8 consecutive k of a row: whole 32-deep K-steps, 16-byte aligned rows
    if (K % 32 != 0 || N % (glu ? 32 : 16) != 0 || lda % 8 || ldw % 8 || ldo % 4 || (residual && ldr % 4) || lda < K || ldw < K ||
        ldo < n_out)
        return PAFC_ERR_UNSUPPORTED;
    if (glu && residual) return PAFC_ERR_UNSUPPORTED;
    const bool folded = ln_stats_in != nullptr || ln_self != 0;
    if (folded != (ln_csum != nullptr) || (ln_stats_in && (ln_parts_in <= 0 || ln_self))) return PAFC_ERR_BAD_DIMS;
    if (mix_maa && (mix_T <= 0 || M % mix_T != 0 || batch != 1 || glu)) return PAFC_ERR_BAD_DIMS;
    if (!mix_maa && mix_prev) return PAFC_ERR_BAD_DIMS;
    if ((norm_gamma != nullptr) != (norm_beta != nullptr) || (norm_gamma && (mix_maa || folded || glu || batch != 1)))
        return PAFC_ERR_BAD_DIMS;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)mix_maa | (uintptr_t)mix_prev | (uintptr_t)norm_gamma | (uintptr_t)norm_beta) & 15 || ((uintptr_t)out & 7) || ((uintptr_t)bias & 7) ||
        ((uintptr_t)residual & 7) || ((uintptr_t)ln_csum & 15) || ((uintptr_t)ln_stats_in & 7) || ((uintptr_t)ln_stats_out & 7))
        return PAFC_ERR_UNSUPPORTED;
    pafc::SkParams p{};
    p.A = (const pafc::bf16_t *)A; p.W = (const pafc::bf16_t *)W; p.bias = (const pafc::bf16_t *)bias;
    p.res = (const pafc::bf16_t *)residual; p.out = (pafc::bf16_t *)out;
    p.M = M; p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldo = ldo; p.sA = strideA; p.sW = strideW; p.sR = strideR; p.sO = strideO;
    p.sBias = strideBias; p.N = N; p.K = K; p.alpha = alpha; p.act = act; p.round_first = round_first;
    p.st_in = ln_stats_in; p.parts_in = ln_parts_in; p.ln_self = ln_self; p.csum = ln_csum; p.eps = ln_eps; p.inv_c = 1.f / (float)K;
    p.st_out = ln_stats_out;
    p.mix_maa = (const pafc::bf16_t *)mix_maa; p.mix_prev = (const pafc::bf16_t *)mix_prev; p.T = mix_T;
    p.nrm_g = (const pafc::bf16_t *)norm_gamma; p.nrm_b = (const pafc::bf16_t *)norm_beta; p.nrm_eps = norm_eps;
    hipStream_t s = (hipStream_t)stream;
    const long mt = (M + 15) / 16;
    const long cols = (long)(n_out / 16) * batch;
    const bool k8 = K >= 2048;                     // eight waves split a long K
    // Row tiles per block: all of them (<= 8: one group, the weights are read once) when the column blocks alone occupy a good
    // part of the chip; else one tile per block, so that ~100+ blocks share the work (the weight slices are re-read from L2)
    if (norm_gamma) {
        if (k8) pafc::launch_sk<1, false, 8, 2>(p, batch, s);
        else pafc::launch_sk<1, false, 4, 2>(p, batch, s);
    } else if (mix_maa) {
        if (k8) pafc::launch_sk<1, false, 8, 1>(p, batch, s);
        else pafc::launch_sk<1, false, 4, 1>(p, batch, s);
    } else if (glu) {
        if (cols < 96) { if (k8) pafc::launch_sk<1, true, 8, 0>(p, batch, s); else pafc::launch_sk<1, true, 4, 0>(p, batch, s); }
        else if (mt <= 4) pafc::launch_sk<4, true, 4, 0>(p, batch, s);
        else pafc::launch_sk<5, true, 4, 0>(p, batch, s);
    } else {
        if (cols < 96) { if (k8) pafc::launch_sk<1, false, 8, 0>(p, batch, s); else pafc::launch_sk<1, false, 4, 0>(p, batch, s); }
        else if (mt <= 4) { if (k8) pafc::launch_sk<4, false, 8, 0>(p, batch, s); else pafc::launch_sk<4, false, 4, 0>(p, batch, s); }
        else if (mt <= 5) pafc::launch_sk<5, false, 4, 0>(p, batch, s);
        else pafc::launch_sk<8, false, 4, 0>(p, batch, s);
    }
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_decay_lora_skinny_bf16(long M, int C, int H, const void *x, long ldx, const void *d1n, const void *d2n,
                                           const void *bias, void *out, long ldo, pafc_stream_t stream) {
    if (!x || !d1n || !d2n || !out) return PAFC_ERR_NULL_POINTER;
    if (M <= 0 || C <= 0) return PAFC_ERR_BAD_DIMS;
    if (H != 64 || C % 32 != 0 || C % 16 != 0 || ldx % 8 || ldx < C || ldo % 4 || ldo < C) return PAFC_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)d1n | (uintptr_t)d2n) & 15 || ((uintptr_t)out & 7) || ((uintptr_t)bias & 7)) return PAFC_ERR_UNSUPPORTED;
    pafc::SkParams p{};
    p.A = nullptr; p.W = (const pafc::bf16_t *)d2n; p.bias = (const pafc::bf16_t *)bias; p.out = (pafc::bf16_t *)out;
    p.M = M; p.lda = H; p.ldw = H; p.ldo = ldo; p.N = C; p.K = H; p.alpha = 1.f; p.act = 0; p.round_first = bias != nullptr;
    p.inv_c = 1.f / (float)H;
    p.lora_x = (const pafc::bf16_t *)x; p.lora_w1 = (const pafc::bf16_t *)d1n; p.ldx = ldx; p.K1 = C;
    pafc::launch_sk<1, false, 4, 3>(p, 1, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_gemm_skinny_bf16(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W, long ldw,
                                     long strideW, const void *bias, long strideBias, const void *residual, long ldr, long strideR,
                                     void *out, long ldo, long strideO, float alpha, int act, const float *ln_stats_in,
                                     int ln_parts_in, const float *ln_csum, float ln_eps, float *ln_stats_out,
                                     pafc_stream_t stream) {
    return pafc_gemm_skinny_bf16_ex(M, N, K, batch, A, lda, strideA, W, ldw, strideW, bias, strideBias, residual, ldr, strideR, out,
                                    ldo, strideO, alpha, act, 0, ln_stats_in, ln_parts_in, 0, ln_csum, ln_eps, ln_stats_out, nullptr,
                                    nullptr, 0, nullptr, nullptr, 0.f, stream);
}
