// HBM-bound glue kernels of the Conformer layer and the RWKV time-mix for gfx950.
// C ABI: include/pafc_encoder_ops.h.  In the reference every one of these is a chain of separate PyTorch
// element-wise kernels, each a full (B,T,C) round trip; here each chain is one pass.  bf16 rounding points
// are those of the op-by-op PyTorch chain (every intermediate that PyTorch would materialise in bf16 is
// rounded to bf16 in registers), so the fused result tracks the reference's numerics, not a re-association.
#include <stdlib.h>

#include <type_traits>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {

struct SplitBf16 {};          // output form PAFC_SPLIT_BF16 (see store8_split below)
template <> struct Elem<SplitBf16> {
    __device__ static __forceinline__ float round(float v) { return v; }
};

namespace {

constexpr int VEC = 8;       // channels per lane per iteration
constexpr int MAXIT = 2;     // => C <= 64 * 8 * 2 = 1024

template <typename ET> __device__ __forceinline__ void load8(const ET *p, float *f);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t *p, float *f) {
    Elem<bf16_t>::unpack(*reinterpret_cast<const uint4 *>(p), f);
}
template <> __device__ __forceinline__ void load8<float>(const float *p, float *f) {
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename ET> __device__ __forceinline__ void store8(ET *p, const float *f);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t *p, const float *f) {
    uint4 q;
    q.x = f32_to_bf16_bits(f[0]) | (f32_to_bf16_bits(f[1]) << 16);
    q.y = f32_to_bf16_bits(f[2]) | (f32_to_bf16_bits(f[3]) << 16);
    q.z = f32_to_bf16_bits(f[4]) | (f32_to_bf16_bits(f[5]) << 16);
    q.w = f32_to_bf16_bits(f[6]) | (f32_to_bf16_bits(f[7]) << 16);
    *reinterpret_cast<uint4 *>(p) = q;
}
template <> __device__ __forceinline__ void store8<float>(float *p, const float *f) {
    reinterpret_cast<float4 *>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4 *>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}

// An fp32 value as two bf16 planes, hi = bf16(x), lo = bf16(x - hi): 16 significant bits, the form the phase-pipelined GEMM
// takes as the A operand of an fp32 model (csrc/gemm_ph.hip).  Output "dtype" PAFC_SPLIT_BF16: a row is [hi (C) | lo (C)].
__device__ __forceinline__ void store8_split(bf16_t *hi_p, bf16_t *lo_p, const float *f) {
    float lo[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) lo[e] = f[e] - round_bf16(f[e]);
    store8<bf16_t>(hi_p, f);
    store8<bf16_t>(lo_p, lo);
}
// row `row` of an LN output: ld elements apart, 8 values from column c
template <typename EO>
__device__ __forceinline__ void store_out(void *base, size_t row, long ld, int c, int C, const float *f) {
    if constexpr (std::is_same<EO, SplitBf16>::value) store8_split((bf16_t *)base + row * ld + c, (bf16_t *)base + row * ld + C + c, f);
    else store8<EO>((EO *)base + row * ld + c, f);
}

__device__ __forceinline__ float silu_(float x) { return x / (1.f + __expf(-x)); }

struct LnArgs {
    const void *x;        // (rows, C) residual stream
    const void *y;        // (rows, C) branch output to add, or null
    float alpha;          // x_new = x + alpha * y
    const int32_t *lens;  // (B) valid frames per batch item, or null
    int T;                // rows per batch item (for lens)
    int mask_y;           // zero y rows with t >= lens[b] before the add (convolution.py:140-141)
    void *x_out;          // (rows, C) x_new, or null
    const void *g1, *b1;  // LayerNorm 1 (C)
    void *o1;             // LN1 output, row stride ld1 (elements), or null
    long ld1;
    int silu1;            // o1 = silu(LN1(.))
    int zero1;            // zero o1 rows with t >= lens[b] (the masked_fill on the conv-module input, convolution.py:109-110)
    const void *g2, *b2;  // optional LayerNorm 2 applied to the (rounded) LN1 output
    void *o2;
    long ld2;
    int rows, C;
    float eps;
    int split2;           // EO = float only: o2 is written as bf16 planes [hi | lo] (o1 stays fp32)
    // row statistics for a LayerNorm folded into the GEMM that follows (csrc/gemm_ph.hip, LNF): float2 [rows][8], pair 0 =
    // (sum, sum of squares) of the row, pairs 1..7 zero -- of x_new (stats_x) and / or of the LN1 output as stored (stats_o1)
    float *stats_x, *stats_o1;
};

__device__ __forceinline__ void write_row_stats(float *stats, size_t row, int lane, float s1, float s2) {
    if (lane < 8) reinterpret_cast<float2 *>(stats)[row * 8 + lane] = lane == 0 ? make_float2(s1, s2) : make_float2(0.f, 0.f);
}

// One wave per row.  EX: residual / parameter dtype; EO: dtype of the LayerNorm outputs.
template <typename EX, typename EO>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int C = a.C;
    const EX *x = (const EX *)a.x + (size_t)row * C;
    bool beyond = false;
    if (a.lens) beyond = (row % a.T) >= a.lens[row / a.T];

    float v[MAXIT][VEC];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < C) {
            load8<EX>(x + c, v[it]);
            if (a.y) {
                float yv[VEC];
                load8<EX>((const EX *)a.y + (size_t)row * C + c, yv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float t = (a.mask_y && beyond) ? 0.f : yv[e];
                    if (a.alpha != 1.f) t = Elem<EX>::round(a.alpha * t);
                    v[it][e] = Elem<EX>::round(v[it][e] + t);
                }
                if (a.x_out) store8<EX>((EX *)a.x_out + (size_t)row * C + c, v[it]);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum += v[it][e];
        }
    }
    if (a.stats_x) {
        float sq0 = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
            if ((it * 64 + lane) * VEC < C) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sq0 = fmaf(v[it][e], v[it][e], sq0);
            }
        write_row_stats(a.stats_x, (size_t)row, lane, wave_sum(sum), wave_sum(sq0));
    }
    if (!a.o1) return;
    const float inv_c = 1.f / (float)C;
    float mean = wave_sum(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v[it][e] - mean; sq = fmaf(d, d, sq); }
        }
    }
    float rstd = rsqrtf(wave_sum(sq) * inv_c + a.eps);
    float sum2 = 0.f, sq2 = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < C) {
            float g[VEC], b[VEC];
            load8<EX>((const EX *)a.g1 + c, g);
            load8<EX>((const EX *)a.b1 + c, b);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float o = Elem<EO>::round(fmaf((v[it][e] - mean) * rstd, g[e], b[e]));
                if (a.silu1) o = Elem<EO>::round(silu_(o));
                if (a.zero1 && beyond) o = 0.f;
                v[it][e] = o;
                sum2 += o;
                sq2 = fmaf(o, o, sq2);
            }
            store_out<EO>(a.o1, (size_t)row, a.ld1, c, C, v[it]);
        }
    }
    if (a.stats_o1) write_row_stats(a.stats_o1, (size_t)row, lane, wave_sum(sum2), wave_sum(sq2));
    if (!a.o2) return;
    mean = wave_sum(sum2) * inv_c;
    sq = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v[it][e] - mean; sq = fmaf(d, d, sq); }
        }
    }
    rstd = rsqrtf(wave_sum(sq) * inv_c + a.eps);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < C) {
            float g[VEC], b[VEC], o[VEC];
            load8<EX>((const EX *)a.g2 + c, g);
            load8<EX>((const EX *)a.b2 + c, b);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = fmaf((v[it][e] - mean) * rstd, g[e], b[e]);
            if (std::is_same<EO, float>::value && a.split2) store_out<SplitBf16>(a.o2, (size_t)row, a.ld2, c, C, o);
            else store_out<EO>(a.o2, (size_t)row, a.ld2, c, C, o);
        }
    }
}

// ---- RWKV time-mix glue -----------------------------------------------------------------------------------------
// token shift + first lerp, both directions from one read of x (src/model.py:274-276):
//   xx_d = shift_d(x) - x;  xxx_d = x + xx_d * maa_x_d        d = 0: x_{t-1} (zero at t = 0), d = 1: x_{t+1} (zero at T-1)
template <typename ET>
__global__ __launch_bounds__(256) void tmix_shift_mix_kernel(int T, int C, long rows, int ndir, int rev0,
                                                             const ET *__restrict__ x, const ET *__restrict__ maa0,
                                                             const ET *__restrict__ maa1, ET *__restrict__ out,
                                                             const ET *__restrict__ prev) {
    // prev: (B, C) or null -- the frame before each sequence's first one (streaming: the previous chunk's last frame)
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int cpr = C / VEC;  // lanes per row
    const long row = gid / cpr;
    if (row >= rows) return;
    const int c = (int)(gid % cpr) * VEC;
    const int t = (int)(row % T);
    float xc[VEC], xp[VEC], xn[VEC];
    load8<ET>(x + row * C + c, xc);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { xp[e] = 0.f; xn[e] = 0.f; }
    if (t > 0) load8<ET>(x + (row - 1) * C + c, xp);
    else if (prev) load8<ET>(prev + (row / T) * C + c, xp);
    if (t < T - 1) load8<ET>(x + (row + 1) * C + c, xn);
    for (int d = 0; d < ndir; ++d) {
        const bool rev = (d == 0) ? (rev0 != 0) : true;
        float m[VEC], o[VEC];
        load8<ET>((d == 0 ? maa0 : maa1) + c, m);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xx = Elem<ET>::round((rev ? xn[e] : xp[e]) - xc[e]);
            o[e] = Elem<ET>::round(xc[e] + Elem<ET>::round(xx * m[e]));
        }
        store8<ET>(out + ((size_t)d * rows + row) * C + c, o);
    }
}

// the four data-dependent lerps (src/model.py:280-284):  z_q = x + xx * (maa_q + m_q),  q in {r, k, v, w}
//   m:   [ndir][4][rows][C]   (LoRA outputs, per direction)
//   maa: [ndir][4][C]
//   z:   [4][ndir][rows][C]   (q-major so that r,k,v of all directions are one contiguous batch of GEMM inputs)
template <typename ET>
__global__ __launch_bounds__(256) void tmix_mix4_kernel(int T, int C, long rows, int ndir, int rev0,
                                                        const ET *__restrict__ x, const ET *__restrict__ m,
                                                        const ET *__restrict__ maa, ET *__restrict__ z) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int cpr = C / VEC;
    const long row = gid / cpr;
    if (row >= rows) return;
    const int c = (int)(gid % cpr) * VEC;
    const int t = (int)(row % T);
    float xc[VEC], xp[VEC], xn[VEC];
    load8<ET>(x + row * C + c, xc);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { xp[e] = 0.f; xn[e] = 0.f; }
    if (t > 0) load8<ET>(x + (row - 1) * C + c, xp);
    if (t < T - 1) load8<ET>(x + (row + 1) * C + c, xn);
    for (int d = 0; d < ndir; ++d) {
        const bool rev = (d == 0) ? (rev0 != 0) : true;
        float xx[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) xx[e] = Elem<ET>::round((rev ? xn[e] : xp[e]) - xc[e]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float mv[VEC], av[VEC], o[VEC];
            load8<ET>(m + (((size_t)d * 4 + q) * rows + row) * C + c, mv);
            load8<ET>(maa + ((size_t)d * 4 + q) * C + c, av);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                o[e] = Elem<ET>::round(xc[e] + Elem<ET>::round(xx[e] * Elem<ET>::round(av[e] + mv[e])));
            store8<ET>(z + (((size_t)q * ndir + d) * rows + row) * C + c, o);
        }
    }
}

// LoRA up-projection + the four lerps in one pass (bf16):
//   m_q = bf16( tanh(xxx W1)[:, 32q:32q+32] . W2[q] )   (src/model.py:277-278, a K = 32 batched GEMM in the reference)
//   z_q = x + xx * (maa_q + m_q)                         (src/model.py:280-284)
// The K = 32 product is exactly one v_mfma_f32_16x16x32_bf16 per 16x16 tile, so the (rows, C) x 4 x ndir LoRA maps
// are never written to or read from HBM (368 MB per layer at the 30-minute shape).  One wave = 16 time rows x all
// columns; the MFMA is issued with M = output column, N = time row, so each lane ends up with 4 consecutive columns
// of ITS row per MFMA and, with the column slots of two MFMAs interleaved, 8 = one 16-byte store.
//   x: (rows, C); t: (ndir, rows, 128); w2t: (ndir, 4, C, 32) (W2 with K innermost); maa: (ndir, 4, C); z: (4, ndir, rows, C)
typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
template <bool LDSW, bool FULLROW = false>
__global__ __launch_bounds__(256) void tmix_lora_mix4_kernel(int T, int C, long rows, int ndir, int rev0,
                                                             const bf16_t *__restrict__ x, const bf16_t *__restrict__ t,
                                                             const bf16_t *__restrict__ w2t,
                                                             const bf16_t *__restrict__ maa, bf16_t *__restrict__ z,
                                                             const bf16_t *__restrict__ prev) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, qq = lane >> 4;
    const long row = ((long)blockIdx.x * 4 + wave) * 16 + r16;
    const long rowc = row < rows ? row : rows - 1;     // clamp (every lane takes part in the MFMAs), skip the store
    const int tt = (int)(rowc % T);
    const f32x4g zero = {0.f, 0.f, 0.f, 0.f};
    // the MFMA leaves a lane with 8 columns of ITS row; through LDS the tile goes back to whole 128-byte row
    // segments per 8 lanes, so z is stored in full lines
    constexpr int LDZ = 64 + 8;
    __shared__ __attribute__((aligned(16))) bf16_t s_z[4][16][LDZ];   // [wave][row][col]: one map's tile at a time
    // the block's slice of W2 (this block's 64 columns x 32 LoRA inputs, 4 maps, both directions = 32 KiB) goes through LDS
    // once: its four waves (64 rows) would otherwise each fetch the same 16 KiB per direction from L2 (720 MB per layer at the
    // 30-minute shape, more than the kernel's HBM traffic)
    __shared__ __attribute__((aligned(16))) bf16_t s_w2[LDSW ? 2 : 1][LDSW ? 4 : 1][LDSW ? 64 : 1][32];
    if constexpr (LDSW) {
        for (int i = threadIdx.x; i < ndir * 4 * 64 * 4; i += 256) {          // 16-byte chunks: [d][q][col][4 chunks of 8]
            const int ck = i & 3, col = (i >> 2) & 63, dq = i >> 8;           // dq = d * 4 + q
            *reinterpret_cast<uint4 *>(&s_w2[0][0][0][0] + ((size_t)dq * 64 + col) * 32 + 8 * ck) =
                *reinterpret_cast<const uint4 *>(w2t + ((size_t)dq * C + blockIdx.y * 64 + col) * 32 + 8 * ck);
        }
        __syncthreads();
    }
    for (int d = 0; d < ndir; ++d) {
        const bool rev = (d == 0) ? (rev0 != 0) : true;
        const bool in_seq = rev ? (tt < T - 1) : (tt > 0);
        const bool has_nb = in_seq || (prev != nullptr && !rev);       // prev: (B, C) the frame before each sequence, or null
        const bf16_t *nbrow = in_seq ? x + (rev ? rowc + 1 : rowc - 1) * C : (has_nb ? prev + (rowc / T) * C : x + rowc * C);
        uint4 tb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            tb[q] = *reinterpret_cast<const uint4 *>(t + ((size_t)d * rows + rowc) * 128 + 32 * q + 8 * qq);
        // a wave covers 2 column blocks of 32 (blockIdx.y picks which 64 columns): x and the shifted difference of
        // both stay in registers across the four maps
        float xc[2][VEC], xx[2][VEC];
#pragma unroll
        for (int cbi = 0; cbi < 2; ++cbi) {
            const int col = (blockIdx.y * 2 + cbi) * 32 + 8 * qq;            // this lane's 8 output columns
            float xn[VEC];
            load8<bf16_t>(x + rowc * C + col, xc[cbi]);
            load8<bf16_t>(nbrow + col, xn);                                 // branch-free: select after the load
#pragma unroll
            for (int e = 0; e < VEC; ++e) xx[cbi][e] = round_bf16((has_nb ? xn[e] : 0.f) - xc[cbi][e]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int cbi = 0; cbi < 2; ++cbi) {
                const int cb = blockIdx.y * 2 + cbi;
                const int col = cb * 32 + 8 * qq;
                const int colA = cb * 32 + 8 * (r16 >> 2) + (r16 & 3);  // column whose W2 row this lane feeds (slot r16)
                uint4 a1, a2;
                if constexpr (LDSW) {
                    const int colL = colA - blockIdx.y * 64;             // column inside this block's 64
                    a1 = *reinterpret_cast<const uint4 *>(&s_w2[d][q][colL][8 * qq]);
                    a2 = *reinterpret_cast<const uint4 *>(&s_w2[d][q][colL + 4][8 * qq]);
                } else {
                    const bf16_t *wq = w2t + ((size_t)(d * 4 + q) * C) * 32 + 8 * qq;
                    a1 = *reinterpret_cast<const uint4 *>(wq + (size_t)colA * 32);
                    a2 = *reinterpret_cast<const uint4 *>(wq + (size_t)(colA + 4) * 32);
                }
                const f32x4g m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8g, a1),
                                                                         __builtin_bit_cast(bf16x8g, tb[q]), zero, 0, 0, 0);
                const f32x4g m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8g, a2),
                                                                         __builtin_bit_cast(bf16x8g, tb[q]), zero, 0, 0, 0);
                float av[VEC], o[VEC];
                load8<bf16_t>(maa + ((size_t)d * 4 + q) * C + col, av);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float mv = round_bf16(e < 4 ? m1[e] : m2[e - 4]);
                    o[e] = round_bf16(xc[cbi][e] + round_bf16(xx[cbi][e] * round_bf16(av[e] + mv)));
                }
                store8<bf16_t>(&s_z[wave][r16][cbi * 32 + 8 * qq], o);
            }
            // s_z[wave] is private to this wave and LDS operations of one wave complete in order: a compiler fence is
            // all the synchronisation the re-layout needs (no block barrier: the four waves run independently)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int rr = half * 8 + (lane >> 3), cc = (lane & 7) * 8;
                const long orow = ((long)blockIdx.x * 4 + wave) * 16 + rr;
                if (orow < rows)
                    *reinterpret_cast<uint4 *>(z + (((size_t)q * ndir + d) * rows + orow) * C + blockIdx.y * 64 + cc) =
                        *reinterpret_cast<const uint4 *>(&s_z[wave][rr][cc]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}


// Weight-stationary form of tmix_lora_mix4_kernel: a block keeps ONE 64-column slice of W2 and maa of ONE direction in
// registers (16 MFMA A fragments + the lerp coefficients) and walks over row tiles with it, so the weights are read from L2
// once per block instead of once per 16 rows (720 MB of fragment reads per layer at the 30-minute shape otherwise).
// grid = (G, C / 64, ndir); a wave takes the 16-row tiles blockIdx.x * 4 + wave, + 4 G, ...
template <int NW, int WCOLS>   // waves per block; how many of them sit side by side on one row tile (adjacent 64-column slices)
__global__ __launch_bounds__(NW * 64) void tmix_lora_mix4_ws_kernel(int T, int C, long rows, int ndir, int rev0,
                                                                const bf16_t *__restrict__ x, const bf16_t *__restrict__ t,
                                                                const bf16_t *__restrict__ w2t,
                                                                const bf16_t *__restrict__ maa, bf16_t *__restrict__ z,
                                                                const bf16_t *__restrict__ prev) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, qq = lane >> 4;
    constexpr int WROWS = NW / WCOLS;
    const int cy = blockIdx.y * WCOLS + wave % WCOLS, d = blockIdx.z;
    const int wrow = wave / WCOLS;
    const bool rev = (d == 0) ? (rev0 != 0) : true;
    const f32x4g zero = {0.f, 0.f, 0.f, 0.f};
    constexpr int LDZ = 64 + 8;
    __shared__ __attribute__((aligned(16))) bf16_t s_z[NW][16][LDZ];
    const int colslot = 8 * (r16 >> 2) + (r16 & 3);
    uint4 a1[4][2], a2[4][2], av[4][2];       // [map][32-column block]: W2 rows (MFMA A operands), maa of my 8 columns (packed)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int cbi = 0; cbi < 2; ++cbi) {
            const int cb = cy * 2 + cbi;
            const bf16_t *wq = w2t + ((size_t)(d * 4 + q) * C + cb * 32 + colslot) * 32 + 8 * qq;
            a1[q][cbi] = *reinterpret_cast<const uint4 *>(wq);
            a2[q][cbi] = *reinterpret_cast<const uint4 *>(wq + 4 * 32);
            av[q][cbi] = *reinterpret_cast<const uint4 *>(maa + ((size_t)d * 4 + q) * C + cb * 32 + 8 * qq);
        }
    const long ntiles = (rows + 15) / 16;
    for (long tile = (long)blockIdx.x * WROWS + wrow; tile < ntiles; tile += (long)gridDim.x * WROWS) {
        const long row = tile * 16 + r16;
        const long rowc = row < rows ? row : rows - 1;     // clamp (every lane takes part in the MFMAs), skip the store
        const int tt = (int)(rowc % T);
        const bool in_seq = rev ? (tt < T - 1) : (tt > 0);
        const bool has_nb = in_seq || (prev != nullptr && !rev);       // prev: (B, C) the frame before each sequence, or null
        const bf16_t *nbrow = in_seq ? x + (rev ? rowc + 1 : rowc - 1) * C : (has_nb ? prev + (rowc / T) * C : x + rowc * C);
        uint4 tb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            tb[q] = *reinterpret_cast<const uint4 *>(t + ((size_t)d * rows + rowc) * 128 + 32 * q + 8 * qq);
        float xc[2][VEC], xx[2][VEC];
#pragma unroll
        for (int cbi = 0; cbi < 2; ++cbi) {
            const int col = (cy * 2 + cbi) * 32 + 8 * qq;
            float xn[VEC];
            load8<bf16_t>(x + rowc * C + col, xc[cbi]);
            load8<bf16_t>(nbrow + col, xn);
#pragma unroll
            for (int e = 0; e < VEC; ++e) xx[cbi][e] = round_bf16((has_nb ? xn[e] : 0.f) - xc[cbi][e]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int cbi = 0; cbi < 2; ++cbi) {
                const f32x4g m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8g, a1[q][cbi]),
                                                                         __builtin_bit_cast(bf16x8g, tb[q]), zero, 0, 0, 0);
                const f32x4g m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8g, a2[q][cbi]),
                                                                         __builtin_bit_cast(bf16x8g, tb[q]), zero, 0, 0, 0);
                const unsigned aw[4] = {av[q][cbi].x, av[q][cbi].y, av[q][cbi].z, av[q][cbi].w};
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float a = (e & 1) ? __uint_as_float(aw[e >> 1] & 0xffff0000u) : __uint_as_float(aw[e >> 1] << 16);
                    const float mv = round_bf16(e < 4 ? m1[e] : m2[e - 4]);
                    o[e] = round_bf16(xc[cbi][e] + round_bf16(xx[cbi][e] * round_bf16(a + mv)));
                }
                store8<bf16_t>(&s_z[wave][r16][cbi * 32 + 8 * qq], o);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int rr = half * 8 + (lane >> 3), cc = (lane & 7) * 8;
                const long orow = tile * 16 + rr;
                if (orow < rows)
                    *reinterpret_cast<uint4 *>(z + (((size_t)q * ndir + d) * rows + orow) * C + cy * 64 + cc) =
                        *reinterpret_cast<const uint4 *>(&s_z[wave][rr][cc]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The decay LoRA in one pass (bf16):  w = bf16( bf16(tanh(z_w D1)) . D2 ) [+ time_decay]   (src/model.py:286-287)
// z_w: (ndir, rows, C) the fourth lerp; d1n: (ndir, 64, C) = time_decay_w1^T; d2n: (ndir, C, 64) = time_decay_w2^T.
// Both weight matrices of a direction (2 x 64 KiB) stay in LDS in MFMA-fragment order (one lane-linear ds_read_b128 per
// fragment, conflict-free); a wave takes 16 rows at a time: the transposed product td^T[n1][row] = sum_c D1^T[n1][c]
// z_w^T[c][row] takes the row's own 16-byte pieces as its B operand straight from global memory, the accumulators (lane = row,
// register = 4 consecutive n1) become after tanh + rounding the B operand of w^T[c][row] = sum_n1 D2^T[c][n1] td^T[n1][row]
// with the k-slots of a step taken as (4 of tile 2p | 4 of tile 2p+1), and w leaves through a per-wave LDS tile in full
// lines.  The 64-wide hidden tensor never exists in memory and z_w / w cross HBM once each (two library-shaped GEMMs: 83 us
// for 206 MB at the 30-minute shape).
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float tanh_as_gemm_epilogue(float v) {   // the form gemm_bf16.hip / gemm_ph.hip apply (act 2)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
}
__device__ __forceinline__ unsigned pack_bf16_rne(float lo, float hi) {     // round to nearest even and pack: one v_cvt_pk_bf16_f32
    typedef float f32x2r __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2r __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2r{lo, hi}, bf16x2r));
}
__device__ __forceinline__ unsigned pack_bf16_exact(float lo, float hi) {   // both already bf16 values
    return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}
constexpr int DL_C = 512, DL_H = 64;      // model width and LoRA width this kernel is built for (checked on the host)
constexpr int DL_WAVES = 8;
__global__ __launch_bounds__(DL_WAVES * 64) void decay_lora_kernel(long rows, int ndir, const bf16_t *__restrict__ zw,
                                                                   const bf16_t *__restrict__ d1n,
                                                                   const bf16_t *__restrict__ d2n,
                                                                   const bf16_t *__restrict__ bias, bf16_t *__restrict__ wout) {
    constexpr int C = DL_C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, qq = lane >> 4;
    const f32x4g zero = {0.f, 0.f, 0.f, 0.f};
    constexpr int LDZ = 64 + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char dl_lds[];
    uint4 *s_d1 = reinterpret_cast<uint4 *>(dl_lds);                       // [ks 16][tau 4][lane 64]   64 KiB
    uint4 *s_d2 = s_d1 + 16 * 4 * 64;                                      // [cb 16][a 2][p 2][lane 64] 64 KiB
    bf16_t(*s_z)[16][LDZ] = reinterpret_cast<bf16_t(*)[16][LDZ]>(s_d2 + 16 * 2 * 2 * 64);   // [wave][row][col]
    const int colslot = 8 * (r16 >> 2) + (r16 & 3);   // column (in a 32-block) whose weight row this lane feeds as MFMA row r16
    const long ntiles = (rows + 15) / 16;
    {
        const int d = blockIdx.y;                       // a block serves one direction: its weights are staged once
        for (int f = wave; f < 64; f += DL_WAVES) {     // fragment (ks, tau): lane (r16, qq) <- D1^T[16 tau + r16][32 ks + 8 qq ..]
            const int ks = f >> 2, tau = f & 3;
            s_d1[f * 64 + lane] = *reinterpret_cast<const uint4 *>(d1n + ((size_t)d * DL_H + 16 * tau + r16) * C + 32 * ks + 8 * qq);
        }
        for (int f = wave; f < 64; f += DL_WAVES) {     // fragment (cb, a, p): the row of column cb*32 + colslot + 4a, k-slots of step p
            const int cb = f >> 2, a = (f >> 1) & 1, p = f & 1;
            const bf16_t *dq = d2n + ((size_t)d * C + cb * 32 + colslot + 4 * a) * DL_H + 32 * p + 4 * qq;
            const uint2 lo = *reinterpret_cast<const uint2 *>(dq), hi = *reinterpret_cast<const uint2 *>(dq + 16);
            s_d2[f * 64 + lane] = uint4{lo.x, lo.y, hi.x, hi.y};
        }
        __syncthreads();
        for (long tile = (long)blockIdx.x * DL_WAVES + wave; tile < ntiles; tile += (long)gridDim.x * DL_WAVES) {
            const long row = tile * 16 + r16;
            const long rowc = row < rows ? row : rows - 1;     // clamp (every lane takes part in the MFMAs), skip the store
            const bf16_t *zr = zw + ((size_t)d * rows + rowc) * C + 8 * qq;
            uint4 zf[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) zf[ks] = *reinterpret_cast<const uint4 *>(zr + 32 * ks);
            f32x4g acc1[4] = {zero, zero, zero, zero};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int tau = 0; tau < 4; ++tau)
                    acc1[tau] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8g, s_d1[(ks * 4 + tau) * 64 + lane]),
                                                                         __builtin_bit_cast(bf16x8g, zf[ks]), acc1[tau], 0, 0, 0);
            // tanh + rounding; k-step p of the second product takes n1 = 32 p + 4 qq + e (e < 4) | 32 p + 16 + 4 qq + e - 4
            u32x4g b2[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = round_bf16(tanh_as_gemm_epilogue(acc1[2 * p + (e >> 2)][e & 3]));
                b2[p] = u32x4g{pack_bf16_exact(v[0], v[1]), pack_bf16_exact(v[2], v[3]),
                               pack_bf16_exact(v[4], v[5]), pack_bf16_exact(v[6], v[7])};
            }
            for (int cy = 0; cy < C / 64; ++cy) {
#pragma unroll
                for (int cbi = 0; cbi < 2; ++cbi) {
                    const int cb = cy * 2 + cbi;
                    f32x4g o[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        o[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8g, s_d2[((cb * 2 + a) * 2 + 0) * 64 + lane]),
                                                                       __builtin_bit_cast(bf16x8g, b2[0]), zero, 0, 0, 0);
                        o[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8g, s_d2[((cb * 2 + a) * 2 + 1) * 64 + lane]),
                                                                       __builtin_bit_cast(bf16x8g, b2[1]), o[a], 0, 0, 0);
                    }
                    float ov[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ov[e] = e < 4 ? o[0][e] : o[1][e - 4];
                    if (bias) {   // w = time_decay + ww, each rounded where the reference's op chain rounds it
                        float bv[VEC];
                        load8<bf16_t>(bias + (size_t)d * C + cb * 32 + 8 * qq, bv);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) ov[e] = bv[e] + round_bf16(ov[e]);
                    }
                    store8<bf16_t>(&s_z[wave][r16][cbi * 32 + 8 * qq], ov);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int rr = half * 8 + (lane >> 3), cc = (lane & 7) * 8;
                    const long orow = tile * 16 + rr;
                    if (orow < rows)
                        *reinterpret_cast<uint4 *>(wout + ((size_t)d * rows + orow) * C + cy * 64 + cc) =
                            *reinterpret_cast<const uint4 *>(&s_z[wave][rr][cc]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// Token shift + the first lerp + the LoRA down-projection + tanh in one pass (bf16):
//   xx = x_neighbour - x;  xxx = x + xx * maa_x;  t = bf16(tanh(xxx W1))            (src/model.py:273-277)
// x: (rows, C); maa_x: (ndir, C); w1n: (ndir, 128, C) = time_maa_rkvw_w1^T (K innermost); t: (ndir, rows, 128).
// W1 of a direction (128 KiB) stays in LDS in MFMA-fragment order; a wave takes 16 rows at a time and forms its B operand --
// 8 consecutive columns of ITS row of xxx per K-step -- in registers from x and the neighbouring row, so xxx (2 x 46 MB
// written and read back at the 30-minute shape) never exists in memory.  Transposed product t^T[n][row]; the weight rows of
// a pair of MFMAs are interleaved so that a lane ends up with 8 consecutive n of its row = one 16-byte store.
constexpr int LD_C = 512, LD_N = 128;
template <int NW>
__global__ __launch_bounds__(NW * 64) void tmix_lora_down_kernel(int T, long rows, int rev0, const bf16_t *__restrict__ x,
                                                                       const bf16_t *__restrict__ maa_x,
                                                                       const bf16_t *__restrict__ w1n, bf16_t *__restrict__ tout,
                                                                       const bf16_t *__restrict__ prev) {
    constexpr int C = LD_C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, qq = lane >> 4;
    const f32x4g zero = {0.f, 0.f, 0.f, 0.f};
    extern __shared__ __attribute__((aligned(16))) unsigned char dl_lds[];
    uint4 *s_w1 = reinterpret_cast<uint4 *>(dl_lds);                       // [ks 16][np 4][a 2][lane 64]   128 KiB
    const int colslot = 8 * (r16 >> 2) + (r16 & 3);
    const int d = blockIdx.y;                           // a block serves one direction: its weights are staged once
    const bool rev = (d == 0) ? (rev0 != 0) : true;
    for (int f = wave; f < 128; f += NW) {        // fragment (ks, np, a): lane (r16, qq) <- W1^T[32 np + colslot + 4 a][32 ks + 8 qq ..]
        const int ks = f >> 3, np = (f >> 1) & 3, a = f & 1;
        s_w1[f * 64 + lane] = *reinterpret_cast<const uint4 *>(w1n + ((size_t)d * LD_N + 32 * np + colslot + 4 * a) * C + 32 * ks + 8 * qq);
    }
    __syncthreads();
    const long ntiles = (rows + 15) / 16;
    for (long tile = (long)blockIdx.x * NW + wave; tile < ntiles; tile += (long)gridDim.x * NW) {
        const long row = tile * 16 + r16;
        const long rowc = row < rows ? row : rows - 1;     // clamp (every lane takes part in the MFMAs), skip the store
        const int tt = (int)(rowc % T);
        const bool in_seq = rev ? (tt < T - 1) : (tt > 0);
        const bool has_nb = in_seq || (prev != nullptr && !rev);       // prev: (B, C) the frame before each sequence, or null
        const bf16_t *nbrow = in_seq ? x + (rev ? rowc + 1 : rowc - 1) * C : (has_nb ? prev + (rowc / T) * C : x + rowc * C);
        const bf16_t *xr = x + rowc * C + 8 * qq, *xnr = nbrow + 8 * qq, *mr = maa_x + (size_t)d * C + 8 * qq;
        f32x4g acc[4][2];
#pragma unroll
        for (int np = 0; np < 4; ++np) { acc[np][0] = zero; acc[np][1] = zero; }
#pragma unroll 1
        for (int kc = 0; kc < 16; kc += 4)   // four K-steps at a time: their 12 loads are in flight together, no more
#pragma unroll
        for (int ks = kc; ks < kc + 4; ++ks) {
            float xc[VEC], xn[VEC], mm[VEC];
            load8<bf16_t>(xr + 32 * ks, xc);
            load8<bf16_t>(xnr + 32 * ks, xn);                // branch-free: select after the load
            load8<bf16_t>(mr + 32 * ks, mm);
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xx = round_bf16((has_nb ? xn[e] : 0.f) - xc[e]);
                o[e] = xc[e] + round_bf16(xx * mm[e]);              // (rounded by the packing conversion below: one v_cvt_pk per pair)
            }
            const u32x4g bq = {pack_bf16_rne(o[0], o[1]), pack_bf16_rne(o[2], o[3]), pack_bf16_rne(o[4], o[5]),
                               pack_bf16_rne(o[6], o[7])};
            // the K-step's products at raised wave priority: a wave that has formed its operand (115 VALU instructions per K-step)
            // gets the matrix pipe ahead of the waves still forming theirs (8 waves per block: 49.0 -> 43.5 us; 16: 47.6 -> 45.3)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int np = 0; np < 4; ++np)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[np][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8g, s_w1[((ks * 4 + np) * 2 + a) * 64 + lane]), __builtin_bit_cast(bf16x8g, bq),
                        acc[np][a], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (row < rows) {
#pragma unroll
            for (int np = 0; np < 4; ++np) {
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = tanh_as_gemm_epilogue(e < 4 ? acc[np][0][e] : acc[np][1][e - 4]);
                store8<bf16_t>(tout + ((size_t)d * rows + row) * LD_N + 32 * np + 8 * qq, o);
            }
        }
    }
}

// x (rows, cols) fp32, rows ldx apart -> out (rows, ldo) bf16 = [hi | lo] (lo at column lo_off), optionally scaled.
// `triple`: out = [hi | hi | lo] at columns 0, cols, 2 cols -- the weight of a split-operand GEMM (gemm_ph.hip).
__global__ __launch_bounds__(256) void split_planes_kernel(long rows, int cols, const float *__restrict__ x, long ldx,
                                                           bf16_t *__restrict__ out, long ldo, long lo_off, int triple) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int cpr = cols / VEC;
    const long row = gid / cpr;
    if (row >= rows) return;
    const int c = (int)(gid % cpr) * VEC;
    float f[VEC];
    load8<float>(x + row * ldx + c, f);
    if (triple) {
        store8<bf16_t>(out + row * ldo + c, f);
        store8_split(out + row * ldo + cols + c, out + row * ldo + 2 * cols + c, f);
    } else {
        store8_split(out + row * ldo + c, out + row * ldo + lo_off + c, f);
    }
}

template <typename EX>
int launch_ln(int dtype_out, const LnArgs &a, hipStream_t s) {
    dim3 grid((a.rows + 3) / 4), block(256);
    if (dtype_out == PAFC_BF16)
        hipLaunchKernelGGL((add_layernorm_kernel<EX, bf16_t>), grid, block, 0, s, a);
    else if (dtype_out == PAFC_SPLIT_BF16)
        hipLaunchKernelGGL((add_layernorm_kernel<EX, SplitBf16>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((add_layernorm_kernel<EX, float>), grid, block, 0, s, a);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // namespace
}  // namespace pafc

extern "C" {

int pafc_add_layernorm(int dtype, int dtype_out, int rows, int C, const void *x, const void *y, float alpha,
                       const int32_t *lens, int T, int mask_y, void *x_out, const void *gamma1, const void *beta1,
                       void *out1, long ld1, int silu1, int zero1, const void *gamma2, const void *beta2, void *out2,
                       long ld2, float eps, pafc_stream_t stream) {
    return pafc_add_layernorm_ex(dtype, dtype_out, dtype_out, rows, C, x, y, alpha, lens, T, mask_y, x_out, gamma1, beta1, out1, ld1,
                                 silu1, zero1, gamma2, beta2, out2, ld2, eps, nullptr, nullptr, stream);
}

int pafc_add_layernorm_ex(int dtype, int dtype_out, int dtype_out2, int rows, int C, const void *x, const void *y, float alpha,
                          const int32_t *lens, int T, int mask_y, void *x_out, const void *gamma1, const void *beta1,
                          void *out1, long ld1, int silu1, int zero1, const void *gamma2, const void *beta2, void *out2,
                          long ld2, float eps, float *stats_x, float *stats_out1, pafc_stream_t stream) {
    if (!x) return PAFC_ERR_NULL_POINTER;
    if (stats_out1 && !out1) return PAFC_ERR_NULL_POINTER;
    if (dtype_out2 != dtype_out && !(dtype_out == PAFC_F32 && dtype_out2 == PAFC_SPLIT_BF16)) return PAFC_ERR_DTYPE;
    if (out1 && (!gamma1 || !beta1)) return PAFC_ERR_NULL_POINTER;
    if (out2 && (!gamma2 || !beta2 || !out1)) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || C <= 0 || C % pafc::VEC || C > 64 * pafc::VEC * pafc::MAXIT || (lens && T <= 0))
        return PAFC_ERR_BAD_DIMS;
    if (ld1 % pafc::VEC || ld2 % pafc::VEC) return PAFC_ERR_BAD_DIMS;
    if (dtype == PAFC_BF16 && dtype_out != PAFC_BF16) return PAFC_ERR_DTYPE;
    if ((dtype_out == PAFC_SPLIT_BF16 && out1 && ld1 < 2 * C) || (dtype_out2 == PAFC_SPLIT_BF16 && out2 && ld2 < 2 * C))
        return PAFC_ERR_BAD_DIMS;
    pafc::LnArgs a{x, y, alpha, lens, T, mask_y, x_out, gamma1, beta1, out1, ld1, silu1, zero1,
                   gamma2, beta2, out2, ld2, rows, C, eps, dtype_out2 != dtype_out, stats_x, stats_out1};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_BF16) return pafc::launch_ln<pafc::bf16_t>(dtype_out, a, s);
    if (dtype == PAFC_F32) return pafc::launch_ln<float>(dtype_out, a, s);
    return PAFC_ERR_DTYPE;
}

int pafc_split_planes(long rows, int cols, const float *x, long ldx, void *out, long ldo, long lo_off, int triple,
                      pafc_stream_t stream) {
    if (!x || !out) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || cols <= 0 || cols % pafc::VEC || ldx < cols || ldx % 4) return PAFC_ERR_BAD_DIMS;
    if (triple ? ldo < 3L * cols : (lo_off < cols || ldo < lo_off + cols)) return PAFC_ERR_BAD_DIMS;
    if (ldo % pafc::VEC || lo_off % pafc::VEC || (((uintptr_t)x | (uintptr_t)out) & 15)) return PAFC_ERR_ALIGNMENT;
    const long threads = rows * (cols / pafc::VEC);
    hipLaunchKernelGGL(pafc::split_planes_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows,
                       cols, x, ldx, (pafc::bf16_t *)out, ldo, lo_off, triple);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

int pafc_tmix_shift_mix(int dtype, int B, int T, int C, int ndir, int reverse0, const void *x, const void *maa_x0,
                        const void *maa_x1, void *out, pafc_stream_t stream) {
    return pafc_tmix_shift_mix_prev(dtype, B, T, C, ndir, reverse0, x, maa_x0, maa_x1, nullptr, out, stream);
}

int pafc_tmix_shift_mix_prev(int dtype, int B, int T, int C, int ndir, int reverse0, const void *x, const void *maa_x0,
                             const void *maa_x1, const void *prev, void *out, pafc_stream_t stream) {
    if (!x || !maa_x0 || !out || (ndir == 2 && !maa_x1)) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || C <= 0 || C % pafc::VEC || ndir < 1 || ndir > 2) return PAFC_ERR_BAD_DIMS;
    const long rows = (long)B * T;
    const long threads = rows * (C / pafc::VEC);
    dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_BF16)
        hipLaunchKernelGGL(pafc::tmix_shift_mix_kernel<pafc::bf16_t>, grid, block, 0, s, T, C, rows, ndir, reverse0,
                           (const pafc::bf16_t *)x, (const pafc::bf16_t *)maa_x0, (const pafc::bf16_t *)maa_x1,
                           (pafc::bf16_t *)out, (const pafc::bf16_t *)prev);
    else if (dtype == PAFC_F32)
        hipLaunchKernelGGL(pafc::tmix_shift_mix_kernel<float>, grid, block, 0, s, T, C, rows, ndir, reverse0,
                           (const float *)x, (const float *)maa_x0, (const float *)maa_x1, (float *)out, (const float *)prev);
    else
        return PAFC_ERR_DTYPE;
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

int pafc_tmix_mix4(int dtype, int B, int T, int C, int ndir, int reverse0, const void *x, const void *m,
                   const void *maa, void *z, pafc_stream_t stream) {
    if (!x || !m || !maa || !z) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || C <= 0 || C % pafc::VEC || ndir < 1 || ndir > 2) return PAFC_ERR_BAD_DIMS;
    const long rows = (long)B * T;
    const long threads = rows * (C / pafc::VEC);
    dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_BF16)
        hipLaunchKernelGGL(pafc::tmix_mix4_kernel<pafc::bf16_t>, grid, block, 0, s, T, C, rows, ndir, reverse0,
                           (const pafc::bf16_t *)x, (const pafc::bf16_t *)m, (const pafc::bf16_t *)maa,
                           (pafc::bf16_t *)z);
    else if (dtype == PAFC_F32)
        hipLaunchKernelGGL(pafc::tmix_mix4_kernel<float>, grid, block, 0, s, T, C, rows, ndir, reverse0,
                           (const float *)x, (const float *)m, (const float *)maa, (float *)z);
    else
        return PAFC_ERR_DTYPE;
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}


int pafc_tmix_lora_mix4_bf16(int B, int T, int C, int ndir, int reverse0, const void *x, const void *t, const void *w2t,
                             const void *maa, void *z, pafc_stream_t stream) {
    return pafc_tmix_lora_mix4_bf16_prev(B, T, C, ndir, reverse0, x, t, w2t, maa, nullptr, z, stream);
}

int pafc_tmix_lora_mix4_bf16_prev(int B, int T, int C, int ndir, int reverse0, const void *x, const void *t, const void *w2t,
                                  const void *maa, const void *prev, void *z, pafc_stream_t stream) {
    if (!x || !t || !w2t || !maa || !z) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || C <= 0 || C % 64 || ndir < 1 || ndir > 2) return PAFC_ERR_BAD_DIMS;
    const long rows = (long)B * T;
    dim3 grid((unsigned)((rows + 63) / 64), C / 64), block(256);
    const char *e = getenv("PAFC_LORA_LDSW");      // A/B measurements: 1 = the block's W2 slice staged through LDS, 0 = reloaded per 16 rows
    const long ntiles = (rows + 15) / 16;
    if (!e && ntiles >= 256) {                      // weight-stationary: each wave walks over row tiles with its W2 slice
        const char *g = getenv("PAFC_LORA_WS_BLOCKS");
        const char *wc = getenv("PAFC_LORA_WS_WCOLS");
        const int wcols = wc ? atoi(wc) : 1;
        long G = g ? atol(g) : 2048 / ((C / 64) * ndir);   // measured at the 30-minute shape: 512 blocks 145 us, 1024: 128, 2048: 123
        if (G < 1) G = 1;
#define PAFC_WS(NW, WCOLS)                                                                                               \
    hipLaunchKernelGGL((pafc::tmix_lora_mix4_ws_kernel<NW, WCOLS>), dim3((unsigned)G, (C / 64) / WCOLS, ndir), dim3(NW * 64), 0, \
                       (hipStream_t)stream, T, C, rows, ndir, reverse0, (const pafc::bf16_t *)x, (const pafc::bf16_t *)t,  \
                       (const pafc::bf16_t *)w2t, (const pafc::bf16_t *)maa, (pafc::bf16_t *)z, (const pafc::bf16_t *)prev)
        if (wcols == 8 && (C / 64) % 8 == 0) PAFC_WS(8, 8);
        else if (wcols == 4 && (C / 64) % 4 == 0) PAFC_WS(4, 4);
        else PAFC_WS(4, 1);
#undef PAFC_WS
        return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
    }
    if (e && e[0] == '1')
        hipLaunchKernelGGL(pafc::tmix_lora_mix4_kernel<true>, grid, block, 0, (hipStream_t)stream, T, C, rows, ndir, reverse0,
                           (const pafc::bf16_t *)x, (const pafc::bf16_t *)t, (const pafc::bf16_t *)w2t,
                           (const pafc::bf16_t *)maa, (pafc::bf16_t *)z, (const pafc::bf16_t *)prev);
    else
        hipLaunchKernelGGL(pafc::tmix_lora_mix4_kernel<false>, grid, block, 0, (hipStream_t)stream, T, C, rows, ndir, reverse0,
                           (const pafc::bf16_t *)x, (const pafc::bf16_t *)t, (const pafc::bf16_t *)w2t,
                           (const pafc::bf16_t *)maa, (pafc::bf16_t *)z, (const pafc::bf16_t *)prev);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

int pafc_decay_lora_bf16(long rows, int C, int H, int ndir, const void *zw, const void *d1n, const void *d2n, const void *bias,
                         void *w, pafc_stream_t stream) {
    if (!zw || !d1n || !d2n || !w) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || ndir < 1 || ndir > 2) return PAFC_ERR_BAD_DIMS;
    if (C != pafc::DL_C || H != pafc::DL_H) return PAFC_ERR_UNSUPPORTED;
    const size_t lds = 2 * 65536 + pafc::DL_WAVES * 16 * (64 + 8) * sizeof(pafc::bf16_t);
    if (hipFuncSetAttribute((const void *)pafc::decay_lora_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return PAFC_ERR_LAUNCH;
    const long ntiles = (rows + 15) / 16;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return PAFC_ERR_LAUNCH;
    long grid = cus / ndir;                       // one block per CU in all (128 KiB of LDS each)
    if (grid > (ntiles + pafc::DL_WAVES - 1) / pafc::DL_WAVES) grid = (ntiles + pafc::DL_WAVES - 1) / pafc::DL_WAVES;
    hipLaunchKernelGGL(pafc::decay_lora_kernel, dim3((unsigned)grid, ndir), dim3(pafc::DL_WAVES * 64), lds, (hipStream_t)stream, rows, ndir,
                       (const pafc::bf16_t *)zw, (const pafc::bf16_t *)d1n, (const pafc::bf16_t *)d2n,
                       (const pafc::bf16_t *)bias, (pafc::bf16_t *)w);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

int pafc_tmix_lora_down_bf16(int B, int T, int C, int N, int ndir, int reverse0, const void *x, const void *maa_x,
                             const void *w1n, void *t, pafc_stream_t stream) {
    return pafc_tmix_lora_down_bf16_prev(B, T, C, N, ndir, reverse0, x, maa_x, w1n, nullptr, t, stream);
}

int pafc_tmix_lora_down_bf16_prev(int B, int T, int C, int N, int ndir, int reverse0, const void *x, const void *maa_x,
                                  const void *w1n, const void *prev, void *t, pafc_stream_t stream) {
    if (!x || !maa_x || !w1n || !t) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T <= 0 || ndir < 1 || ndir > 2) return PAFC_ERR_BAD_DIMS;
    if (C != pafc::LD_C || N != pafc::LD_N) return PAFC_ERR_UNSUPPORTED;
    const size_t lds = 2 * 65536;
    const long rows = (long)B * T, ntiles = (rows + 15) / 16;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return PAFC_ERR_LAUNCH;
    const char *e = getenv("PAFC_LORA_DOWN_WAVES");      // A/B measurements: 8 or 16 waves per block (one block per CU)
    const int nw = e ? atoi(e) : 8;
    long grid = cus / ndir;                       // one block per CU in all (128 KiB of LDS each)
    if (grid > (ntiles + nw - 1) / nw) grid = (ntiles + nw - 1) / nw;
#define PAFC_DOWN(NW)                                                                                                      \
    do {                                                                                                                   \
        if (hipFuncSetAttribute((const void *)pafc::tmix_lora_down_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)lds) != hipSuccess)                                                                   \
            return PAFC_ERR_LAUNCH;                                                                                        \
        hipLaunchKernelGGL(pafc::tmix_lora_down_kernel<NW>, dim3((unsigned)grid, ndir), dim3(NW * 64), lds, (hipStream_t)stream, T, \
                           rows, reverse0, (const pafc::bf16_t *)x, (const pafc::bf16_t *)maa_x, (const pafc::bf16_t *)w1n, \
                           (pafc::bf16_t *)t, (const pafc::bf16_t *)prev);                                                 \
    } while (0)
    if (nw == 8) PAFC_DOWN(8);
    else PAFC_DOWN(16);
#undef PAFC_DOWN
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // extern "C"

