// Weight gradient of the dense projections (training step, config c4) as a hand-written bf16 GEMM for gfx950.
// C ABI: include/pafc_encoder_ops.h: pafc_gemm_tn_bf16.
//
//   dW[m][n] = sum_r dY[r][m] * X[r][n]            dY: (R, M), X: (R, N), both row-major (the contraction index r =
//                                                  batch x time is the SLOW axis of both operands)
// i.e. what autograd computes for nn.Linear's weight (grad_output^T @ input), R = 16 000 rows against a 512 x 2048
// output at the c4 shape.  The library's pick for this layout is a 64 x 64 tile walking all of R serially
// (95 us = 130 TFLOP/s for every such product, 18 % of the training step); here
//   * the R axis is split over S blocks per output tile so that every CU holds ONE block of eight waves; each leaves an
//     fp32 partial tile, a second kernel adds the S partials in a fixed order (deterministic: no atomics) and writes
//     dW directly in fp32 (or bf16) -- the bf16 rounding of the library path is skipped for fp32 master weights.  What S
//     costs is partial-tile traffic (S x M x N x 4 bytes written and read once: 256 tiles of 64 KiB = 16 MiB when the chip is
//     full, whatever the problem), so S comes from a cost model in rows (plan()), not from "two blocks per CU";
//   * both operand tiles ([64 r][128 columns], 256-byte rows) go global -> LDS by LDS-DMA exactly as they lie in
//     memory, through a ring of four 32 KiB stages (counted vmcnt, one raw s_barrier per K-step:
//     round 6 -- the two-stage loop of rounds 1-5 drained its queue every step and exposed one L2 -> LDS round trip per
//     64 rows), and the MFMA operands (8 consecutive r per lane) are gathered by the transposing LDS read
//     ds_read_b64_tr_b16 (2 per 16 x 32 operand); the 16-byte chunks of a row are XOR-swizzled on the SOURCE side of
//     the DMA with the pattern that keeps those reads conflict-free:
//         chunk' = chunk ^ (((r & 3) << 2) | ((r >> 2) & 3))
//   * the eight waves are two K groups of 2 x 2 waves: group g multiplies rows [32 g, 32 g + 32) of every stage into its
//     own accumulators (wave tile 64 x 64 = 4 x 4 MFMA 16x16x32 bf16), so two waves per SIMD cover each other's LDS
//     reads while the tile stays 128 x 128; the groups' sums meet in LDS in the epilogue, from where the partial tile
//     leaves in whole 512-byte rows (float4 per lane) instead of 64-byte segments from the accumulator layout;
//   * all tiles of one split sit on one XCD (they share that split's rows of dY and X in its L2);
//   * rows past the end of a block's R range are fetched from the last valid row (in bounds) and the dY tile's rows
//     are zeroed in LDS, so ragged R costs one extra pass over <= 63 LDS rows in one block per tile.
#include <cstdlib>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int TBM = 128, TBN = 128, TBK = 64;
constexpr int TN_NST = 4;                      // LDS stages of 32 KiB
constexpr int TN_STAGE = 2 * TBK * TBM;        // bf16 elements per stage
constexpr int TN_DMA_PER_STAGE = 4;            // LDS-DMA instructions per lane and stage
constexpr int TN_EPI_FLOATS = TBM * (TBN + 4) + TBM;   // one K group's fp32 tile [128][132] + its 128 column sums of dY
typedef float f32x4t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));
typedef short s16x4t __attribute__((ext_vector_type(4)));
typedef short s16x8t __attribute__((ext_vector_type(8)));

struct TnParams {
    const bf16_t *A, *B;     // dY (R, M) with row stride lda; X (R, N) with row stride ldb
    float *part;             // (S, M, N) fp32
    float *bias_part;        // (S, M) fp32 column sums of dY (the bias gradient), or null
    long R, lda, ldb, rows_per_split;
    long sA, sB;             // batch strides of dY and X (elements); the partials of entry z start at part + z * S * M * N
    int M, N, mtiles, ntiles, S, batch;
};

__device__ __forceinline__ void tdma16(const bf16_t *src, bf16_t *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

// Eight transposing LDS reads (ds_read_b64_tr_b16: the lane supplies the address of 4 columns of one row and receives
// 4 rows of one column) as inline assembly.  The intrinsic form (__builtin_amdgcn_ds_read_tr16_b64_v4i16) carries no memory
// operand, so the compiler orders it behind EVERY LDS-DMA in flight -- s_waitcnt vmcnt(0) in front of the first read, which
// is the whole ring.  Here the counters are ours: tr_wait() below is the lgkmcnt(0) every fragment passes through.
#define PAFC_TR8(o, a)                                                                                                    \
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\t"                 \
                 "ds_read_b64_tr_b16 %3, %11\n\tds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\t"               \
                 "ds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15"                                                 \
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])    \
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])                    \
                 : "memory")
// (the wait names the values as in-out operands: nothing that uses them can be scheduled above it)
#define PAFC_TR_WAIT(w, o)                                                                                                 \
    asm volatile(w : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7])::"memory")

__global__ __launch_bounds__(512, 1) void gemm_tn_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];   // [TN_NST stages][A 64x128 | B 64x128] (128 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    // block -> (split, tile): the tiles of one split on one XCD (consecutive block ids go round the eight XCDs)
    long bid = blockIdx.x;
    const long per = (long)gridDim.x / 8;
    if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    const int tiles = p.mtiles * p.ntiles;
    const int zs = (int)(bid / tiles), tile = (int)(bid % tiles);      // zs = batch entry * S + split
    const int z = zs / p.S, split = zs % p.S;
    const bf16_t *const Ab = p.A + (long)z * p.sA, *const Bb = p.B + (long)z * p.sB;
    const int mt = tile / p.ntiles, nt = tile % p.ntiles;
    const int m0 = mt * TBM, n0 = nt * TBN;
    const long k_begin = (long)split * p.rows_per_split;
    const long k_end = min(p.R, k_begin + p.rows_per_split);
    const int iters = k_end > k_begin ? (int)((k_end - k_begin + TBK - 1) / TBK) : 0;

    // DMA: instruction j of wave w brings rows 4 (8 j + w) .. + 3 (1 KiB); lane -> (row, swizzled chunk)
    int d_row[2], a_col[2], b_col[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 8 + wave) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        d_row[j] = row;
        a_col[j] = min(m0 + ch * 8, p.M - 8);     // column tail: clamped chunks feed outputs that are never stored
        b_col[j] = min(n0 + ch * 8, p.N - 8);
    }
    auto issue = [&](int it, int buf) {           // TN_DMA_PER_STAGE = 4 instructions per lane
        bf16_t *At = lds + buf * TN_STAGE;
        bf16_t *Bt = At + TBK * TBM;
        const long kb = k_begin + (long)it * TBK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long r = min(kb + d_row[j], k_end - 1);
            const int base = (j * 8 + wave) * 4 * TBM;
            tdma16(Ab + r * p.lda + a_col[j], At + base);
            tdma16(Bb + r * p.ldb + b_col[j], Bt + base);
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
    // 16 blk + 4 pp .. + 3 and receives column (lane & 15) of the four rows.  Byte offsets inside a stage, [2 i + h]:
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    unsigned rdA[8], rdB[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 8 * g + 4 * h + q;
            const int base = kg * 8192 + 256 * row + 8 * (pp & 1);       // this K group's 32 rows of the stage
            const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
            rdA[2 * i + h] = (unsigned)(base + 16 * (((wm * 8 + 2 * i) | (pp >> 1)) ^ swz));
            rdB[2 * i + h] = (unsigned)(TBK * TBM * 2 + base + 16 * (((wn * 8 + 2 * i) | (pp >> 1)) ^ swz));
        }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;

    // fragments of K-step `st` -> registers (the reads are only ISSUED here; PAFC_TR_WAIT at the top of the step that multiplies
    // them).  The ragged end of the block's R range = its last K-step: rows past the end of the dY tile are zeroed first.
    auto read_frags = [&](int st, s16x4t (&ta)[8], s16x4t (&tb)[8]) {
        const long valid = k_end - (k_begin + (long)st * TBK);
        if (valid < TBK) {   // block-uniform; the stage has landed for everyone (the caller's barrier)
            char *At = (char *)(lds + (st % TN_NST) * TN_STAGE);
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            for (int c = tid; c < (TBK - (int)valid) * 16; c += 512) *reinterpret_cast<uint4 *>(At + valid * 256 + (long)c * 16) = z;
            __syncthreads();
        }
        const unsigned sb = lds0 + (unsigned)(st % TN_NST) * (TN_STAGE * 2);
        unsigned ad[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ad[e] = sb + rdA[e];
        PAFC_TR8(ta, ad);
#pragma unroll
        for (int e = 0; e < 8; ++e) ad[e] = sb + rdB[e];
        PAFC_TR8(tb, ad);
    };
    // One K-step: multiply the fragments in (ca, cb) -- read from LDS during the previous step -- while the next step's are read
    // into (na, nb): a wave's LDS reads run under its own MFMAs (the first form of this loop read, waited and multiplied in turn:
    // with all eight waves in step behind the barrier, LDS and the matrix cores took turns -- 1 900 cycles per K-step).
    //   DMA ring: stages st .. st + TN_NST - 1 are in flight or landed at the top of step st; stage st + 1 has landed when at most
    //   the TN_NST - 2 stages issued after it are outstanding (loads complete in order).
    auto step = [&](int it, s16x4t (&ca)[8], s16x4t (&cb)[8], s16x4t (&na)[8], s16x4t (&nb)[8]) {
        const int newer = max(0, min(TN_NST - 2, iters - 2 - it));
        switch (newer) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TN_DMA_PER_STAGE) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TN_DMA_PER_STAGE) : "memory"); break;
        }
        PAFC_TR_WAIT("s_waitcnt lgkmcnt(0)", ca);     // this wave's reads of stage `it` have returned ...
        PAFC_TR_WAIT("", cb);
        // ... and behind the raw barrier (__syncthreads() would drain the ring: vmcnt(0)) everyone's have, so the buffer of stage `it`
        // takes stage it + TN_NST; and everyone's part of stage it + 1 has landed
        asm volatile("s_barrier" ::: "memory");
        if (it + TN_NST < iters) issue(it + TN_NST, it % TN_NST);
        if (it + 1 < iters) read_frags(it + 1, na, nb);
        bf16x8t af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = __builtin_bit_cast(bf16x8t, __builtin_shufflevector(ca[2 * i], ca[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7));
            bfr[i] = __builtin_bit_cast(bf16x8t, __builtin_shufflevector(cb[2 * i], cb[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7));
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
    };

    s16x4t fa0[8], fb0[8], fa1[8], fb1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) fa0[e] = fb0[e] = fa1[e] = fb1[e] = s16x4t{0, 0, 0, 0};
#pragma unroll
    for (int s_ = 0; s_ < TN_NST; ++s_)
        if (iters > s_) issue(s_, s_);
    if (iters > 0) {
        switch (min(TN_NST - 1, iters - 1)) {          // stage 0 has landed
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TN_DMA_PER_STAGE) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TN_DMA_PER_STAGE) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * TN_DMA_PER_STAGE) : "memory"); break;
        }
        asm volatile("s_barrier" ::: "memory");
        read_frags(0, fa0, fb0);
    }
    for (int it = 0; it < iters; it += 2) {
        step(it, fa0, fb0, fa1, fb1);
        if (it + 1 < iters) step(it + 1, fa1, fb1, fa0, fb0);
    }
    __syncthreads();   // the ring is free: the two K groups' sums meet in its memory

    // C/D layout: column = lane & 15 (n), row = 4 (lane >> 4) + reg (m).  fp32 staging [2 K groups][128][132]: the four 16-lane
    // groups of a store sit 4 rows = 16 banks apart -- conflict-free; the groups' sums are added on the way out
    constexpr int LDF = TBN + 4;
    float *O = reinterpret_cast<float *>(lds) + kg * TN_EPI_FLOATS;
    float *SB = O + TBM * LDF;                       // [128] column sums of dY
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wm * 64 + i * 16 + 4 * g + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) O[row * LDF + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
            if (want_bias && (lane & 15) == 0) SB[row] = accb[i][r];     // all 16 columns of accb hold the same sums
        }
    __syncthreads();
    const float *O0 = reinterpret_cast<const float *>(lds), *O1 = O0 + TN_EPI_FLOATS;
    float *out = p.part + (size_t)zs * p.M * p.N;
#pragma unroll
    for (int qq = 0; qq < TBM * TBN / 4 / 512; ++qq) {
        const int idx = qq * 512 + tid;
        const int row = idx / (TBN / 4), c4 = (idx % (TBN / 4)) * 4;
        const int m = m0 + row, n = n0 + c4;
        if (m < p.M && n < p.N) {    // (N is a multiple of 8: a float4 never straddles the edge)
            float4 a = *reinterpret_cast<const float4 *>(O0 + row * LDF + c4);
            const float4 b = *reinterpret_cast<const float4 *>(O1 + row * LDF + c4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            *reinterpret_cast<float4 *>(out + (size_t)m * p.N + n) = a;
        }
    }
    if (p.bias_part != nullptr && nt == 0 && tid < TBM && m0 + tid < p.M)
        p.bias_part[(size_t)zs * p.M + m0 + tid] = O0[TBM * LDF + tid] + O1[TBM * LDF + tid];
}

// dW = sum of the S partials in a FIXED order (four quarter sums of consecutive partials, then (q0 + q1) + (q2 + q3));
// out_f32 or out_bf16 (exactly one is non-null).  The bias gradient's partials ride in the same launch: indices
// [n4, n4 + nb4) reduce bias_part (stride nb4) into the bias outputs.  A block = 64 float4 columns x 4 quarter sums, four
// independent loads in flight per lane: the one-lane-per-column loop of rounds 1-5 waited out S dependent round trips
// (10-15 us whatever the size).
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(long n4, int S, long stride4, const float4 *__restrict__ part,
                                                             float4 *__restrict__ out_f32, uint2 *__restrict__ out_bf16,
                                                             long nb4, const float4 *__restrict__ bias_part,
                                                             float4 *__restrict__ bias_f32, uint2 *__restrict__ bias_bf16) {
    __shared__ float4 sh[4][64];
    const int col = threadIdx.x & 63, sg = threadIdx.x >> 6;
    long i = (long)blockIdx.x * 64 + col;
    const bool live = i < n4 + nb4;
    {   // batch entry blockIdx.y: its S partials and its outputs
        const long z = blockIdx.y;
        part += z * S * n4;
        if (out_f32) out_f32 += z * n4;
        if (out_bf16) out_bf16 += z * n4;
        if (nb4) {
            bias_part += z * S * nb4;
            if (bias_f32) bias_f32 += z * nb4;
            if (bias_bf16) bias_bf16 += z * nb4;
        }
    }
    if (i >= n4) {                                   // (block-uniform except in one block)
        i -= n4; part = bias_part; stride4 = nb4; out_f32 = bias_f32; out_bf16 = bias_bf16;
    }
    const int per = (S + 3) / 4, s0 = sg * per, s1 = min(S, s0 + per);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float4 *q = part + i;
        int s_ = s0;
        for (; s_ + 4 <= s1; s_ += 4) {
            const float4 b0 = q[(long)s_ * stride4], b1 = q[(long)(s_ + 1) * stride4], b2 = q[(long)(s_ + 2) * stride4],
                         b3 = q[(long)(s_ + 3) * stride4];
            a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
            a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
            a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
        }
        for (; s_ < s1; ++s_) {
            const float4 b = q[(long)s_ * stride4];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    }
    sh[sg][col] = a;
    __syncthreads();
    if (sg != 0 || !live) return;
    const float4 q0 = sh[0][col], q1 = sh[1][col], q2 = sh[2][col], q3 = sh[3][col];
    a.x = (q0.x + q1.x) + (q2.x + q3.x); a.y = (q0.y + q1.y) + (q2.y + q3.y);
    a.z = (q0.z + q1.z) + (q2.z + q3.z); a.w = (q0.w + q1.w) + (q2.w + q3.w);
    if (out_f32) out_f32[i] = a;
    else
        out_bf16[i] = make_uint2(f32_to_bf16_bits(a.x) | (f32_to_bf16_bits(a.y) << 16),
                                 f32_to_bf16_bits(a.z) | (f32_to_bf16_bits(a.w) << 16));
}

// Splits of R per output tile.  Cost of a choice in ROWS of the K loop (one 64-row K-step of a block ~ 0.37 us): rounds of
// one-block-per-CU work x (rows per block + ~448 rows for prologue and the 64 KiB partial-tile store) + the reduce pass over
// S x M x N x 4 bytes at ~3 TB/s (~58 rows per MiB).  Measured at 15 392 rows (profiles/r06x_gemm_tn_by_shape.txt).
void plan(long R, int M, int N, int batch, int *S, long *rows_per_split) {
    const long tiles = (long)((M + TBM - 1) / TBM) * ((N + TBN - 1) / TBN) * batch;
    const long cus = device_cus();
    long max_s = (R + 2 * TBK - 1) / (2 * TBK);         // at least two K steps per block
    if (max_s > 256) max_s = 256;
    if (max_s < 1) max_s = 1;
    long best_s = 1;
    double best = -1.0;
    for (long s = 1; s <= max_s; ++s) {
        const long rps = ((R + s - 1) / s + TBK - 1) / TBK * TBK;
        const long sr = (R + rps - 1) / rps;              // the split count this row share really gives
        const long rounds = (tiles * sr + cus - 1) / cus;
        const double cost = (double)rounds * (double)(rps + 448) + 58.0 * (double)sr * batch * M * N * 4.0 / 1048576.0;
        if (best < 0 || cost < best) { best = cost; best_s = s; }
    }
    if (const char *e = getenv("PAFC_GEMM_TN_S")) {     // A/B runs: force the split count
        const long v = atol(e);
        if (v >= 1 && v <= max_s) best_s = v;
    }
    long rps = ((R + best_s - 1) / best_s + TBK - 1) / TBK * TBK;
    *rows_per_split = rps;
    *S = (int)((R + rps - 1) / rps);
}

}  // namespace
}  // namespace pafc

extern "C" size_t pafc_gemm_tn_batched_workspace_bytes(long R, int M, int N, int batch) {
    if (R <= 0 || M <= 0 || N <= 0 || batch <= 0) return 0;
    int S; long rps;
    pafc::plan(R, M, N, batch, &S, &rps);
    return (size_t)batch * S * M * (N + 1) * sizeof(float);      // partial tiles + partial column sums
}

extern "C" size_t pafc_gemm_tn_workspace_bytes(long R, int M, int N) { return pafc_gemm_tn_batched_workspace_bytes(R, M, N, 1); }

// `batch` products of the same shape in one launch pair (the r / k / v weight gradients of a time-mix block: three 512 x 512
// products of 15 392 rows are 48 tiles -- S = 5 instead of 16 each, a third of the launches): entry z reads dy + z * stride_dy,
// x + z * stride_x and writes dw + z * M * N (dbias + z * M).
extern "C" int pafc_gemm_tn_bf16_batched(long R, int M, int N, int batch, const void *dy, long lda, long stride_dy, const void *x,
                                         long ldb, long stride_x, void *dw, void *dbias, int dw_dtype, void *workspace,
                                         size_t workspace_bytes, pafc_stream_t stream) {
    if (!dy || !x || !dw || !workspace) return PAFC_ERR_NULL_POINTER;
    if (R <= 0 || M < 8 || N < 8 || (M % 8) || (N % 8) || lda < M || ldb < N || (lda % 8) || (ldb % 8) || batch < 1 || batch > 65535 ||
        (batch > 1 && ((stride_dy % 8) || (stride_x % 8))))
        return PAFC_ERR_BAD_DIMS;
    if (dw_dtype != PAFC_F32 && dw_dtype != PAFC_BF16) return PAFC_ERR_DTYPE;
    if ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw | (uintptr_t)dbias | (uintptr_t)workspace) & 15) != 0)
        return PAFC_ERR_ALIGNMENT;
    pafc::TnParams p{};
    pafc::plan(R, M, N, batch, &p.S, &p.rows_per_split);
    if (workspace_bytes < (size_t)batch * p.S * M * (N + 1) * sizeof(float)) return PAFC_ERR_WORKSPACE;
    p.A = (const pafc::bf16_t *)dy; p.B = (const pafc::bf16_t *)x; p.part = (float *)workspace;
    p.bias_part = dbias ? p.part + (size_t)batch * p.S * M * N : nullptr;
    p.R = R; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.batch = batch; p.sA = stride_dy; p.sB = stride_x;
    p.mtiles = (M + pafc::TBM - 1) / pafc::TBM; p.ntiles = (N + pafc::TBN - 1) / pafc::TBN;
    if ((long)p.mtiles * p.ntiles * p.S * batch > 2147483647L) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    size_t lds = (size_t)pafc::TN_NST * pafc::TN_STAGE * sizeof(pafc::bf16_t);                   // the ring: 128 KiB
    if (lds < 2 * pafc::TN_EPI_FLOATS * sizeof(float)) lds = 2 * pafc::TN_EPI_FLOATS * sizeof(float);   // the epilogue: 133 KiB
    if (hipFuncSetAttribute((const void *)pafc::gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return PAFC_ERR_LAUNCH;
    hipLaunchKernelGGL(pafc::gemm_tn_kernel, dim3((unsigned)(p.mtiles * p.ntiles * p.S * batch)), dim3(512), lds, s, p);
    const long n4 = (long)M * N / 4, nb4 = dbias ? M / 4 : 0;
    hipLaunchKernelGGL(pafc::gemm_tn_reduce_kernel, dim3((unsigned)((n4 + nb4 + 63) / 64), (unsigned)batch), dim3(256), 0, s, n4, p.S, n4,
                       (const float4 *)workspace, dw_dtype == PAFC_F32 ? (float4 *)dw : nullptr,
                       dw_dtype == PAFC_BF16 ? (uint2 *)dw : nullptr, nb4, (const float4 *)p.bias_part,
                       dw_dtype == PAFC_F32 ? (float4 *)dbias : nullptr, dw_dtype == PAFC_BF16 ? (uint2 *)dbias : nullptr);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_gemm_tn_bf16(long R, int M, int N, const void *dy, long lda, const void *x, long ldb, void *dw,
                                 void *dbias, int dw_dtype, void *workspace, size_t workspace_bytes,
                                 pafc_stream_t stream) {
    return pafc_gemm_tn_bf16_batched(R, M, N, 1, dy, lda, 0, x, ldb, 0, dw, dbias, dw_dtype, workspace, workspace_bytes, stream);
}
