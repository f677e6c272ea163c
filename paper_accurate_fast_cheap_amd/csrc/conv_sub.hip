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
// Block = 256 threads (4 waves, 2 x 2); each wave owns (16 WM) x (16 WN) outputs = WM x WN MFMA 16x16x32 tiles, K
// step 64.  A and B tiles go global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPRs): the LDS image is
// lane-linear, so the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied to the per-lane
// SOURCE address (chunk p of row r holds source chunk p ^ (r & 7)).  STAGES LDS buffers: the loads of K-step
// i + STAGES - 1 are in flight while step i is multiplied.
// Tile shape is the lever against the real limit of this loop, LDS bandwidth: a 64 x 64 wave tile reads 512 B of LDS
// per MFMA = the CU's whole 128 B/clk at full MFMA rate; 128 x 64 reads 384 B, 128 x 128 256 B (and then needs
// the 512-register budget of one wave per SIMD).
// Epilogue: bias + ReLU on the fp32 accumulators, one rounding to bf16, staged through LDS and stored as whole rows.
// Blocks are numbered so that the N-tiles of one M-tile run on one XCD (they share the A tile in that XCD's L2).
#include <stdlib.h>
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int BK = 64;
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

template <int WM, int WN, int STAGES, int BLOCKS_PER_CU>
__global__ __launch_bounds__(256, BLOCKS_PER_CU) void conv3x3s2_kernel(const ConvParams p) {
    constexpr int BM = 32 * WM, BN = 32 * WN;          // block tile (2 x 2 waves)
    constexpr int STAGE = (BM + BN) * BK;              // elements per stage: A rows then B rows
    constexpr int AJ = BM / 32, BJ = BN / 32;          // DMA instructions (8 rows each) per wave for A / B
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
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

    // ---- per-lane DMA sources: wave w fills rows [w BM/4, (w+1) BM/4) of A and the same share of B ---------------
    const int sub = lane >> 3, pch = lane & 7;             // row within the 8-row group, LDS chunk position
    const bf16_t *a_src[AJ];
    const bf16_t *b_src[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = wave * (BM / 4) + j * 8 + sub;
        const int c = pch ^ (row & 7);                      // source chunk that lands at position pch
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;                          // clamp: the row is computed but never stored
        const int f2 = (int)(m % p.F2);
        const long bt = m / p.F2;
        const int t2 = (int)(bt % p.T2);
        const int b = (int)(bt / p.T2);
        a_src[j] = p.in + (((long)b * p.T1 + 2 * t2) * p.F1 + 2 * f2) * p.Ci + 8 * c;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = wave * (BN / 4) + j * 8 + sub;
        b_src[j] = p.wt + (long)(n0 + row) * p.Ci + 8 * (pch ^ (row & 7));
    }
    const int kblocks = p.Ci / BK;
    const int iters = 9 * kblocks;
    const long tap_stride_b = (long)p.Co * p.Ci;

    auto issue = [&](int it, int buf) {
        const int tap = it / kblocks, kb = it - tap * kblocks;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const long aoff = ((long)kh * p.F1 + kw) * p.Ci + kb * BK;
        const long boff = tap * tap_stride_b + kb * BK;
        bf16_t *A = lds + buf * STAGE;
        bf16_t *Bt = A + BM * BK;
#pragma unroll
        for (int j = 0; j < AJ; ++j) dma16(a_src[j] + aoff, A + (wave * (BM / 4) + j * 8) * BK);
#pragma unroll
        for (int j = 0; j < BJ; ++j) dma16(b_src[j] + boff, Bt + (wave * (BN / 4) + j * 8) * BK);
    };

    f32x4c acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4c{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < iters) issue(s, s);
    for (int it = 0; it < iters; ++it) {
        // the stage of iteration `it` has landed when at most the (STAGES - 2) younger stages are still in flight
        if (STAGES == 2 || it + STAGES - 2 >= iters) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (AJ + BJ)) : "memory");
        __syncthreads();                                    // stage it % STAGES landed; everyone left the stage refilled next
        if (it + STAGES - 1 < iters) issue(it + STAGES - 1, (it + STAGES - 1) % STAGES);
        const bf16_t *A = lds + (it % STAGES) * STAGE;
        const bf16_t *Bt = A + BM * BK;
        // LDS -> register reads are software-pipelined one unit (= one A fragment x WN B fragments = WN MFMAs) ahead,
        // B fragments of the second k-step double-buffered and fetched a few per unit: with one or two waves per SIMD
        // nobody else hides the ds_read latency, and left alone the compiler waits for every fragment it just asked for
        auto ldA = [&](int ks, int i) {
            const int row = wm * (16 * WM) + i * 16 + fr;
            return *reinterpret_cast<const bf16x8c *>(A + row * BK + (((ks * 4 + kq) ^ (row & 7)) * 8));
        };
        auto ldB = [&](int ks, int j) {
            const int row = wn * (16 * WN) + j * 16 + fr;
            return *reinterpret_cast<const bf16x8c *>(Bt + row * BK + (((ks * 4 + kq) ^ (row & 7)) * 8));
        };
        bf16x8c bq[2][WN], aq[2];
#pragma unroll
        for (int j = 0; j < WN; ++j) bq[0][j] = ldB(0, j);
        aq[0] = ldA(0, 0);
#pragma unroll
        for (int u = 0; u < 2 * WM; ++u) {
            const int ks = u / WM, i = u % WM;
            if (u + 1 < 2 * WM) aq[(u + 1) & 1] = ldA((u + 1) / WM, (u + 1) % WM);
            if (ks == 0) {
#pragma unroll
                for (int j = (i * WN) / WM; j < ((i + 1) * WN) / WM; ++j) bq[1][j] = ldB(1, j);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[u & 1], bq[ks][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // all waves done with the operand buffers: reuse them to stage the output tile

    // ---- epilogue: C/D layout col = lane & 15 (n), row = 4 (lane >> 4) + reg (m) --------------------------------
    constexpr int LDO = BN + 8;
    bf16_t *O = lds;   // [BM][BN + 8] bf16 (the launch sizes the LDS for the larger of stages and staging)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = wn * (16 * WN) + j * 16 + fr;
        const float bv = p.bias ? bf16_bits_to_f32(p.bias[n0 + col]) : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = acc[i][j][g] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                O[(wm * (16 * WM) + i * 16 + 4 * kq + g) * LDO + col] = (bf16_t)f32_to_bf16_bits(v);
            }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;            // 16-byte chunks per output row
    constexpr int RPP = 256 / CPR;         // rows per pass of the block
#pragma unroll
    for (int q = 0; q < BM / RPP; ++q) {
        const int row = q * RPP + tid / CPR, c8 = (tid % CPR) * 8;
        const long m = m0 + row;
        if (m < p.M)
            *reinterpret_cast<uint4 *>(p.out + m * p.Co + n0 + c8) = *reinterpret_cast<const uint4 *>(O + row * LDO + c8);
    }
}

// ---- first convolution: Conv2d(1, C, 3, 2) + ReLU, one input channel, NHWC out ----------------------------------------
// 9 MACs per output against 2 bytes written: a pure write-bound pass (3.6 GB for a 30-minute file).  A block takes
// C1F_ROWS output rows (b, t1) in turn: the three input rows of the current one sit in LDS as fp32, a thread owns 8
// consecutive channels -- its 9 x 8 weights stay in registers for the whole block (fetching them per output row was a
// third of the kernel: 1000 -> 700 us = 5.1 TB/s of stores) -- and walks the row's positions, storing 16 bytes per
// position: every wave writes whole 1 KiB rows.
constexpr int C1F_ROWS = 4;   // output rows per block: the 9 x 8 weights of a thread are fetched once per block

__global__ __launch_bounds__(256) void conv3x3s2_c1_kernel(int T, int F, int T1, int F1, int C, long nrows, const bf16_t *x,
                                                           const bf16_t *w /* (C, 9) */, const bf16_t *bias, bf16_t *out,
                                                           int relu) {
    extern __shared__ float s_x[];   // [3][F]
    const int tid = threadIdx.x;
    const int cgs = C / 8;                              // channel groups per position
    const int ppi = 256 / cgs;                          // positions per block iteration (C = 512: 4)
    const int cg = tid % cgs, pl = tid / cgs;
    float wr[9][8], bv[8];
    {
        const bf16_t *wp = w + (long)cg * 8 * 9;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wr[k][c] = bf16_bits_to_f32(wp[c * 9 + k]);
            bv[c] = bias ? bf16_bits_to_f32(bias[cg * 8 + c]) : 0.f;
        }
    }
    const long r_end = min(nrows, ((long)blockIdx.x + 1) * C1F_ROWS);
    for (long bt = (long)blockIdx.x * C1F_ROWS; bt < r_end; ++bt) {   // bt = b * T1 + t1
        const int b = (int)(bt / T1), t1 = (int)(bt % T1);
        const bf16_t *xr = x + ((long)b * T + 2 * t1) * F;
        __syncthreads();
        for (int i = tid; i < 3 * F; i += 256) s_x[i] = bf16_bits_to_f32(xr[i]);
        __syncthreads();
        bf16_t *orow = out + bt * (long)F1 * C + cg * 8;
        for (int f1 = pl; f1 < F1; f1 += ppi) {
            float xv[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = s_x[kh * F + 2 * f1 + kw];
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = bv[c];
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaf(xv[k], wr[k][c], acc[c]);
            if (relu) {
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaxf(acc[c], 0.f);
            }
            uint4 o;
            o.x = f32_to_bf16_bits(acc[0]) | (f32_to_bf16_bits(acc[1]) << 16);
            o.y = f32_to_bf16_bits(acc[2]) | (f32_to_bf16_bits(acc[3]) << 16);
            o.z = f32_to_bf16_bits(acc[4]) | (f32_to_bf16_bits(acc[5]) << 16);
            o.w = f32_to_bf16_bits(acc[6]) | (f32_to_bf16_bits(acc[7]) << 16);
            *reinterpret_cast<uint4 *>(orow + (long)f1 * C) = o;
        }
    }
}

// ---- first convolution, weight / bias gradient for the training step (config c4) ---------------------------------------
//   dW[c][k] = sum_{b,t1,f1} dY[b][t1][f1][c] x[b][2 t1 + kh][2 f1 + kw],  db[c] = sum dY,   dY = dA * (A > 0)
// (A = the forward's ReLU output, dA its incoming gradient: the ReLU mask is applied here, so the 1.3 GB gradient map of
// the c4 step is read once and never rewritten).  Same ownership as the forward: a thread owns 8 channels and a quarter
// of a row's positions, the three input rows of an output row sit in LDS; a block walks C1_ROWS output rows keeping its
// 10 x 8 sums in registers and leaves one fp32 partial per (k, c); a second kernel adds the partials in a fixed order.
constexpr int C1_ROWS = 32;

__global__ __launch_bounds__(256) void conv3x3s2_c1_wgrad_kernel(int T, int F, int T1, int F1, int C, long nrows,
                                                                 const bf16_t *__restrict__ x, const bf16_t *__restrict__ act,
                                                                 const bf16_t *__restrict__ dact, float *__restrict__ part) {
    extern __shared__ float s_x[];   // [3][F], then reused for the cross-wave sums
    const int tid = threadIdx.x;
    const int cgs = C / 8, ppi = 256 / cgs;
    const int cg = tid % cgs, pl = tid / cgs;
    float acc[10][8];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[k][c] = 0.f;
    const long r_end = min(nrows, ((long)blockIdx.x + 1) * C1_ROWS);
    for (long bt = (long)blockIdx.x * C1_ROWS; bt < r_end; ++bt) {
        const int b = (int)(bt / T1), t1 = (int)(bt % T1);
        const bf16_t *xr = x + ((long)b * T + 2 * t1) * F;
        __syncthreads();
        for (int i = tid; i < 3 * F; i += 256) s_x[i] = bf16_bits_to_f32(xr[i]);
        __syncthreads();
        const bf16_t *arow = act + bt * (long)F1 * C + cg * 8, *grow = dact + bt * (long)F1 * C + cg * 8;
        for (int f1 = pl; f1 < F1; f1 += ppi) {
            float a[8], g[8];
            Elem<bf16_t>::unpack(*reinterpret_cast<const uint4 *>(arow + (long)f1 * C), a);
            Elem<bf16_t>::unpack(*reinterpret_cast<const uint4 *>(grow + (long)f1 * C), g);
#pragma unroll
            for (int c = 0; c < 8; ++c) g[c] = a[c] > 0.f ? g[c] : 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float xv = s_x[kh * F + 2 * f1 + kw];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[kh * 3 + kw][c] = fmaf(g[c], xv, acc[kh * 3 + kw][c]);
                }
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[9][c] += g[c];
        }
    }
    // sums of the ppi position lanes of a channel group -> [k][C] partial of this block
    float *po = part + (size_t)blockIdx.x * 10 * C;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) s_x[pl * C + cg * 8 + c] = acc[k][c];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float s = 0.f;
            for (int q = 0; q < ppi; ++q) s += s_x[q * C + c];
            po[(size_t)k * C + c] = s;
        }
    }
}

__global__ __launch_bounds__(1024) void conv_c1_wgrad_reduce_kernel(int n, int nblk, const float *__restrict__ part,
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

// ---- fp32 activations at bf16 matrix-core speed: split operands ---------------------------------------------------------
// An fp32 model (the YAML default keeps everything but the slot in fp32) would run conv2 on the fp32 MFMA path: 57 ms per
// 30-minute file through the library against 3.8 ms in bf16.  Every fp32 value is hi + lo with hi = bf16(x) and
// lo = bf16(x - hi) (16 significant bits); a product needs three bf16 MFMAs (hi hi + lo hi + hi lo; lo lo is below
// 2^-16 relative) and accumulates in fp32 -- ~1e-5 relative to the fp32 convolution, two orders inside the 1e-3 bar, at
// 3x the bf16 cost instead of 15x.  conv1 writes its output directly as the two bf16 planes (the same bytes as one fp32
// plane), the weights are split once when the plan is built.

// conv1 for fp32 inputs: fp32 arithmetic, output as hi / lo bf16 planes.  As the bf16 kernel above, a block walks C1F_ROWS
// output rows with its 9 x 8 weights per thread in registers (round 5: fetched per output row they were 73 KB of L2 reads per
// 80 KB of output: 1 830 us for the 7.2 GB of planes of a 30-minute file = 3.9 TB/s).
__global__ __launch_bounds__(256) void conv3x3s2_c1_split_kernel(int T, int F, int T1, int F1, int C, long nrows, const float *x,
                                                                 const float *w /* (C, 9) */, const float *bias,
                                                                 bf16_t *out_hi, bf16_t *out_lo, int relu,
                                                                 long ps /* elements per output pixel in each plane */) {
    extern __shared__ float s_x[];   // [3][F]
    const int tid = threadIdx.x;
    const int cgs = C / 8, ppi = 256 / cgs;
    const int cg = tid % cgs, pl = tid / cgs;
    float wr[9][8], bv[8];
    {
        const float *wp = w + (long)cg * 8 * 9;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wr[k][c] = wp[c * 9 + k];
            bv[c] = bias ? bias[cg * 8 + c] : 0.f;
        }
    }
    const long r_end = min(nrows, ((long)blockIdx.x + 1) * C1F_ROWS);
    for (long bt = (long)blockIdx.x * C1F_ROWS; bt < r_end; ++bt) {   // bt = b * T1 + t1
        const int b = (int)(bt / T1), t1 = (int)(bt % T1);
        const float *xr = x + ((long)b * T + 2 * t1) * F;
        __syncthreads();
        for (int i = tid; i < 3 * F; i += 256) s_x[i] = xr[i];
        __syncthreads();
        const long obase = bt * (long)F1 * ps + cg * 8;
        for (int f1 = pl; f1 < F1; f1 += ppi) {
            float xv[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = s_x[kh * F + 2 * f1 + kw];
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = bv[c];
            // same summation order as the framework's direct convolution is not defined; fp32 FMA chain over the 9 taps
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaf(xv[k], wr[k][c], acc[c]);
            unsigned hi[8], lo[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float v = relu ? fmaxf(acc[c], 0.f) : acc[c];
                hi[c] = f32_to_bf16_bits(v);
                lo[c] = f32_to_bf16_bits(v - bf16_bits_to_f32(hi[c]));
            }
            uint4 oh, ol;
            oh.x = hi[0] | (hi[1] << 16); oh.y = hi[2] | (hi[3] << 16); oh.z = hi[4] | (hi[5] << 16); oh.w = hi[6] | (hi[7] << 16);
            ol.x = lo[0] | (lo[1] << 16); ol.y = lo[2] | (lo[3] << 16); ol.z = lo[4] | (lo[5] << 16); ol.w = lo[6] | (lo[7] << 16);
            *reinterpret_cast<uint4 *>(out_hi + obase + (long)f1 * ps) = oh;
            *reinterpret_cast<uint4 *>(out_lo + obase + (long)f1 * ps) = ol;
        }
    }
}

struct ConvSplitParams {
    const bf16_t *in_hi, *in_lo;   // (B, T1, F1, Ci) each
    const bf16_t *wt_hi, *wt_lo;   // (9, Co, Ci) each
    const float *bias;             // (Co) or null
    float *out;                    // (B, T2, F2, Co) fp32
    int B, T1, F1, Ci, Co, T2, F2;
    long M;
    int relu;
    int mtiles, ntiles;
};

// 128 x 128 x 64 tiles, stage = [A_hi | A_lo | W_hi | W_lo] (64 KiB), two stages, one block per CU
__global__ __launch_bounds__(256, 1) void conv3x3s2_split_kernel(const ConvSplitParams p) {
    constexpr int BM = 128, BN = 128, TILE = 128 * BK, STAGE = 4 * TILE;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long nblk = (long)p.mtiles * p.ntiles;
    long bid = blockIdx.x;
    const long per = nblk / 8;
    if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    const int mt0 = (int)(bid / p.ntiles), nt0 = (int)(bid % p.ntiles);
    const long m0 = (long)mt0 * BM;
    const int n0 = nt0 * BN;
    const int sub = lane >> 3, pch = lane & 7;
    long a_off[4], b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + sub;
        const int c = pch ^ (row & 7);
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        const int f2 = (int)(m % p.F2);
        const long bt = m / p.F2;
        const int t2 = (int)(bt % p.T2);
        const int b = (int)(bt / p.T2);
        a_off[j] = (((long)b * p.T1 + 2 * t2) * p.F1 + 2 * f2) * p.Ci + 8 * c;
        b_off[j] = (long)(n0 + row) * p.Ci + 8 * c;
    }
    const int kblocks = p.Ci / BK;
    const int iters = 9 * kblocks;
    const long tap_stride_b = (long)p.Co * p.Ci;
    auto issue = [&](int it, int buf) {
        const int tap = it / kblocks, kb = it - tap * kblocks;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const long aoff = ((long)kh * p.F1 + kw) * p.Ci + kb * BK;
        const long boff = tap * tap_stride_b + kb * BK;
        bf16_t *S = lds + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rowbase = (wave * 32 + j * 8) * BK;
            dma16(p.in_hi + a_off[j] + aoff, S + rowbase);
            dma16(p.in_lo + a_off[j] + aoff, S + TILE + rowbase);
            dma16(p.wt_hi + b_off[j] + boff, S + 2 * TILE + rowbase);
            dma16(p.wt_lo + b_off[j] + boff, S + 3 * TILE + rowbase);
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
        __syncthreads();
        if (it + 1 < iters) issue(it + 1, (it + 1) & 1);
        const bf16_t *S = lds + (it & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8c ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                const int o = row * BK + (((ks * 4 + kq) ^ (row & 7)) * 8);
                ah[i] = *reinterpret_cast<const bf16x8c *>(S + o);
                al[i] = *reinterpret_cast<const bf16x8c *>(S + TILE + o);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + fr;
                const int o = row * BK + (((ks * 4 + kq) ^ (row & 7)) * 8);
                bh[j] = *reinterpret_cast<const bf16x8c *>(S + 2 * TILE + o);
                bl[j] = *reinterpret_cast<const bf16x8c *>(S + 3 * TILE + o);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // small terms first, the dominant hi * hi last
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    constexpr int LDF = BN + 4;
    float *O = reinterpret_cast<float *>(lds);   // [128][132] fp32 = 66 KiB of the 128 KiB
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = wn * 64 + j * 16 + fr;
        const float bv = p.bias ? p.bias[n0 + col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = acc[i][j][g] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                O[(wm * 64 + i * 16 + 4 * kq + g) * LDF + col] = v;
            }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = q * 8 + (tid >> 5), c4 = (tid & 31) * 4;
        const long m = m0 + row;
        if (m < p.M)
            *reinterpret_cast<float4 *>(p.out + m * p.Co + n0 + c4) = *reinterpret_cast<const float4 *>(O + row * LDF + c4);
    }
}

template <int WM, int WN, int STAGES, int BPC>
int launch_conv(ConvParams &p, hipStream_t stream) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    if (p.Co % BN) return PAFC_ERR_BAD_DIMS;
    p.mtiles = (int)((p.M + BM - 1) / BM);
    p.ntiles = p.Co / BN;
    constexpr size_t stage_bytes = (size_t)STAGES * (BM + BN) * BK * sizeof(bf16_t);
    constexpr size_t out_bytes = (size_t)BM * (BN + 8) * sizeof(bf16_t);
    const size_t lds = stage_bytes > out_bytes ? stage_bytes : out_bytes;
    auto kern = conv3x3s2_kernel<WM, WN, STAGES, BPC>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PAFC_ERR_LAUNCH;
    const long nblk = (long)p.mtiles * p.ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_conv3x3s2_nhwc_bf16_ph(int B, int T1, int F1, int Ci, int Co, const void *in, const void *w_tap_co_ci,
                                           const void *bias, void *out, int relu, int tile_m, pafc_stream_t stream);

extern "C" int pafc_conv3x3s2_nhwc_bf16(int B, int T1, int F1, int Ci, int Co, const void *in, const void *w_tap_co_ci,
                                        const void *bias, void *out, int relu, pafc_stream_t stream) {
    if (!in || !w_tap_co_ci || !out) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T1 < 3 || F1 < 3 || Ci <= 0 || Co <= 0 || Ci % pafc::BK || Co % 128) return PAFC_ERR_BAD_DIMS;
    {   // problems that fill the chip with 256-wide tiles: the phase-pipelined implicit GEMM (csrc/gemm_ph.hip);
        // PAFC_CONV_TILE (A/B measurements) keeps the kernels of this file
        const long M_ = (long)B * ((T1 - 3) / 2 + 1) * ((F1 - 3) / 2 + 1);
        if (!getenv("PAFC_CONV_TILE") && Co % 256 == 0 && ((M_ + 255) / 256) * (Co / 256) >= 512) {
            const int rc = pafc_conv3x3s2_nhwc_bf16_ph(B, T1, F1, Ci, Co, in, w_tap_co_ci, bias, out, relu, 256, stream);
            if (rc != PAFC_ERR_UNSUPPORTED) return rc;
        }
    }
    pafc::ConvParams p{};
    p.in = (const pafc::bf16_t *)in; p.wt = (const pafc::bf16_t *)w_tap_co_ci; p.bias = (const pafc::bf16_t *)bias;
    p.out = (pafc::bf16_t *)out;
    p.B = B; p.T1 = T1; p.F1 = F1; p.Ci = Ci; p.Co = Co;
    p.T2 = (T1 - 3) / 2 + 1; p.F2 = (F1 - 3) / 2 + 1;
    p.M = (long)B * p.T2 * p.F2;
    p.relu = relu;
    // PAFC_CONV_TILE selects the tile shape for A/B measurements (read per call, no global state); default below.
    const char *e = getenv("PAFC_CONV_TILE");
    const int v = e ? atoi(e) : 0;
    hipStream_t s = (hipStream_t)stream;
    if (v == 1) return pafc::launch_conv<4, 4, 2, 2>(p, s);     // 128 x 128 block, 64 x 64 wave tiles, 2 blocks per CU
    if (v == 2) return pafc::launch_conv<8, 4, 3, 1>(p, s);     // 256 x 128 block, 128 x 64 wave tiles, 3 stages
    if (v == 3) return pafc::launch_conv<8, 4, 2, 1>(p, s);     // 256 x 128 block, 2 stages
    if (v == 4 && Co % 256 == 0) return pafc::launch_conv<8, 8, 2, 1>(p, s);   // 256 x 256 block, 128 x 128 wave tiles
    // default: 128 x 128 blocks, two per CU.  (256 x 256 blocks with 128 x 128 wave tiles halve the LDS traffic per flop
    // and win 6 % on random data in isolation, but lose 3 % inside the model: kept as variant 4 for measurements.)
    return pafc::launch_conv<4, 4, 2, 2>(p, s);
}

extern "C" int pafc_conv3x3s2_c1_nhwc_bf16(int B, int T, int F, int C, const void *x, const void *w_c_9, const void *bias,
                                           void *out, int relu, pafc_stream_t stream) {
    if (!x || !w_c_9 || !out) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T < 3 || F < 3 || C <= 0 || C % 8 || (256 % (C / 8)) || C > 2048) return PAFC_ERR_BAD_DIMS;
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const long nblk = (long)B * T1;
    if (nblk > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipLaunchKernelGGL(pafc::conv3x3s2_c1_kernel, dim3((unsigned)((nblk + pafc::C1F_ROWS - 1) / pafc::C1F_ROWS)), dim3(256),
                       3 * F * sizeof(float), (hipStream_t)stream, T, F, T1, F1, C, nblk, (const pafc::bf16_t *)x,
                       (const pafc::bf16_t *)w_c_9, (const pafc::bf16_t *)bias, (pafc::bf16_t *)out, relu);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_conv3x3s2_c1_nhwc_f32split(int B, int T, int F, int C, const float *x, const float *w_c_9,
                                               const float *bias, void *out_hi, void *out_lo, int relu,
                                               pafc_stream_t stream) {
    return pafc_conv3x3s2_c1_nhwc_f32split_ps(B, T, F, C, x, w_c_9, bias, out_hi, out_lo, C, relu, stream);
}

// ... with `pixel_stride` elements between the pixels of each plane: out_lo = out_hi + C and pixel_stride = 2 C give one
// tensor (B, T1, F1, 2 C) = [hi C | lo C] per pixel, the input of pafc_conv3x3s2_nhwc_split_ph.
extern "C" int pafc_conv3x3s2_c1_nhwc_f32split_ps(int B, int T, int F, int C, const float *x, const float *w_c_9,
                                                  const float *bias, void *out_hi, void *out_lo, long pixel_stride, int relu,
                                                  pafc_stream_t stream) {
    if (!x || !w_c_9 || !out_hi || !out_lo) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T < 3 || F < 3 || C <= 0 || C % 8 || (256 % (C / 8)) || C > 2048 || pixel_stride < C || pixel_stride % 8)
        return PAFC_ERR_BAD_DIMS;
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const long nrows = (long)B * T1;
    const long nblk = (nrows + pafc::C1F_ROWS - 1) / pafc::C1F_ROWS;
    if (nblk > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipLaunchKernelGGL(pafc::conv3x3s2_c1_split_kernel, dim3((unsigned)nblk), dim3(256), 3 * F * sizeof(float),
                       (hipStream_t)stream, T, F, T1, F1, C, nrows, x, w_c_9, bias, (pafc::bf16_t *)out_hi, (pafc::bf16_t *)out_lo,
                       relu, pixel_stride);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" int pafc_conv3x3s2_nhwc_f32split(int B, int T1, int F1, int Ci, int Co, const void *in_hi, const void *in_lo,
                                            const void *w_hi_tap_co_ci, const void *w_lo_tap_co_ci, const float *bias,
                                            float *out, int relu, pafc_stream_t stream) {
    if (!in_hi || !in_lo || !w_hi_tap_co_ci || !w_lo_tap_co_ci || !out) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T1 < 3 || F1 < 3 || Ci <= 0 || Co <= 0 || Ci % pafc::BK || Co % 128) return PAFC_ERR_BAD_DIMS;
    pafc::ConvSplitParams p{};
    p.in_hi = (const pafc::bf16_t *)in_hi; p.in_lo = (const pafc::bf16_t *)in_lo;
    p.wt_hi = (const pafc::bf16_t *)w_hi_tap_co_ci; p.wt_lo = (const pafc::bf16_t *)w_lo_tap_co_ci;
    p.bias = bias; p.out = out;
    p.B = B; p.T1 = T1; p.F1 = F1; p.Ci = Ci; p.Co = Co;
    p.T2 = (T1 - 3) / 2 + 1; p.F2 = (F1 - 3) / 2 + 1;
    p.M = (long)B * p.T2 * p.F2;
    p.relu = relu;
    p.mtiles = (int)((p.M + 127) / 128);
    p.ntiles = Co / 128;
    const size_t lds = 2 * 4 * 128 * pafc::BK * sizeof(pafc::bf16_t);   // 128 KiB
    if (hipFuncSetAttribute((const void *)pafc::conv3x3s2_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return PAFC_ERR_LAUNCH;
    hipLaunchKernelGGL(pafc::conv3x3s2_split_kernel, dim3((unsigned)((long)p.mtiles * p.ntiles)), dim3(256), lds,
                       (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

extern "C" size_t pafc_conv3x3s2_c1_wgrad_workspace_bytes(int B, int T, int C) {
    if (B <= 0 || T < 3 || C <= 0) return 0;
    const long nrows = (long)B * ((T - 3) / 2 + 1);
    return (size_t)((nrows + pafc::C1_ROWS - 1) / pafc::C1_ROWS) * 10 * C * sizeof(float);
}

extern "C" int pafc_conv3x3s2_c1_wgrad_bf16(int B, int T, int F, int C, const void *x, const void *act, const void *dact,
                                            float *dw_db, void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    if (!x || !act || !dact || !dw_db || !workspace) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T < 3 || F < 3 || C <= 0 || C % 8 || (256 % (C / 8)) || C > 2048) return PAFC_ERR_BAD_DIMS;
    if (workspace_bytes < pafc_conv3x3s2_c1_wgrad_workspace_bytes(B, T, C)) return PAFC_ERR_WORKSPACE;
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const long nrows = (long)B * T1;
    const int nblk = (int)((nrows + pafc::C1_ROWS - 1) / pafc::C1_ROWS);
    const size_t lds = sizeof(float) * (size_t)((3 * F > (256 / (C / 8)) * C) ? 3 * F : (256 / (C / 8)) * C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pafc::conv3x3s2_c1_wgrad_kernel, dim3(nblk), dim3(256), lds, s, T, F, T1, F1, C, nrows,
                       (const pafc::bf16_t *)x, (const pafc::bf16_t *)act, (const pafc::bf16_t *)dact, (float *)workspace);
    hipLaunchKernelGGL(pafc::conv_c1_wgrad_reduce_kernel, dim3((10 * C + 15) / 16), dim3(1024), 0, s, 10 * C, nblk,
                       (const float *)workspace, dw_db);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
