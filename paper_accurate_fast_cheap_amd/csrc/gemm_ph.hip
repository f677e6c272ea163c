// The long-form projections as a phase-pipelined bf16 GEMM (gfx950): 256 x 256 tiles, 8 waves, K-step 64.
// C ABI: include/pafc_encoder_ops.h: pafc_gemm_bf16 dispatches here for large problems (see gemm_bf16.hip for the
// 128 x 128 kernel that keeps the small and oddly shaped ones); pafc_gemm_ph_ex is the general entry point.
//
//   out[z][m][n] = act(alpha * sum_k A[z][m][k] * W[z][n][k] + bias[z][n] + residual[z][m][n])
// nn.Linear (weight (N, K), K contiguous) with bias, SiLU / tanh / ReLU / GLU, ff_scale and the residual add applied to
// the fp32 accumulator before the single rounding: the FFN, 1x1-conv, r/k/v and output projections of
// ConformerEncoderLayer, ConvolutionModule and RWKV_Tmix_x060c (wenet/transformer/positionwise_feed_forward.py:47-55,
// convolution.py:118-141, rwkv_v6/src/model.py:286-324, encoder_layer.py:201-259) at the 30-minute shape (44 998 rows).
//
// Why this structure.  With one 512-thread block per CU every SIMD hosts two waves.  The two waves of a SIMD run the SAME
// program one barrier apart (waves 4-7 take one extra barrier at the start): while one multiplies (16 MFMAs = 256 cycles of
// the SIMD's matrix pipe) the other reads its next operand fragments from LDS and issues the block's next LDS-DMA piece,
// then they swap.  A K-step of a wave is four such phases, one per quadrant of its 128 x 64 output.
//
// LDS holds two K-steps (2 x 64 KiB), each cut into four UNITS by the phase in which they are read:
//   A_m0 (the first 64 rows of every wave's 128; read in phase 1)   B_n0 (first half of every wave's columns; phase 1)
//   B_n1 (second half of the columns; phase 2)                       A_m1 (the other 64 rows; phase 3)
// A unit is dead two phases after its read phase, so it is re-filled for the K-step after next right then: every phase issues
// exactly one unit (2 LDS-DMA instructions per thread), four to six phases (1 000+ cycles) ahead of its use, and waits only
// `vmcnt(8)` -- the unit issued four phases ago -- never vmcnt(0).  Barriers are raw s_barrier (a __syncthreads() would
// drain the DMA queue); the XOR swizzle of the 16-byte chunks sits on the SOURCE address, the LDS image is lane-linear, and
// the fragment reads apply the same XOR (conflict-free ds_read_b128).
//
// Round 3: the epilogue leaves from registers.  The product is formed transposed (W fragment as the first MFMA operand), so
// a lane owns 4 consecutive output columns of one row per accumulator; the weight rows of a wave's column half are dealt to
// the MFMA's n index so that the lane's two accumulators of a half are 8 CONSECUTIVE columns (row r of the half's LDS image
// holds weight row (r & 15 >> 2) * 8 + (r >> 4) * 4 + (r & 3): a permutation of the LDS-DMA source rows, free): one 16-byte
// store per row group and half, four lanes cover 64 contiguous bytes of a row.  No output image in LDS, no barrier in the
// epilogue -- so LDS belongs to the operands alone and the pipeline runs THROUGH the tile boundary: the staging slots the
// last two K-steps of a tile would leave empty (their units belong to K-steps that do not exist) carry the first six units of
// the block's NEXT tile, in the same regions and phases as in the steady state.  A K = 512 tile used to spend 6 500 cycles
// waiting for its first operands and 10 500 cycles in its epilogue (profiles/r02a_gemm_ph_cycle_stamps.log); now the operands
// of K-step 0 have landed long before the previous epilogue ends.  The residual (bf16 or fp32) is loaded to registers in the
// epilogue.  Output bf16, fp32, or fp32 as two bf16 planes hi | lo (the split-operand form the next GEMM of an fp32 model
// reads as ITS A operand, see below).
//
// fp32 activations on the bf16 matrix cores (fp32 models; the reference's default precision is an fp32 model around the bf16
// slot, rwkv_wrapper_bidirectional.py:40-56): x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significant bits), the
// same for the weight, and x w = hi_x hi_w + lo_x hi_w + hi_x lo_w (+ 2^-16 relative).  As a GEMM: A stored as [hi | lo]
// (M x 2K), W' = [hi_w | hi_w | lo_w] (N x 3K), and the K walk of A visits hi, lo, hi (`a_soff`): three bf16 products on
// the fast path instead of one fp32 product at 1/16 of the rate.  The planes may also alternate in blocks ([hi 512 | lo 512]
// per pixel: what the split convolution writes and Linear(9728, 512) then reads).
//
// Round 5 (SPL): the three products share their operand fragments.  The walk above stages the hi plane of A twice and the
// hi plane of W twice (3 x 64 KiB of LDS-DMA and 3 x 24 fragment reads per wave for 64 fp32-equivalent columns).  With SPL a
// K-step covers 32 columns of BOTH planes: an LDS row is [hi 32 | lo 32] (the same 128-byte rows, units, swizzle, phases and
// counted waits -- only the DMA source of chunks 4-7 moves to the lo plane), the fragment a lane reads at `ks = 0` is hi and at
// `ks = 1` lo, and a phase multiplies hi_w hi_a + hi_w lo_a + lo_w hi_a from the fragments it holds: 24 MFMAs per phase instead
// of 16 behind the same reads, barriers and DMA -- two K-steps (2 x 64 KiB, 2 x 24 reads) per 64 columns, a third less operand
// traffic per product and a third more matrix work per barrier pair.  A and W keep their layouts ([hi | lo] / [hi | hi | lo]).
// PAFC_SPLIT_WALK=hilohi selects the round-3 walk for A/B runs.
#include <climits>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));

struct PhParams {
    const bf16_t *A, *W;
    const void *bias;                 // bf16 (OUT 0) or fp32 (OUT 1, 2)
    const void *res;                  // bf16 (RES 1) or fp32 (RES 2)
    void *out;                        // bf16 / fp32 / bf16 planes hi | lo
    long M;
    int N, K;                         // K = columns walked by the K loop (3 x the logical K for a split-operand A)
    long lda, ldw, ldo, ldr;          // row strides (elements of the respective type)
    long sA, sW, sO, sB, sR;          // batch strides (elements); sB = 0 shares the bias
    float alpha;
    int mtiles, ntiles;
    int tm;                           // rows per tile actually used: 256, 192, 128 or 64 (balances the grid over the CUs)
    int batch;
    // split-operand A (planes hi | lo of an fp32 activation): the K walk visits hi, lo, hi -- nk1 64-deep K-steps each
    // (nk1 = a huge number: a plain bf16 A).  The planes alternate in blocks of 64 << pb_shift columns: [hi PB | lo PB] ...
    // (pb_shift = 31: one block, the row is [hi K | lo K]); pb_bytes = bytes of one plane block.
    int nk1, pb_shift;
    long pb_bytes;
    // SPL: K-steps per tile (32 columns of both planes each), and the byte offset of the lo plane from the hi plane inside a
    // row of A / of W (pb_shift then counts in 32-column steps)
    int nsteps, a_lo, w_lo;
    int w_step;                       // SPL: bytes a K-step advances inside a row of W (64: [hi | hi | lo] planes)
    long lo_off;                      // OUT 2: column offset of the lo plane inside an output row
    // LayerNorm folded into the GEMMs either side of it (LNF): row statistics as 8 partial (sum, sum of squares) pairs per
    // row, float2 [M][8] -- written by the GEMM that produces the row (LNF 2: one pair per 64-column wave slice of the
    // 512-column output) and read by the GEMM that consumes it (LNF 1) together with ln_csum[n] = sum_k W'[n][k].
    float *ln_stats;
    const float *ln_csum;
    float ln_eps, ln_inv_c;
    // implicit-GEMM mode (CONV): A is an NHWC image (B, T1, F1, Ci), row m = output position (b, t2, f2) of a 3 x 3 stride-2
    // convolution, K = 9 taps x Ci; W is (9, N, Ci) tap-major.  K-step kt = (tap, 64-channel slice).
    // CiA: elements per input pixel (Ci, or 2 Ci for planes [hi Ci | lo Ci]); CiW: elements per weight row and tap (Ci, or
    // 3 Ci = [hi | hi | lo]); spt = CiW / 64 K-steps per tap, inv_spt = ceil(2^16 / spt) (tap = kt * inv_spt >> 16, checked on
    // the host for every kt the kernel forms).
    int T1, F1, T2, F2, Ci, CiA, CiW, spt, inv_spt;
    long in_bytes;                    // bytes of the whole image tensor
#ifdef PH_STAMPS
    unsigned long long *stamps;       // diagnostic build only: [block][2 waves][64] shader-clock stamps (s_memtime)
#endif
};

#ifdef PH_STAMPS
#define PH_STAMP(i) do { if (stamp_on) st[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_STAMP(i) do { } while (0)
#endif

// LDS-DMA through a buffer descriptor: 16 bytes per lane from (descriptor base + per-lane byte offset + wave-uniform byte
// offset) to (wave-uniform LDS address + 16 * lane).  One VGPR of address per source chunk (a flat pointer costs two, and
// 64-bit adds), and the K offset rides in an SGPR.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uniform_off, void *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, lane_off,
                                             uniform_off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}

typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {     // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2p{lo, hi}, bf16x2p));
}

template <int ACT>
__device__ __forceinline__ float act_apply(float v) {
    if constexpr (ACT == 1) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    if constexpr (ACT == 2) return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);   // saturates correctly at +-inf
    if constexpr (ACT == 3) return fmaxf(v, 0.f);
    return v;
}

// cache policy of the output stores (buffer_store aux bits: 1 = sc0, 2 = nt, 16 = sc1).  Measured with nt = non-temporal (round 4,
// profiles/r04i_step_hbm_bytes_nt_stores.txt): the w_1 GEMM's A-panel re-reads halve (147 -> 72 MB per launch: the 128 KiB a
// tile writes no longer evict the panel from the XCD's 4 MiB L2) and w_1 / w_2 gain 3 %, but every 64-byte half line a store
// instruction writes goes to memory on its own (writes +35 %), the CTC head (N = 5000) loses 45 % and the step is unchanged
// (21.60 vs 21.70 ms): default policy kept
#ifndef PH_ST_AUX
#define PH_ST_AUX 0
#endif
#define PH_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int PBN = 256, PBK = 64;      // tile columns, K-step (tile rows: PhParams::tm <= 256)
constexpr unsigned PH_OOB = 0xC0000000u;       // a byte offset beyond every descriptor: the hardware drops the access

// GLU: weight rows in blocks of 64 = 32 values + the 32 gates of the same channels (a wave's 64 columns are one block, its
//      column half 0 the values and half 1 the gates; the output has N / 2 columns).
// ACT: 0 none, 1 SiLU, 2 tanh, 3 ReLU (not with GLU or a residual).
// RES: 0 none, 1 bf16 residual, 2 fp32 residual (both loaded to registers in the epilogue).
// OUT: 0 bf16, 1 fp32, 2 fp32 as bf16 planes hi | lo.  With OUT != 0 the bias is fp32.
// CONV: the second subsampling convolution as an implicit GEMM.
// LNF:  the pre-norm LayerNorm between a residual GEMM and the projection that follows it, folded into the two:
//       LN(x) W^T + b = rstd (x W'^T - mean csum) + b'  with W' = gamma * W, csum[n] = sum_k W'[n][k], b' = b + W beta.
//       2 = producer (RES 1): besides x_new the epilogue writes each row's partial (sum, sum of squares) of its 64 columns;
//       1 = consumer (SiLU or GLU, alpha 1): A is the UN-normalised stream, W / bias are W' / b', the epilogue applies
//       rstd (acc - mean csum) + b' per row.  The normalised tensor never exists in memory (encoder_layer.py:201-259 writes
//       and re-reads it once per sub-block).
template <bool GLU, int ACT, int RES, int OUT, bool CONV = false, int LNF = 0, bool SPL = false>
__global__ __launch_bounds__(512, 2) void gemm_ph_kernel(const PhParams p) {
    constexpr int BN = PBN;
    constexpr int UA = 128 * 128;                 // bytes of an A unit: 128 rows x 64 k
    constexpr int UB = (BN / 2) * 128;            // bytes of a B unit
    constexpr int STEP = 2 * UA + 2 * UB;         // one K-step in LDS
    constexpr int OFF_A0 = 0, OFF_A1 = UA, OFF_B0 = 2 * UA, OFF_B1 = 2 * UA + UB;
    constexpr int TN = 2;                         // 16-column tiles per column half
    constexpr int DA = UA / 8192;                 // LDS-DMA instructions per thread for an A unit (2)
    constexpr int DB = UB / 8192;                 // ... for a B unit (2)
    constexpr int OSZ = OUT == 1 ? 4 : 2;         // bytes per stored element
    constexpr bool BIASF32 = OUT != 0;
    // 16-byte stores per lane and tile, ALWAYS issued (rows / columns outside the matrix go to an out-of-range offset): the
    // counted waits of the next tile's first K-step rely on the exact number
    constexpr int NST = 8 * (GLU ? 1 : 2) * (OUT == 0 ? 1 : 2) + (LNF == 2 ? 8 : 0);
    static_assert(LNF == 0 || (OUT == 0 && !CONV), "LayerNorm folding: bf16 GEMMs");
    static_assert(!SPL || LNF == 0, "shared-fragment split operands: not with a folded LayerNorm");
    static_assert(LNF != 1 || (RES == 0 && (GLU || ACT == 1)), "LNF 1: the w_1 (SiLU) / pointwise_conv1 (GLU) projections");
    static_assert(LNF != 2 || RES == 1, "LNF 2: a residual GEMM");
    static_assert(!(GLU && (ACT != 0 || RES != 0)), "GLU excludes an activation and a residual");
    static_assert(!(RES != 0 && ACT != 0), "the layer never pairs a residual with an activation");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // 2 x STEP + 8 KiB (bias slots) [+ 8 KiB csum slots + 16 KiB row statistics]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // ---- persistent tile loop: one block per CU walks the tiles of all batch entries; tile ids are dealt so that the blocks
    //      of one XCD (block ids equal mod 8) work on consecutive tiles -- the N-tiles of an M-tile share that XCD's L2
    const long nblk = (long)p.mtiles * p.ntiles;
    const long total = nblk * p.batch;
    const long per = total / 8;
    const int nt = SPL ? p.nsteps : p.K / PBK;    // K-steps per tile: even, >= 2 (checked on the host)

    // CONV: input offset of output position m (its tap (0, 0) pixel)
    auto pix_off = [&](long m) -> long {
        const long tf = (long)p.T2 * p.F2;
        const long bb = m / tf, rem = m - bb * tf;
        const long t2 = rem / p.F2, f2 = rem - t2 * p.F2;
        return (((bb * p.T1 + 2 * t2) * p.F1 + 2 * f2) * p.CiA) * 2;
    };

    // ---- a tile = its coordinates + the buffer descriptors of its operands, all wave-uniform (SGPRs).  The descriptors START
    //      at the tile's first row of A / first row of W, so the per-lane byte offsets of the LDS-DMA sources are the same for
    //      every tile (formed once, below); rows beyond the matrix lie beyond the descriptor's extent and read as zeros --
    //      loaded, never stored -- so nothing is clamped.  A descriptor of extent 0 (no next tile) turns every access into a
    //      no-op that still counts in vmcnt: the staging code is branch-free across the last tile.
    struct Tile {
        long m0;
        int n0, z, vr;
        __amdgpu_buffer_rsrc_t Ar, Wr;
    };
    auto make_tile = [&](long t, Tile &T) {
        const bool real = t < total;
        long tt = real ? t : 0;
        if (tt < per * 8) tt = (tt % 8) * per + tt / 8;
        T.z = (int)(tt / nblk);
        const long bid = tt % nblk;
        const int mt0 = (int)(bid / p.ntiles), nt0 = (int)(bid % p.ntiles);
        T.m0 = (long)mt0 * p.tm;
        T.n0 = nt0 * BN;
        T.vr = (int)min((long)p.tm, p.M - T.m0);     // valid rows of this tile
        if constexpr (CONV) {
            const long a_base = pix_off(T.m0);
            T.Ar = make_rsrc(reinterpret_cast<const unsigned char *>(p.A) + a_base, real ? p.in_bytes - a_base : 0);
            T.Wr = make_rsrc(p.W + (long)T.n0 * p.CiW, real ? ((long)8 * p.N + p.N - T.n0) * p.CiW * 2 : 0);
        } else {
            T.Ar = make_rsrc(p.A + T.z * p.sA + T.m0 * p.lda, real ? (p.M - T.m0) * p.lda * 2 : 0);   // (whole rows)
            T.Wr = make_rsrc(p.W + T.z * p.sW + (long)T.n0 * p.ldw, real ? ((long)(p.N - 1 - T.n0) * p.ldw + p.K) * 2 : 0);
        }
    };

    // ---- LDS-DMA sources: lane (sub = lane >> 3, pch = lane & 7) fills LDS chunk pch of row `sub` of its 8-row piece with
    //      source chunk pch ^ sub; unit row u = (wave * D + j) * 8 + sub.  Byte offsets relative to the tile's descriptors.
    unsigned a_off[2][DA], b_off[2][DB];
    // A offsets of one 64-row half (unit A_m0 / A_m1).  Tile-invariant for a plain GEMM (formed once); CONV: a row's pixel is
    // not affine in the row, so they are re-formed per tile.
    auto a_offsets = [&](const Tile &T, int h) {
        int sl = lane;
        asm volatile("" : "+v"(sl));
        const int sub = sl >> 3, pch = sl & 7;
#pragma unroll
        for (int j = 0; j < DA; ++j) {
            const int u = (wave * DA + j) * 8 + sub;                             // 0..127: wave-row group u >> 6, row u & 63
            const int row = (u >> 6) * 128 + h * 64 + (u & 63);
            // source chunk sc of the 128-byte row; SPL: chunks 0-3 = 32 columns of the hi plane, 4-7 = the same columns of lo
            const int sc = pch ^ sub;
            const unsigned coff = SPL ? (unsigned)(16 * (sc & 3) + (sc >> 2) * p.a_lo) : (unsigned)(16 * sc);
            a_off[h][j] = CONV ? (unsigned)(pix_off(T.m0 + row) - pix_off(T.m0)) + coff
                               : (unsigned)((long)row * p.lda * 2) + coff;
        }
    };
    auto b_offsets = [&]() {
        const int sub = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < DB; ++j) {
                const int u = (wave * DB + j) * 8 + sub;                         // 0..127: wave column group u >> 5, row u & 31
                const int r = u & 31;                                            // MFMA tile r >> 4, its n index r & 15
                const int pc = ((r & 15) >> 2) * 8 + (r >> 4) * 4 + (r & 3);     // -> column inside the half (see the head comment)
                const int sc = pch ^ sub;
                const unsigned coff = SPL ? (unsigned)(16 * (sc & 3) + (sc >> 2) * p.w_lo) : (unsigned)(16 * sc);
                b_off[h][j] = (unsigned)((long)((u >> 5) * 64 + h * 32 + pc) * (CONV ? p.CiW : p.ldw) * 2) + coff;
            }
    };
    // wave-uniform byte offset of K-step kt inside a row of A / W (all scalar arithmetic)
    auto a_soff = [&](int kt) -> unsigned {
        if constexpr (SPL && !CONV) {              // 32-column steps inside plane blocks [hi PB | lo PB]
            const int blk = kt >> p.pb_shift, c = kt - (blk << p.pb_shift);
            return (unsigned)(blk * 2 * p.pb_bytes + c * 64);
        } else if constexpr (SPL) {                // implicit GEMM: (tap, 32-channel slice); the pixel is [hi Ci | lo Ci]
            const int tap = (kt * p.inv_spt) >> 16, kc = kt - tap * p.spt;
            const int dt = (tap * 11) >> 5, df = tap - 3 * dt;
            return (unsigned)(((dt * p.F1 + df) * p.CiA) * 2 + kc * 64);
        } else if constexpr (!CONV) {
            const int seg = (kt >= p.nk1) + (kt >= 2 * p.nk1);           // hi, lo, hi (plain bf16 A: always 0)
            const int r = kt - seg * p.nk1;
            const int blk = r >> p.pb_shift, c = r - (blk << p.pb_shift);
            return (unsigned)(blk * 2 * p.pb_bytes + (seg == 1 ? p.pb_bytes : 0) + c * 128);
        } else {
            const int tap = (kt * p.inv_spt) >> 16, kc = kt - tap * p.spt;
            const int seg = (kc >= p.nk1) + (kc >= 2 * p.nk1), cc = kc - seg * p.nk1;
            const int dt = (tap * 11) >> 5, df = tap - 3 * dt;              // tap / 3, tap % 3 for tap < 9
            return (unsigned)(((dt * p.F1 + df) * p.CiA + (seg == 1 ? p.Ci : 0)) * 2 + cc * 128);
        }
    };
    auto w_soff = [&](int kt) -> unsigned {
        if constexpr (!CONV) return SPL ? kt * p.w_step : kt * 128;
        const int tap = (kt * p.inv_spt) >> 16, kc = kt - tap * p.spt;
        return (unsigned)(tap * p.N * p.CiW * 2 + kc * (SPL ? 64 : 128));
    };
    auto stage_a = [&](const Tile &T, int h, int buf, int kt) {
#ifdef PH_ABL_NODMA
        return;
#endif
#pragma unroll
        for (int j = 0; j < DA; ++j)
            dma16(T.Ar, a_off[h][j], a_soff(kt), lds + buf * STEP + (h ? OFF_A1 : OFF_A0) + (wave * DA + j) * 1024);
    };
    auto stage_b = [&](const Tile &T, int h, int buf, int kt) {
#ifdef PH_ABL_NODMA
        return;
#endif
#pragma unroll
        for (int j = 0; j < DB; ++j)
            dma16(T.Wr, b_off[h][j], w_soff(kt), lds + buf * STEP + (h ? OFF_B1 : OFF_B0) + (wave * DB + j) * 1024);
    };

    // ---- fragment read addresses: row fr of a 16-row tile, k chunk (ks * 4 + kq) ^ (row & 7); (row & 7) == (fr & 7)
    const int fr = lane & 15, kq = lane >> 4;
    const unsigned frag = fr * 128 + ((kq ^ (fr & 7)) * 16);                 // ks = 1 flips bit 6
    const unsigned la = wr * (64 * 128) + frag;                              // + i * 2048 inside a unit
    const unsigned lb = wc * (32 * 128) + frag;                              // + j * 2048

    bf16x8p af[4][2], bf0[TN][2], bf1[TN][2];
    auto read_a = [&](int h, int buf) {
#ifdef PH_ABL_NOREAD
        return;
#endif
        const unsigned char *base = lds + buf * STEP + (h ? OFF_A1 : OFF_A0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *reinterpret_cast<const bf16x8p *>(base + la + i * 2048);
            af[i][1] = *reinterpret_cast<const bf16x8p *>(base + (la ^ 64) + i * 2048);
        }
    };
    auto read_b = [&](int h, int buf, bf16x8p (&dst)[TN][2]) {
#ifdef PH_ABL_NOREAD
        return;
#endif
        const unsigned char *base = lds + buf * STEP + (h ? OFF_B1 : OFF_B0);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            dst[j][0] = *reinterpret_cast<const bf16x8p *>(base + lb + j * 2048);
            dst[j][1] = *reinterpret_cast<const bf16x8p *>(base + (lb ^ 64) + j * 2048);
        }
    };

    // ---- first tile: the six units the steady state would have in flight or landed when K-step 0 starts
    Tile cur, nxt;
    make_tile(blockIdx.x, cur);
    a_offsets(cur, 0);
    a_offsets(cur, 1);
    b_offsets();
    stage_a(cur, 0, 0, 0); stage_b(cur, 0, 0, 0); stage_b(cur, 1, 0, 0); stage_a(cur, 1, 0, 0);
    stage_a(cur, 0, 1, 1); stage_b(cur, 0, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    wait_vm<8>();
    __builtin_amdgcn_sched_barrier(0);
    bool first = true;                            // later tiles: NST stores + the bias DMA sit between the units of K-step 0 / 1 and the loop

    for (long t = blockIdx.x; t < total; t += gridDim.x) {
#ifdef PH_STAMPS
    const bool stamp_on = (wave == 0 || wave == 4) && lane == 0 && cur.z == 0;
    unsigned long long *st = p.stamps + ((size_t)t * 2 + (wave >> 2)) * 64;
    PH_STAMP(0);
#endif
    // the next tile of this block: its first six units ride in the staging slots the last two K-steps of THIS tile would
    // leave empty (their units belong to K-steps that do not exist) -- same regions, same phases, so the same hazards are
    // already covered, every phase keeps issuing exactly one unit, the waits stay at their steady-state counts and the
    // pipeline never drains between tiles.  (K / 64 is even: K-step 0 of the next tile lands in buffer 0 again.)
    make_tile(CONV ? total : t + gridDim.x, nxt);
    if constexpr (CONV) {
        // implicit GEMM: 72 K-steps per tile and A offsets that are not affine in the row (they are re-formed per tile) -- the
        // tile boundary is not worth the registers here: no next-tile units (the empty descriptors make the tail's staging a
        // no-op that still counts), an explicit prologue per tile instead
        if (!first) {
            a_offsets(cur, 0);
            a_offsets(cur, 1);
            stage_a(cur, 0, 0, 0); stage_b(cur, 0, 0, 0); stage_b(cur, 1, 0, 0); stage_a(cur, 1, 0, 0);
            stage_a(cur, 0, 1, 1); stage_b(cur, 0, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            wait_vm<8>();                         // (the previous tile's stores are older: done as well)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const bool after_stores = !CONV && !first;    // the previous tile's NST stores sit between this tile's first units and the loop
    const bool half_on[2] = {wr * 128 < cur.vr, wr * 128 + 64 < cur.vr};   // which 64-row halves of this wave multiply at all
    // The tile's bias row (256 columns: 512 B of bf16 or 1 KiB of fp32) comes by LDS-DMA too, one instruction per wave into
    // a slot of the wave's own behind the two K-step buffers: an LDS-DMA stays where it is written (a register load could be
    // moved to its use by the compiler, and the first K-step's counted waits rely on the exact sequence), it holds no
    // registers across the K loop, and the wave reads it back in its epilogue, long after the K loop's waits retired it.
    constexpr int NBL = LNF == 1 ? 4 : 1;     // bias; LNF 1: + csum + the tile's row statistics (2, below)
    {
        const long esz = BIASF32 ? 4 : 2;
        const __amdgpu_buffer_rsrc_t Br = make_rsrc(reinterpret_cast<const unsigned char *>(p.bias) + ((long)cur.z * p.sB + cur.n0) * esz,
                                                    p.bias ? (long)(p.N - cur.n0) * esz : 0);
        dma16(Br, lane * 16, 0, lds + 2 * STEP + wave * 1024);
        if constexpr (LNF == 1) {             // csum of the tile's 256 columns, fp32: the second slot of the wave
            const __amdgpu_buffer_rsrc_t Cr = make_rsrc(p.ln_csum + cur.n0, (long)(p.N - cur.n0) * 4);
            dma16(Cr, lane * 16, 0, lds + 2 * STEP + 8 * 1024 + wave * 1024);
        }
    }

    f32x4p acc[2][2][4][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[a][b][i][j] = f32x4p{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](int mi, int nj, const bf16x8p (&bqf)[TN][2]) {
        __builtin_amdgcn_s_setprio(1);
#ifdef PH_ABL_NOMFMA
        if (false) {
#else
        if (half_on[mi]) {   // wave-uniform, one branch per phase: a 64-row half beyond the tile's rows costs nothing
#endif
            if constexpr (SPL) {
                // fragments [.][0] = hi plane, [.][1] = lo plane of the same 32 columns: hi_w hi_a + hi_w lo_a + lo_w hi_a
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[mi][nj][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bqf[j][pr == 2], af[i][pr == 1], acc[mi][nj][i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)   // transposed product: D[n][m], a lane owns 4 consecutive n of one m
                            acc[mi][nj][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bqf[j][ks], af[i][ks], acc[mi][nj][i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // One K-step = four phases; every phase issues one unit and waits `vmcnt(8)`: everything but the four units issued last
    // has landed (2 LDS-DMA instructions per thread and unit).  The units a phase issues belong to the next K-step (phases 1,
    // 2) or to the one after (phases 3, 4) -- of this tile, or, in the last two K-steps, of the next one.  In the first
    // K-step of a tile that follows another one, the previous tile's NST output stores and this tile's bias DMA were issued
    // after the units of K-step 0 / 1 and ahead of everything the loop issues: the count grows by them.
// Ablation switches of the diagnostic build (tools/micro/gemm_ph_stamps.cpp; results are then WRONG, only the timing means
// something): PH_ABL_NOEND drops the closing barrier of a phase, PH_ABL_NOBAR the barrier after the counted wait, PH_ABL_NOMFMA
// the matrix instructions, PH_ABL_NODMA the global -> LDS units, PH_ABL_NOREAD the fragment reads.
#ifdef PH_ABL_NOBAR
#define PH_BAR1() do { } while (0)
#else
#define PH_BAR1() __builtin_amdgcn_s_barrier()
#endif
#ifdef PH_ABL_NOEND
#define PH_BAR2() do { } while (0)
#else
#define PH_BAR2() __builtin_amdgcn_s_barrier()
#endif
#define PH_SYNC()                                                                  \
    do {                                                                           \
        if (kt == 0) { if (after_stores) wait_vm<8 + NST + NBL>(); else wait_vm<8 + NBL>(); } \
        else wait_vm<8>();                                                         \
        PH_BAR1();                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)
#define PH_END()                                                                   \
    do {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
        PH_BAR2();                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)

    auto kstep = [&](auto parc, int kt) {
        constexpr int PAR = decltype(parc)::value;
        const int last = nt - 1 - kt;
        // phase 1: quadrant (rows 0-63, column half 0)
        read_b(0, PAR, bf0);
        read_a(0, PAR);
        if (last >= 1) stage_b(cur, 1, PAR ^ 1, kt + 1);
        else stage_b(nxt, 1, PAR ^ 1, 0);
        PH_SYNC();
        mma(0, 0, bf0);
        PH_END();
        // phase 2: (rows 0-63, column half 1)
        read_b(1, PAR, bf1);
        if (last >= 1) stage_a(cur, 1, PAR ^ 1, kt + 1);
        else stage_a(nxt, 1, PAR ^ 1, 0);
        PH_SYNC();
        mma(0, 1, bf1);
        PH_END();
        // phase 3: (rows 64-127, column half 1)
        read_a(1, PAR);
        if (last >= 2) stage_a(cur, 0, PAR, kt + 2);
        else stage_a(nxt, 0, PAR, 1 - last);
        PH_SYNC();
        mma(1, 1, bf1);
        PH_END();
        // phase 4: (rows 64-127, column half 0)
        if (last >= 2) stage_b(cur, 0, PAR, kt + 2);
        else stage_b(nxt, 0, PAR, 1 - last);
        PH_SYNC();
        mma(1, 0, bf0);
        PH_END();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    __builtin_amdgcn_s_barrier();
    if constexpr (LNF == 1) {
        // the tile's row statistics (256 rows x 8 pairs = 16 KiB) to LDS, two pieces per wave, behind the barrier every wave
        // passes after its previous epilogue (which read this region); read back in the epilogue, many K-step barriers later
        const __amdgpu_buffer_rsrc_t Sd = make_rsrc(p.ln_stats + (cur.z * p.M + cur.m0) * 16, (p.M - cur.m0) * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            dma16(Sd, (wave * 2 + j) * 1024 + lane * 16, 0, lds + 2 * STEP + 16 * 1024 + (wave * 2 + j) * 1024);
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind waves 0-3
    PH_STAMP(1);

    for (int kt = 0; kt < nt; kt += 2) {
        kstep(I0{}, kt);
        PH_STAMP(2 + (kt < 40 ? kt : 40));
        kstep(I1{}, kt + 1);
        PH_STAMP(3 + (kt < 40 ? kt : 40));
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // re-join: every wave has finished reading this tile's operands
    PH_STAMP(50);
#undef PH_SYNC
#undef PH_END

    // ---- epilogue: from the accumulators to memory, no LDS, no barrier ------------------------------------------
    int etid = tid;
    asm volatile("" : "+v"(etid));               // epilogue addresses are formed here, per tile, not hoisted out of the tile loop
    const int efr = etid & 15, ekq = (etid >> 4) & 3;
    auto unpack8 = [&](const u32x4p &q, float (&f)[8]) {
        f[0] = bf16_bits_to_f32(q.x & 0xffffu); f[1] = __uint_as_float(q.x & 0xffff0000u);
        f[2] = bf16_bits_to_f32(q.y & 0xffffu); f[3] = __uint_as_float(q.y & 0xffff0000u);
        f[4] = bf16_bits_to_f32(q.z & 0xffffu); f[5] = __uint_as_float(q.z & 0xffff0000u);
        f[6] = bf16_bits_to_f32(q.w & 0xffffu); f[7] = __uint_as_float(q.w & 0xffff0000u);
    };
    const int ncol = GLU ? p.N / 2 : p.N;
    const int nrt = (cur.vr + 15) >> 4;           // 16-row groups with a valid row
    // descriptors start at the tile's first row: offsets stay small whatever the size of the tensors
    const __amdgpu_buffer_rsrc_t Or = make_rsrc(reinterpret_cast<unsigned char *>(p.out) + (cur.z * p.sO + cur.m0 * p.ldo) * OSZ,
                                                ((p.M - 1 - cur.m0) * p.ldo + (OUT == 2 ? p.lo_off : 0) + ncol) * OSZ);
    constexpr int RSZ = RES == 2 ? 4 : 2;         // bytes per residual element
    __amdgpu_buffer_rsrc_t Rr = Or;
    if constexpr (RES != 0)
        Rr = make_rsrc(reinterpret_cast<const unsigned char *>(p.res) + (cur.z * p.sR + cur.m0 * p.ldr) * RSZ,
                       ((p.M - 1 - cur.m0) * p.ldr + p.N) * RSZ);
    const bool has_bias = p.bias != nullptr;
    auto bias8 = [&](int nj, float (&f)[8]) {     // the lane's 8 columns of column half nj, from the wave's bias slot
        const unsigned char *slot = lds + 2 * STEP + wave * 1024 + (wc * 64 + nj * 32 + ekq * 8) * (BIASF32 ? 4 : 2);
        if (!has_bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = 0.f;
        } else if constexpr (BIASF32) {
            const u32x4p q0 = *reinterpret_cast<const u32x4p *>(slot), q1 = *reinterpret_cast<const u32x4p *>(slot + 16);
            f[0] = __uint_as_float(q0.x); f[1] = __uint_as_float(q0.y); f[2] = __uint_as_float(q0.z); f[3] = __uint_as_float(q0.w);
            f[4] = __uint_as_float(q1.x); f[5] = __uint_as_float(q1.y); f[6] = __uint_as_float(q1.z); f[7] = __uint_as_float(q1.w);
        } else {
            unpack8(*reinterpret_cast<const u32x4p *>(slot), f);
        }
    };
    // store the lane's 8 consecutive values o[] of (row, col): bf16 one 16-byte store, fp32 two, hi | lo planes two
    auto store8 = [&](const float (&o)[8], bool live, int row, int col) {
        const bool ok = row < cur.vr && col < ncol;
        const unsigned off = ok ? (unsigned)(((long)row * p.ldo + col) * OSZ) : PH_OOB;
        if constexpr (OUT == 0) {
            u32x4p w{0u, 0u, 0u, 0u};
            if (live) w = u32x4p{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
            __builtin_amdgcn_raw_buffer_store_b128(w, Or, off, 0, PH_ST_AUX);
        } else if constexpr (OUT == 1) {
            u32x4p w0{0u, 0u, 0u, 0u}, w1{0u, 0u, 0u, 0u};
            if (live) {
                w0 = u32x4p{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
                w1 = u32x4p{__float_as_uint(o[4]), __float_as_uint(o[5]), __float_as_uint(o[6]), __float_as_uint(o[7])};
            }
            __builtin_amdgcn_raw_buffer_store_b128(w0, Or, off, 0, PH_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(w1, Or, ok ? off + 16 : PH_OOB, 0, PH_ST_AUX);
        } else {
            u32x4p wh{0u, 0u, 0u, 0u}, wl{0u, 0u, 0u, 0u};
            if (live) {
                float lo[8], hf[8];
                wh = u32x4p{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
                unpack8(wh, hf);
#pragma unroll
                for (int e = 0; e < 8; ++e) lo[e] = o[e] - hf[e];
                wl = u32x4p{pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(lo[4], lo[5]), pack_bf16(lo[6], lo[7])};
            }
            __builtin_amdgcn_raw_buffer_store_b128(wh, Or, off, 0, PH_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(wl, Or, ok ? off + (unsigned)(p.lo_off * 2) : PH_OOB, 0, PH_ST_AUX);
        }
    };
    // LNF: the rows' statistics.  8 partial (sum, sum of squares) pairs per row, [M][8] float2; the descriptor starts at the
    // tile's first row.
    __amdgpu_buffer_rsrc_t Sr = Or;
    if constexpr (LNF == 2) Sr = make_rsrc(p.ln_stats + (cur.z * p.M + cur.m0) * 16, (p.M - cur.m0) * 64);
    // consumer: a row's eight pairs from the LDS copy (no cross-lane traffic) -> rstd, -mean * rstd.  Software-pipelined over
    // the row groups: a group's pairs are read while the previous group's values are computed (a group is one scheduling
    // region: read and used inside it, every group would expose an LDS round trip and the rsqrt chain)
    auto row_read = [&](int g, u32x4p (&v)[4]) {
        const unsigned char *rp = lds + 2 * STEP + 16 * 1024 + (wr * 128 + g * 16 + efr) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const u32x4p *>(rp + q * 16);
    };
    auto row_norm = [&](const u32x4p (&v)[4], float &rstd, float &nm) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s1 += __uint_as_float(v[q].x) + __uint_as_float(v[q].z);
            s2 += __uint_as_float(v[q].y) + __uint_as_float(v[q].w);
        }
        const float mean = s1 * p.ln_inv_c;
        const float var = fmaxf(fmaf(-mean, mean, s2 * p.ln_inv_c), 0.f);
        rstd = rsqrtf(var + p.ln_eps);
        nm = -mean * rstd;
    };
    auto csum8 = [&](int nj, float (&f)[8]) {     // the lane's 8 columns of column half nj, from the wave's csum slot
        const unsigned char *slot = lds + 2 * STEP + 8 * 1024 + wave * 1024 + (wc * 64 + nj * 32 + ekq * 8) * 4;
        const u32x4p q0 = *reinterpret_cast<const u32x4p *>(slot), q1 = *reinterpret_cast<const u32x4p *>(slot + 16);
        f[0] = __uint_as_float(q0.x); f[1] = __uint_as_float(q0.y); f[2] = __uint_as_float(q0.z); f[3] = __uint_as_float(q0.w);
        f[4] = __uint_as_float(q1.x); f[5] = __uint_as_float(q1.y); f[6] = __uint_as_float(q1.z); f[7] = __uint_as_float(q1.w);
    };
    // The residual comes straight to registers, 64-row half by half: all loads of a half are issued, then consumed (one
    // exposed round trip per half; rows / columns outside the matrix read zeros).  bf16: 8 loads, fp32: 16 per half.
    constexpr int RLD = RES == 2 ? 2 : 1;         // 16-byte loads per (row group, column half)
    // LNF 1: the column sums and the bias of the lane's 16 columns are the same for every row group: held in registers
    float lcs[LNF == 1 ? 2 : 1][8], lbv[LNF == 1 ? 2 : 1][8], nx_rstd = 1.f, nx_nm = 0.f;
    if constexpr (LNF == 1) {
        csum8(0, lcs[0]); csum8(LNF == 1 ? 1 : 0, lcs[LNF == 1 ? 1 : 0]);
        bias8(0, lbv[0]); bias8(LNF == 1 ? 1 : 0, lbv[LNF == 1 ? 1 : 0]);
        u32x4p v0[4];
        row_read(0, v0);
        row_norm(v0, nx_rstd, nx_nm);
    }
    float gst1[LNF == 2 ? 8 : 1], gst2[LNF == 2 ? 8 : 1];      // LNF 2: this lane's share of each row group's (sum, sum of squares)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        u32x4p rr[RES != 0 ? 4 : 1][2][RLD];
        if constexpr (RES != 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) {
                    const int row = wr * 128 + mi * 64 + i * 16 + efr, col = cur.n0 + wc * 64 + nj * 32 + ekq * 8;
                    const bool ok = row < cur.vr && col < p.N;
                    const unsigned off = ok ? (unsigned)(((long)row * p.ldr + col) * RSZ) : PH_OOB;
#pragma unroll
                    for (int q = 0; q < RLD; ++q) rr[i][nj][q] = __builtin_amdgcn_raw_buffer_load_b128(Rr, ok ? off + 16 * q : PH_OOB, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_barrier(0);    // one row group at a time: keeps the epilogue's registers few
            const int g = mi * 4 + i;
            const bool live = wr * 8 + g < nrt;   // wave-uniform: a row group beyond the tile's rows computes nothing
            const int row = wr * 128 + g * 16 + efr;
            const float ln_rstd = nx_rstd, ln_nm = nx_nm;
            u32x4p vn[4];
            if constexpr (LNF == 1) row_read(g < 7 ? g + 1 : 7, vn);      // the next group's pairs: in flight under this group's values
            if constexpr (GLU) {
                float o[8];
                if (live) {
                    float bv[8], bg[8];
                    if constexpr (LNF == 1) {     // b' - mean rstd csum, then rstd acc on top
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            bv[e] = fmaf(ln_nm, lcs[0][e], lbv[0][e]);
                            bg[e] = fmaf(ln_nm, lcs[LNF == 1 ? 1 : 0][e], lbv[LNF == 1 ? 1 : 0][e]);
                        }
                    } else {
                        bias8(0, bv);
                        bias8(1, bg);
                    }
                    const float sc = LNF == 1 ? ln_rstd : p.alpha;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = fmaf(acc[mi][0][i][j][e], sc, bv[j * 4 + e]);
                            const float b = fmaf(acc[mi][1][i][j][e], sc, bg[j * 4 + e]);
                            o[j * 4 + e] = a * __builtin_amdgcn_rcpf(1.f + __expf(-b));
                        }
                }
                store8(o, live, row, cur.n0 / 2 + wc * 32 + ekq * 8);
            } else {
                float st1 = 0.f, st2 = 0.f;   // LNF 2: this lane's share of the row's sum / sum of squares
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) {
                    const int col = cur.n0 + wc * 64 + nj * 32 + ekq * 8;
                    float o[8];
                    if (live) {
                        float bv[8];
                        if constexpr (LNF == 1) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) bv[e] = fmaf(ln_nm, lcs[LNF == 1 ? nj : 0][e], lbv[LNF == 1 ? nj : 0][e]);
                        } else {
                            bias8(nj, bv);
                        }
                        const float sc = LNF == 1 ? ln_rstd : p.alpha;
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                o[j * 4 + e] = act_apply<ACT>(fmaf(acc[mi][nj][i][j][e], sc, bv[j * 4 + e]));
                        if constexpr (RES == 1) {
                            float rv[8];
                            unpack8(rr[i][nj][0], rv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] += rv[e];
                        } else if constexpr (RES == 2) {
                            o[0] += __uint_as_float(rr[i][nj][0].x); o[1] += __uint_as_float(rr[i][nj][0].y);
                            o[2] += __uint_as_float(rr[i][nj][0].z); o[3] += __uint_as_float(rr[i][nj][0].w);
                            o[4] += __uint_as_float(rr[i][nj][RLD - 1].x); o[5] += __uint_as_float(rr[i][nj][RLD - 1].y);
                            o[6] += __uint_as_float(rr[i][nj][RLD - 1].z); o[7] += __uint_as_float(rr[i][nj][RLD - 1].w);
                        }
                        if constexpr (LNF == 2) {
                            // statistics of the row AS STORED (rounded to bf16, the very conversion store8 performs): the
                            // consumer multiplies the stored rows, and mean / rstd must be those of what it multiplies
                            float q[8];
                            unpack8(u32x4p{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])}, q);
#pragma unroll
                            for (int e = 0; e < 8; ++e) { st1 += q[e]; st2 = fmaf(q[e], q[e], st2); }
                        }
                    }
                    store8(o, live, row, col);
                }
                if constexpr (LNF == 2) { gst1[LNF == 2 ? g : 0] = live ? st1 : 0.f; gst2[LNF == 2 ? g : 0] = live ? st2 : 0.f; }
            }
            if constexpr (LNF == 1) row_norm(vn, nx_rstd, nx_nm);
        }
    }
    if constexpr (LNF == 2) {
        // the wave's 64 columns of each row: the four lanes of a row add up (all groups' exchanges back to back: one exposed
        // round trip, not one per group), lane kq = 0 writes the pair
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) { gst1[g] += __shfl_xor(gst1[g], 16, 64); gst2[g] += __shfl_xor(gst2[g], 16, 64); }
#pragma unroll
        for (int g = 0; g < 8; ++g) { gst1[g] += __shfl_xor(gst1[g], 32, 64); gst2[g] += __shfl_xor(gst2[g], 32, 64); }
        typedef unsigned u32x2p __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int row = wr * 128 + g * 16 + efr;
            const bool ok = ekq == 0 && row < cur.vr;
            const unsigned off = ok ? (unsigned)(row * 64 + ((cur.n0 >> 8) * 4 + wc) * 8) : PH_OOB;
            __builtin_amdgcn_raw_buffer_store_b64(u32x2p{__float_as_uint(gst1[g]), __float_as_uint(gst2[g])}, Sr, off, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    first = false;
    if constexpr (CONV) make_tile(t + gridDim.x, cur);    // (nxt is the empty tile here)
    else cur = nxt;
    PH_STAMP(53);
    }   // tile loop
    PH_WAIT(0);                                   // the (empty) prefetch of a tile that does not exist: nothing in flight at exit
}

template <bool GLU, int ACT, int RES, int OUT, bool CONV = false, int LNF = 0, bool SPL = false>
int launch_ph(const PhParams &p, int batch, hipStream_t s) {
    // two K-steps (128 KiB) + a bias slot per wave
    // (LNF 1: + a csum slot per wave + the tile's row statistics = all 160 KiB)
    constexpr size_t lds = 2 * (2 * 128 * 128 + 2 * (PBN / 2) * 128) + 8 * 1024 + (LNF == 1 ? 8 * 1024 + 16 * 1024 : 0);
    auto kern = gemm_ph_kernel<GLU, ACT, RES, OUT, CONV, LNF, SPL>;
    static bool attr_set[64];                     // per device; a racing first call sets the same attribute twice
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PAFC_ERR_LAUNCH;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PAFC_ERR_LAUNCH;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int cus = device_cus();
    PhParams q = p;
    q.batch = batch;
    const long total = (long)p.mtiles * p.ntiles * batch;
    const long grid = total < cus ? total : cus;              // one 512-thread block per CU (128 KiB of LDS each)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, q);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

// PAFC_SPLIT_WALK=hilohi: split-operand problems take the round-3 K walk (hi, lo, hi against [hi | hi | lo]) instead of the
// shared-fragment form (A/B runs; read once)
int g_split_walk = -1;                      // -1: not read yet; 1: shared fragments; 0: hi, lo, hi  (tools/micro set it directly)
bool split_shared_fragments() {
    if (g_split_walk < 0) {
        const char *e = getenv("PAFC_SPLIT_WALK");
        g_split_walk = !(e && std::strcmp(e, "hilohi") == 0);
    }
    return g_split_walk != 0;
}

}  // namespace
}  // namespace pafc

// The general entry point of the phase-pipelined kernel (include/pafc_encoder_ops.h).
//   a_split != 0: A holds an fp32 operand as two bf16 planes [hi | lo] (M x 2K, lda >= 2K) and W the matching
//                 [hi_w | hi_w | lo_w] (N x 3K, ldw >= 3K): the kernel walks 3K columns, A's wrapping after 2K.
//   out_kind: 0 bf16, 1 fp32, 2 fp32 as bf16 planes hi | lo (lo at column offset lo_off of the same row; ldo in bf16 elements).
//   res_kind: 0 none, 1 bf16, 2 fp32 (with res_kind 2 / out_kind != 0 the bias is fp32).
extern "C" int pafc_gemm_ph_ex(long M, int N, int K, int batch, const void *A, long lda, long strideA, int a_split, const void *W,
                               long ldw, long strideW, const void *bias, long strideBias, const void *residual, int res_kind,
                               long ldr, long strideR, void *out, int out_kind, long ldo, long lo_off, long strideO, float alpha,
                               int act, int tile_m, pafc_stream_t stream) {
    return pafc_gemm_ph_ex2(M, N, K, batch, A, lda, strideA, a_split, 0, W, ldw, strideW, bias, strideBias, residual, res_kind, ldr,
                            strideR, out, out_kind, ldo, lo_off, strideO, alpha, act, tile_m, stream);
}

//   a_plane_block (with a_split): 0 = the row is [hi K | lo K]; else the planes alternate in blocks of that many columns,
//                 [hi PB | lo PB] [hi PB | lo PB] ... (PB = 64 << n, K % PB == 0).
extern "C" int pafc_gemm_ph_ex2(long M, int N, int K, int batch, const void *A, long lda, long strideA, int a_split,
                                int a_plane_block, const void *W, long ldw, long strideW, const void *bias, long strideBias,
                                const void *residual, int res_kind, long ldr, long strideR, void *out, int out_kind, long ldo,
                                long lo_off, long strideO, float alpha, int act, int tile_m, pafc_stream_t stream) {
    if (!A || !W || !out) return PAFC_ERR_NULL_POINTER;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535) return PAFC_ERR_BAD_DIMS;
    if (act < 0 || act > 4 || out_kind < 0 || out_kind > 2 || res_kind < 0 || res_kind > 2) return PAFC_ERR_UNSUPPORTED;
    if (!residual) res_kind = 0;
    else if (res_kind == 0) return PAFC_ERR_BAD_DIMS;
    const bool glu = act == 4;
    if (tile_m < 64 || tile_m > 256 || tile_m % 64) return PAFC_ERR_UNSUPPORTED;
    if (N % 8 || K % 128) return PAFC_ERR_UNSUPPORTED;          // an even number of 64-deep K-steps (see the kernel's tile loop)
    if (glu && (N % 256 || residual)) return PAFC_ERR_UNSUPPORTED;
    if (residual && act != 0) return PAFC_ERR_UNSUPPORTED;      // the layer never pairs a residual with an activation
    if (res_kind == 1 && out_kind != 0) return PAFC_ERR_UNSUPPORTED;   // a bf16 residual stream has a bf16 output
    if (res_kind == 2 && out_kind != 1) return PAFC_ERR_UNSUPPORTED;   // an fp32 residual stream stays fp32
    const int Kw = a_split ? 3 * K : K;                         // columns the K loop walks
    const int No = glu ? N / 2 : N;
    if (lda < (a_split ? 2 * K : K) || ldw < Kw || (residual && ldr < N)) return PAFC_ERR_BAD_DIMS;
    if (out_kind == 2 ? (lo_off < No || ldo < lo_off + No) : ldo < No) return PAFC_ERR_BAD_DIMS;
    if ((lda | ldw | strideA | strideW) % 8 || (ldo | strideO | lo_off) % 8 || (residual && ((ldr | strideR) % 8))) return PAFC_ERR_ALIGNMENT;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)bias) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    // 31-bit byte extents inside one batch entry of A / W (buffer descriptors that start at the tensor), and inside one
    // 256-row tile of the output / residual (descriptors that start at the tile)
    if ((double)M * lda * 2 >= 2.0e9 || (double)N * ldw * 2 >= 2.0e9) return PAFC_ERR_UNSUPPORTED;
    if ((double)256 * ldo * 4 >= 2.0e9 || (residual && (double)256 * ldr * 4 >= 2.0e9)) return PAFC_ERR_UNSUPPORTED;
    pafc::PhParams p{};
    p.A = (const pafc::bf16_t *)A; p.W = (const pafc::bf16_t *)W; p.bias = bias; p.res = residual; p.out = out;
    p.M = M; p.N = N; p.K = Kw;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr;
    p.sA = strideA; p.sW = strideW; p.sO = strideO; p.sB = strideBias; p.sR = strideR;
    p.alpha = alpha;
    p.nk1 = a_split ? K / 64 : INT_MAX / 4;
    p.pb_shift = 31;
    p.pb_bytes = (long)K * 2;
    if (a_split && a_plane_block) {
        int sh = 0;
        while ((64 << sh) < a_plane_block) ++sh;
        if ((64 << sh) != a_plane_block || K % a_plane_block) return PAFC_ERR_UNSUPPORTED;
        p.pb_shift = sh;
        p.pb_bytes = (long)a_plane_block * 2;
    }
    p.lo_off = lo_off;
    p.tm = tile_m;
    p.mtiles = (int)((M + tile_m - 1) / tile_m);
    p.ntiles = (N + 255) / 256;
    if ((long)p.mtiles * p.ntiles > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    if (a_split && out_kind != 0 && pafc::split_shared_fragments() && (double)2 * K * 2 + 64 < 2.0e9) {
        // shared-fragment form: K-steps of 32 columns of both planes; pb_shift counts 32-column steps per plane block
        p.nsteps = K / 32;
        p.a_lo = (int)p.pb_bytes;                 // [hi PB | lo PB]: lo follows hi inside a block (one block: PB = K)
        p.w_lo = 2 * K * 2;                       // W' = [hi | hi | lo]: the lo plane starts 2 K columns in
        p.w_step = 64;
        p.pb_shift = p.pb_shift >= 30 ? 30 : p.pb_shift + 1;
        if (out_kind == 1) {
            if (glu) return pafc::launch_ph<true, 0, 0, 1, false, 0, true>(p, batch, s);
            if (res_kind == 2) return pafc::launch_ph<false, 0, 2, 1, false, 0, true>(p, batch, s);
            if (act == 0) return pafc::launch_ph<false, 0, 0, 1, false, 0, true>(p, batch, s);
            return PAFC_ERR_UNSUPPORTED;
        }
        if (glu || residual) return PAFC_ERR_UNSUPPORTED;
        if (act == 1) return pafc::launch_ph<false, 1, 0, 2, false, 0, true>(p, batch, s);
        if (act == 0) return pafc::launch_ph<false, 0, 0, 2, false, 0, true>(p, batch, s);
        return PAFC_ERR_UNSUPPORTED;
    }
    if (out_kind == 0) {
        if (glu) return pafc::launch_ph<true, 0, 0, 0>(p, batch, s);
        if (res_kind == 1) return pafc::launch_ph<false, 0, 1, 0>(p, batch, s);
        switch (act) {
            case 1: return pafc::launch_ph<false, 1, 0, 0>(p, batch, s);
            case 2: return pafc::launch_ph<false, 2, 0, 0>(p, batch, s);
            case 3: return pafc::launch_ph<false, 3, 0, 0>(p, batch, s);
            default: return pafc::launch_ph<false, 0, 0, 0>(p, batch, s);
        }
    }
    if (out_kind == 1) {
        if (glu) return pafc::launch_ph<true, 0, 0, 1>(p, batch, s);
        if (res_kind == 2) return pafc::launch_ph<false, 0, 2, 1>(p, batch, s);
        if (act == 0) return pafc::launch_ph<false, 0, 0, 1>(p, batch, s);
        return PAFC_ERR_UNSUPPORTED;
    }
    if (glu || residual) return PAFC_ERR_UNSUPPORTED;           // planes: the hidden tensor of an FFN, or a plain projection
    if (act == 1) return pafc::launch_ph<false, 1, 0, 2>(p, batch, s);
    if (act == 0) return pafc::launch_ph<false, 0, 0, 2>(p, batch, s);
    return PAFC_ERR_UNSUPPORTED;
}

// The bf16 projections either side of a pre-norm LayerNorm with the norm folded in (include/pafc_encoder_ops.h).
//   ln_mode 2 (producer): out = alpha A W^T + bias + residual as pafc_gemm_bf16_ph, N == 512, and stats[m][8] (float2) receives
//                         each row's (sum, sum of squares) per 64-column slice of the fp32 result.
//   ln_mode 1 (consumer): out = act(rstd_m (A W'^T)[m][n] - rstd_m mean_m csum[n] + bias[n]), act SiLU (1) or GLU (4), with
//                         mean / rstd of row m from stats (C = `ln_c` channels, eps `ln_eps`); A is the un-normalised stream.
extern "C" int pafc_gemm_bf16_ph_ln(long M, int N, int K, const void *A, long lda, const void *W, long ldw, const void *bias,
                                    const void *residual, long ldr, void *out, long ldo, float alpha, int act, int ln_mode,
                                    float *stats, const float *csum, int ln_c, float ln_eps, int tile_m, pafc_stream_t stream) {
    if (!A || !W || !out || !stats) return PAFC_ERR_NULL_POINTER;
    if (M <= 0 || N <= 0 || K <= 0 || ln_c <= 0) return PAFC_ERR_BAD_DIMS;
    if (tile_m < 64 || tile_m > 256 || tile_m % 64) return PAFC_ERR_UNSUPPORTED;
    if (N % 8 || K % 128) return PAFC_ERR_UNSUPPORTED;
    const bool glu = act == 4;
    if (ln_mode == 1) {
        if (!csum || residual || (act != 1 && act != 4) || alpha != 1.f || K != ln_c) return PAFC_ERR_UNSUPPORTED;
        if (glu && N % 256) return PAFC_ERR_UNSUPPORTED;
    } else if (ln_mode == 2) {
        if (!residual || act != 0 || N != 512 || ln_c != N) return PAFC_ERR_UNSUPPORTED;
    } else {
        return PAFC_ERR_UNSUPPORTED;
    }
    const int No = glu ? N / 2 : N;
    if (lda < K || ldw < K || ldo < No || (residual && ldr < N)) return PAFC_ERR_BAD_DIMS;
    if ((lda | ldw | ldo) % 8 || (residual && ldr % 8)) return PAFC_ERR_ALIGNMENT;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)bias | (uintptr_t)stats | (uintptr_t)csum) & 15) != 0)
        return PAFC_ERR_ALIGNMENT;
    if ((double)M * lda * 2 >= 2.0e9 || (double)N * ldw * 2 >= 2.0e9 || (double)256 * ldo * 4 >= 2.0e9 ||
        (residual && (double)256 * ldr * 4 >= 2.0e9))
        return PAFC_ERR_UNSUPPORTED;
    pafc::PhParams p{};
    p.A = (const pafc::bf16_t *)A; p.W = (const pafc::bf16_t *)W; p.bias = bias; p.res = residual; p.out = out;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr;
    p.alpha = alpha;
    p.nk1 = INT_MAX / 4; p.pb_shift = 31; p.pb_bytes = (long)K * 2;
    p.ln_stats = stats; p.ln_csum = csum; p.ln_eps = ln_eps; p.ln_inv_c = 1.f / (float)ln_c;
    p.tm = tile_m;
    p.mtiles = (int)((M + tile_m - 1) / tile_m);
    p.ntiles = (N + 255) / 256;
    if ((long)p.mtiles * p.ntiles > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    if (ln_mode == 2) return pafc::launch_ph<false, 0, 1, 0, false, 2>(p, 1, s);
    if (glu) return pafc::launch_ph<true, 0, 0, 0, false, 1>(p, 1, s);
    return pafc::launch_ph<false, 1, 0, 0, false, 1>(p, 1, s);
}

// Same contract as pafc_gemm_bf16 (which calls this for the shapes it suits); exported for A/B measurements.
extern "C" int pafc_gemm_bf16_ph(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W,
                                 long ldw, long strideW, const void *bias, long strideBias, const void *residual, long ldr,
                                 long strideR, void *out, long ldo, long strideO, float alpha, int act, int tile_n,
                                 int tile_m, pafc_stream_t stream) {
    if (tile_n != 256) return PAFC_ERR_UNSUPPORTED;
    return pafc_gemm_ph_ex(M, N, K, batch, A, lda, strideA, 0, W, ldw, strideW, bias, strideBias, residual, residual ? 1 : 0, ldr,
                           strideR, out, 0, ldo, 0, strideO, alpha, act, tile_m, stream);
}

// The second subsampling convolution, Conv2d(Ci, Co, 3, stride 2) + bias (+ ReLU) on NHWC (wenet/transformer/
// subsampling.py:187-192), as an implicit GEMM on the phase-pipelined kernel.  split = 0: bf16 in, bf16 weights (9, Co, Ci),
// bf16 bias, bf16 out -- same contract as pafc_conv3x3s2_nhwc_bf16, which dispatches here when the problem fills the chip
// with 256-wide tiles.  split = 1 (fp32 models): in = planes per pixel [hi Ci | lo Ci] of the fp32 image (what
// pafc_conv3x3s2_c1_f32split writes with a pixel stride of 2 Ci), weights (9, Co, 3 Ci) = [hi | hi | lo] per tap, fp32 bias,
// out = planes per output position [hi Co | lo Co] of the fp32 result -- read as it lies by pafc_gemm_ph_ex2(a_split = 1,
// a_plane_block = Co) for Linear(F' Co, odim).  PAFC_ERR_UNSUPPORTED = take the other kernel.
static int conv_ph_launch(int B, int T1, int F1, int Ci, int Co, const void *in, const void *w, const void *bias, void *out,
                          int relu, int tile_m, int split, pafc_stream_t stream) {
    if (!in || !w || !out) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T1 < 3 || F1 < 3 || Ci <= 0 || Co <= 0) return PAFC_ERR_BAD_DIMS;
    // 64-channel K-steps and an even number of them in all (9 Ci / 64 even <=> Ci % 128 == 0)
    if (Ci % 128 || Co % 8) return PAFC_ERR_UNSUPPORTED;
    if (tile_m != 256 && tile_m != 192 && tile_m != 128) return PAFC_ERR_UNSUPPORTED;
    if ((((uintptr_t)in | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
    const long M = (long)B * T2 * F2;
    pafc::PhParams p{};
    p.A = (const pafc::bf16_t *)in; p.W = (const pafc::bf16_t *)w; p.bias = bias; p.out = out;
    p.Ci = Ci; p.CiA = split ? 2 * Ci : Ci; p.CiW = split ? 3 * Ci : Ci;
    if ((double)9 * Co * p.CiW * 2 >= 2.0e9 || (double)3 * F1 * p.CiA * 2 >= 2.0e9) return PAFC_ERR_UNSUPPORTED;
    p.M = M; p.N = Co; p.K = 9 * p.CiW; p.lda = p.CiA; p.ldw = p.CiW; p.alpha = 1.f;
    p.ldo = split ? 2 * Co : Co; p.lo_off = split ? Co : 0;
    p.nk1 = Ci / 64;                   // per tap: hi (, lo, hi)
    p.pb_shift = 31; p.pb_bytes = 0;
    p.spt = p.CiW / 64;
    p.inv_spt = (65536 + p.spt - 1) / p.spt;
    for (int kt = 0; kt < p.K / 64 + 2; ++kt)
        if (((kt * p.inv_spt) >> 16) != kt / p.spt) return PAFC_ERR_UNSUPPORTED;     // (never for the sizes above; checked, not assumed)
    p.T1 = T1; p.F1 = F1; p.T2 = T2; p.F2 = F2;
    p.in_bytes = (long)B * T1 * F1 * p.CiA * 2;
    p.tm = tile_m;
    p.mtiles = (int)((M + tile_m - 1) / tile_m);
    p.ntiles = (Co + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (split && pafc::split_shared_fragments()) {
        // shared-fragment form: K-step = (tap, 32-channel slice of both planes); the pixel is [hi Ci | lo Ci], a weight row
        // of a tap [hi Ci | hi Ci | lo Ci]
        p.spt = Ci / 32;
        p.inv_spt = (65536 + p.spt - 1) / p.spt;
        p.nsteps = 9 * p.spt;
        for (int kt = 0; kt < p.nsteps + 2; ++kt)
            if (((kt * p.inv_spt) >> 16) != kt / p.spt) return PAFC_ERR_UNSUPPORTED;
        p.a_lo = Ci * 2;
        p.w_lo = 2 * Ci * 2;
        return relu ? pafc::launch_ph<false, 3, 0, 2, true, 0, true>(p, 1, s) : pafc::launch_ph<false, 0, 0, 2, true, 0, true>(p, 1, s);
    }
    if (split) return relu ? pafc::launch_ph<false, 3, 0, 2, true>(p, 1, s) : pafc::launch_ph<false, 0, 0, 2, true>(p, 1, s);
    return relu ? pafc::launch_ph<false, 3, 0, 0, true>(p, 1, s) : pafc::launch_ph<false, 0, 0, 0, true>(p, 1, s);
}

extern "C" int pafc_conv3x3s2_nhwc_bf16_ph(int B, int T1, int F1, int Ci, int Co, const void *in, const void *w_tap_co_ci,
                                           const void *bias, void *out, int relu, int tile_m, pafc_stream_t stream) {
    return conv_ph_launch(B, T1, F1, Ci, Co, in, w_tap_co_ci, bias, out, relu, tile_m, 0, stream);
}

extern "C" int pafc_conv3x3s2_nhwc_split_ph(int B, int T1, int F1, int Ci, int Co, const void *in_planes, const void *w3_tap_co_3ci,
                                            const float *bias, void *out_planes, int relu, int tile_m, pafc_stream_t stream) {
    return conv_ph_launch(B, T1, F1, Ci, Co, in_planes, w3_tap_co_3ci, bias, out_planes, relu, tile_m, 1, stream);
}
