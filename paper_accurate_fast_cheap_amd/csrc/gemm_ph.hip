// The long-form projections as a phase-pipelined bf16 GEMM (gfx950): 256 x BN tiles, 8 waves, K-step 64.
// C ABI: include/pafc_encoder_ops.h: pafc_gemm_bf16 dispatches here for large problems (see gemm_bf16.hip for the
// 128 x 128 kernel that keeps the small and oddly shaped ones).
//
//   out[z][m][n] = act(alpha * sum_k A[z][m][k] * W[z][n][k] + bias[z][n] + residual[z][m][n])
// nn.Linear (weight (N, K), K contiguous) with bias, SiLU / tanh / ReLU / GLU, ff_scale and the residual add applied to
// the fp32 accumulator before the single rounding: the FFN, 1x1-conv, r/k/v and output projections of
// ConformerEncoderLayer, ConvolutionModule and RWKV_Tmix_x060c (wenet/transformer/positionwise_feed_forward.py:47-55,
// convolution.py:118-141, rwkv_v6/src/model.py:286-324, encoder_layer.py:201-259) at the 30-minute shape (44 998 rows),
// where the library's kernels ran at 25-39 % of the matrix peak (round-1 review).
//
// Why this structure.  With one 512-thread block per CU every SIMD hosts two waves.  The two waves of a SIMD run the SAME
// program one barrier apart (waves 4-7 take one extra barrier at the start): while one multiplies (16 MFMAs = 256 cycles of
// the SIMD's matrix pipe) the other reads its next operand fragments from LDS and issues the block's next LDS-DMA piece,
// then they swap.  A K-step of a wave is four such phases, one per quadrant of its 128 x (BN / 4) output.
//
// LDS holds two K-steps (2 x 64 KiB at BN = 256), each cut into four UNITS by the phase in which they are read:
//   A_m0 (the first 64 rows of every wave's 128; read in phase 1)   B_n0 (first half of every wave's columns; phase 1)
//   B_n1 (second half of the columns; phase 2)                       A_m1 (the other 64 rows; phase 3)
// A unit is dead two phases after its read phase, so it is re-filled for the K-step after next right then: every phase issues
// exactly one unit (2 LDS-DMA instructions per thread at BN = 256), four to six phases (1 000+ cycles) ahead of its use, and
// waits only `vmcnt(8)` -- the unit issued four phases ago -- never vmcnt(0).  Barriers are raw s_barrier (a
// __syncthreads() would drain the DMA queue); the XOR swizzle of the 16-byte chunks sits on the SOURCE address, the LDS
// image is lane-linear, and the fragment reads apply the same XOR (conflict-free ds_read_b128).
//
// The product is formed transposed (W fragment as the first MFMA operand): a lane then owns 4 CONSECUTIVE output columns of
// one row, so the epilogue packs 8-byte pieces into a swizzled [256][BN] bf16 image in LDS (in place over the residual
// tile, which arrives by LDS-DMA) and the tile leaves as whole 512-byte rows.
#include <type_traits>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));

struct PhParams {
    const bf16_t *A, *W, *bias, *res;
    bf16_t *out;
    long M;
    int N, K;
    long lda, ldw, ldo, ldr;          // row strides (elements)
    long sA, sW, sO, sB, sR;          // batch strides (elements); sB = 0 shares the bias
    float alpha;
    int mtiles, ntiles;
    int tm;                           // rows per tile actually used: 256, 192, 128 or 64 (balances the grid over the CUs)
    int batch;
    // implicit-GEMM mode (CONV): A is an NHWC image (B, T1, F1, Ci), row m = output position (b, t2, f2) of a 3 x 3 stride-2
    // convolution, K = 9 taps x Ci; W is (9, N, Ci) tap-major.  K-step kt = (tap, 64-channel slice).
    int T1, F1, T2, F2, Ci, kshift;   // kshift = log2(Ci / 64)
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

#define PH_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int PBM = 256, PBK = 64;

// EPI: 0 plain, 1 residual, 2 GLU (weight rows in blocks of 64 = 32 values + the 32 gates of the same channels: a wave's
// 64 columns are one block, its column half 0 the values and half 1 the gates; the output has N / 2 columns)
// ACT: 0 none, 1 SiLU, 2 tanh, 3 ReLU (EPI 0 only).  CONV: the second subsampling convolution as an implicit GEMM.
template <int BN, int EPI, int ACT, bool CONV = false>
__global__ __launch_bounds__(512, 2) void gemm_ph_kernel(const PhParams p) {
    constexpr int UA = 128 * 128;                 // bytes of an A unit: 128 rows x 64 k
    constexpr int UB = (BN / 2) * 128;            // bytes of a B unit
    constexpr int STEP = 2 * UA + 2 * UB;         // one K-step in LDS
    constexpr int OFF_A0 = 0, OFF_A1 = UA, OFF_B0 = 2 * UA, OFF_B1 = 2 * UA + UB;
    constexpr int WN = BN / 4;                    // columns per wave
    constexpr int TN = WN / 32;                   // 16-column tiles per column half (2 at BN = 256, 1 at BN = 128)
    constexpr int DA = UA / 8192;                 // LDS-DMA instructions per thread for an A unit (2)
    constexpr int DB = UB / 8192;                 // ... for a B unit (2 / 1)
    constexpr int NST = (PBM * (EPI == 2 ? BN / 2 : BN) * 2 / 16) / 512;    // output stores per thread and tile (16 / 8 / 4)
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // 2 x STEP; reused by the epilogue

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // ---- persistent tile loop: one block per CU walks the tiles of all batch entries; tile ids are dealt so that the blocks
    //      of one XCD (block ids equal mod 8) work on consecutive tiles -- the N-tiles of an M-tile share that XCD's L2
    const long nblk = (long)p.mtiles * p.ntiles;
    const long total = nblk * p.batch;
    const long per = total / 8;
    // The output of a tile leaves LDS for registers at the end of its epilogue and goes to memory only AFTER the next tile's
    // prologue DMA has been issued: the stores then drain under that tile's first K-step instead of in front of it.
    constexpr int CPO = (EPI == 2 ? BN / 2 : BN) * 2 / 16;                  // output chunks per row
    bool st_pending = false;
    uint4 img[NST];
    long pm0 = 0;
    int pn0 = 0, pvr = 0, pz = 0;
    typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
    auto flush_stores = [&]() {
        // branch-free through a buffer descriptor: a chunk outside the tile's valid rows / the matrix's columns gets an offset
        // beyond the descriptor's extent, and the hardware drops the store
        // (the descriptor starts at the tile's first row: offsets stay small whatever the size of the output)
        const int ncol = EPI == 2 ? p.N / 2 : p.N;
        const __amdgpu_buffer_rsrc_t Or = make_rsrc(p.out + pz * p.sO + pm0 * p.ldo, ((p.M - 1 - pm0) * p.ldo + ncol) * 2);
        int stid = tid;
        asm volatile("" : "+v"(stid));
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int idx = q * 512 + stid;
            const int row = idx / CPO, pc = idx % CPO;
            const int c = pc ^ (row & (EPI == 2 ? 7 : 15));
            const int col = (EPI == 2 ? pn0 / 2 : pn0) + c * 8;
            const bool ok = row < pvr && col < ncol;
            const unsigned off = ok ? (unsigned)(((long)row * p.ldo + col) * 2) : 0xC0000000u;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4p{img[q].x, img[q].y, img[q].z, img[q].w}, Or, off, 0, 0);
        }
    };
    for (long t = blockIdx.x; t < total; t += gridDim.x) {
    long tt = t;
    if (tt < per * 8) tt = (tt % 8) * per + tt / 8;
    const int z = (int)(tt / nblk);
    const long bid = tt % nblk;
    const int mt0 = (int)(bid / p.ntiles), nt0 = (int)(bid % p.ntiles);
    const long m0 = (long)mt0 * p.tm;
    const int n0 = nt0 * BN;
#ifdef PH_STAMPS
    const bool stamp_on = (wave == 0 || wave == 4) && lane == 0 && z == 0;
    unsigned long long *st = p.stamps + ((size_t)t * 2 + (wave >> 2)) * 64;
    PH_STAMP(0);
#endif
    const int vr = (int)min((long)p.tm, p.M - m0);       // valid rows of this tile
    const int nrt = (vr + 15) >> 4;                       // ... in 16-row tiles (epilogue) ...
    const bool half_on[2] = {wr * 128 < vr, wr * 128 + 64 < vr};   // ... and which 64-row halves of this wave multiply at all

    // ---- LDS-DMA sources: lane (sub = lane >> 3, pch = lane & 7) fills LDS chunk pch of row `sub` of its 8-row piece with
    //      source chunk pch ^ sub; unit row u = (wave * D + j) * 8 + sub
    const int sub = lane >> 3, pch = lane & 7;
    // CONV: input offset of output position m (its tap (0, 0) pixel); the descriptor starts at this tile's first pixel, so the
    // 32-bit lane offsets stay small however long the recording is
    auto pix_off = [&](long m) -> long {
        const long tf = (long)p.T2 * p.F2;
        const long bb = m / tf, rem = m - bb * tf;
        const long t2 = rem / p.F2, f2 = rem - t2 * p.F2;
        return (((bb * p.T1 + 2 * t2) * p.F1 + 2 * f2) * p.Ci) * 2;
    };
    const long a_base = CONV ? pix_off(m0) : 0;
    const long a_bytes = CONV ? p.in_bytes - a_base : ((p.M - 1) * p.lda + p.K) * 2;
    const __amdgpu_buffer_rsrc_t Ar = make_rsrc(reinterpret_cast<const unsigned char *>(p.A + z * p.sA) + a_base, a_bytes);
    const __amdgpu_buffer_rsrc_t Wr = make_rsrc(p.W + z * p.sW, CONV ? (long)9 * p.N * p.Ci * 2 : ((long)(p.N - 1) * p.ldw + p.K) * 2);
    unsigned a_off[2][DA], b_off[2][DB];          // byte offsets of this lane's 16-byte source chunks (K-step 0)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < DA; ++j) {
            const int u = (wave * DA + j) * 8 + sub;                         // 0..127: wave-row group u >> 6, row u & 63
            const long m = m0 + min((u >> 6) * 128 + h * 64 + (u & 63), vr - 1);   // clamp: loaded, never stored
            a_off[h][j] = CONV ? (unsigned)(pix_off(m) - a_base + 16 * (pch ^ sub)) : (unsigned)(m * p.lda * 2 + 16 * (pch ^ sub));
        }
#pragma unroll
        for (int j = 0; j < DB; ++j) {
            const int u = (wave * DB + j) * 8 + sub;                         // 0..BN/2-1: wave column u / (WN/2)
            const int n = min(n0 + (u / (WN / 2)) * WN + h * (WN / 2) + (u % (WN / 2)), p.N - 1);
            b_off[h][j] = (unsigned)((long)n * (CONV ? p.Ci : p.ldw) * 2 + 16 * (pch ^ sub));
        }
    }
    // wave-uniform byte offset of K-step kt inside a row of A / W
    auto a_soff = [&](int kt) -> unsigned {
        if constexpr (!CONV) return kt * 128;
        const int tap = kt >> p.kshift, kc = kt - (tap << p.kshift);
        const int dt = (tap * 11) >> 5, df = tap - 3 * dt;                  // tap / 3, tap % 3 for tap < 9
        return (unsigned)(((dt * p.F1 + df) * p.Ci) * 2 + kc * 128);
    };
    auto w_soff = [&](int kt) -> unsigned {
        if constexpr (!CONV) return kt * 128;
        const int tap = kt >> p.kshift, kc = kt - (tap << p.kshift);
        return (unsigned)(tap * p.N * p.Ci * 2 + kc * 128);
    };
    auto stage_a = [&](int h, int buf, int kt) {
#ifdef PH_ABL_NODMA
        return;
#endif
#pragma unroll
        for (int j = 0; j < DA; ++j)
            dma16(Ar, a_off[h][j], a_soff(kt), lds + buf * STEP + (h ? OFF_A1 : OFF_A0) + (wave * DA + j) * 1024);
    };
    auto stage_b = [&](int h, int buf, int kt) {
#ifdef PH_ABL_NODMA
        return;
#endif
#pragma unroll
        for (int j = 0; j < DB; ++j)
            dma16(Wr, b_off[h][j], w_soff(kt), lds + buf * STEP + (h ? OFF_B1 : OFF_B0) + (wave * DB + j) * 1024);
    };

    // this lane's bias values, packed as they lie in memory: 4 consecutive columns per (column half, tile); fetched here so
    // that their latency is hidden by the main loop, not paid at the start of the epilogue
    const bf16_t *bz = p.bias ? p.bias + z * p.sB : nullptr;
    uint2 bq[2][BN / 128];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int j = 0; j < BN / 128; ++j) {
            const int n = min(n0 + (wave & 3) * (BN / 4) + nj * (BN / 8) + j * 16 + 4 * (lane >> 4), p.N - 4);
            bq[nj][j] = bz ? *reinterpret_cast<const uint2 *>(bz + n) : uint2{0u, 0u};
        }

    // ---- fragment read addresses: row fr of a 16-row tile, k chunk (ks * 4 + kq) ^ (row & 7); (row & 7) == (fr & 7)
    const int fr = lane & 15, kq = lane >> 4;
    const unsigned frag = fr * 128 + ((kq ^ (fr & 7)) * 16);                 // ks = 1 flips bit 6
    const unsigned la = wr * (64 * 128) + frag;                              // + i * 2048 inside a unit
    const unsigned lb = wc * ((WN / 2) * 128) + frag;                        // + j * 2048

    f32x4p acc[2][2][4][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[a][b][i][j] = f32x4p{0.f, 0.f, 0.f, 0.f};

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
    auto mma = [&](int mi, int nj, const bf16x8p (&bq)[TN][2]) {
        __builtin_amdgcn_s_setprio(1);
#ifdef PH_ABL_NOMFMA
        if (false) {
#else
        if (half_on[mi]) {   // wave-uniform, one branch per phase: a 64-row half beyond the tile's rows costs nothing
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)   // transposed product: D[n][m], a lane owns 4 consecutive n of one m
                        acc[mi][nj][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j][ks], af[i][ks], acc[mi][nj][i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    const int nt = p.K / PBK;                    // >= 2 (checked on the host)

    // EPI 1 at BN = 256: the output image [256][512 B] is exactly the two K-step buffers, and a 16 KiB unit region is 32 image
    // rows.  The staging slots the last two K-steps would leave empty (their units belong to K-steps that do not exist) carry
    // the RESIDUAL rows of the regions they would have filled -- same regions, same phases, so the same hazards are already
    // covered, and every phase keeps issuing exactly one unit (the waits stay at their steady-state counts).  Six of the
    // eight regions arrive this way during the last two K-steps; only the last two are fetched after the loop.
    constexpr bool RESPRE = EPI == 1 && BN == 256;
    const __amdgpu_buffer_rsrc_t Rr = make_rsrc(EPI == 1 ? p.res + z * p.sR : p.A, EPI == 1 ? ((p.M - 1) * p.ldr + p.N) * 2 : 0);
    auto stage_res = [&](int region) {            // region: byte offset of a 16 KiB unit region = 32 rows of the image
        int ln = lane;
        asm volatile("" : "+v"(ln));              // addresses are formed where they are used, not hoisted out of the K loop
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = region + (wave * 2 + j) * 1024 + ln * 16;
            const int row = off >> 9, pc = (off >> 4) & 31;
            const int c = pc ^ (row & 15);
            const long m = m0 + min(row, vr - 1);
            const int n = min(n0 + c * 8, p.N - 8);
            dma16(Rr, (unsigned)((m * p.ldr + n) * 2), 0, lds + region + (wave * 2 + j) * 1024);
        }
    };

    // One K-step = four phases.  `last` = K-steps after this one: the units a phase issues belong to the next K-step (phases
    // 1, 2) or to the one after (phases 3, 4), so the tail issues less and waits for less -- a wave-uniform switch around the
    // same code (no separate tail code: the accumulators keep their registers).  Wait counts = LDS-DMA instructions of the
    // units issued in the last four phases that may still be in flight (A unit: 2 per thread; B unit: 2 at BN = 256, 1 at 128).
    // PH_SYNC(PHASE): the counted wait of a phase, then its barrier.  Base count by (tile shape, K-steps left, phase); the
    // first K-step of a tile that follows another one in this block adds that tile's output stores, which were issued behind
    // the prologue units and ahead of everything the loop issues (`pend`).
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
#define PH_SYNC(PHASE)                                                             \
    do {                                                                           \
        constexpr int W1c = DB == 2 ? (PHASE == 3 ? 6 : PHASE == 4 ? 4 : 8) : (PHASE == 3 ? 4 : PHASE == 4 ? 3 : 6); \
        constexpr int W0c = PHASE == 1 ? 2 : 0;                                    \
        constexpr int WS = DB == 2 ? 8 : 6;                                        \
        if (RESPRE || last >= 2) { if (pend) wait_vm<WS + NST>(); else wait_vm<WS>(); }        \
        else if (last == 1) { if (pend) wait_vm<W1c + NST>(); else wait_vm<W1c>(); }           \
        else { if (pend) wait_vm<W0c + NST>(); else wait_vm<W0c>(); }              \
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
        const bool pend = st_pending && kt == 0;
#if defined(PH_STAMPS) && defined(PH_FINE_STAMPS)   // (needs K / 64 <= 18: the indices sit above the per-K-step stamps)
#define PH_FINE(i) do { if (stamp_on && kt == 4) st[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_FINE(i) do { } while (0)
#endif
        // phase 1: quadrant (rows 0-63, column half 0)
        PH_FINE(20);
        read_b(0, PAR, bf0);
        read_a(0, PAR);
        if (last >= 1) stage_b(1, PAR ^ 1, kt + 1);
        else if constexpr (RESPRE) stage_res((PAR ^ 1) * STEP + OFF_B1);
        PH_FINE(21);
        PH_SYNC(1);
        PH_FINE(22);
        mma(0, 0, bf0);
        PH_FINE(23);
        PH_END();
        PH_FINE(24);
        // phase 2: (rows 0-63, column half 1)
        read_b(1, PAR, bf1);
        if (last >= 1) stage_a(1, PAR ^ 1, kt + 1);
        else if constexpr (RESPRE) stage_res((PAR ^ 1) * STEP + OFF_A1);
        PH_FINE(25);
        PH_SYNC(2);
        PH_FINE(26);
        mma(0, 1, bf1);
        PH_FINE(27);
        PH_END();
        PH_FINE(28);
        // phase 3: (rows 64-127, column half 1)
        read_a(1, PAR);
        if (last >= 2) stage_a(0, PAR, kt + 2);
        else if constexpr (RESPRE) stage_res(PAR * STEP + OFF_A0);
        PH_FINE(29);
        PH_SYNC(3);
        PH_FINE(30);
        mma(1, 1, bf1);
        PH_FINE(31);
        PH_END();
        PH_FINE(32);
        // phase 4: (rows 64-127, column half 0)
        if (last >= 2) stage_b(0, PAR, kt + 2);
        else if constexpr (RESPRE) stage_res(PAR * STEP + OFF_B0);
        PH_FINE(33);
        PH_SYNC(4);
        PH_FINE(34);
        mma(1, 0, bf0);
        PH_FINE(35);
        PH_END();
        PH_FINE(36);
#undef PH_FINE
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: the six units the steady state would have in flight or landed when K-step 0 starts
    stage_a(0, 0, 0); stage_b(0, 0, 0); stage_b(1, 0, 0); stage_a(1, 0, 0); stage_a(0, 1, 1); stage_b(0, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (st_pending) {
        flush_stores();
        wait_vm<(DB == 2 ? 8 : 6) + NST>();
    } else {
        wait_vm<(DB == 2 ? 8 : 6)>();
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind waves 0-3
    PH_STAMP(1);

    for (int kt = 0; kt < nt; kt += 2) {
        kstep(I0{}, kt);
        PH_STAMP(2 + (kt < 40 ? kt : 40));
        if (kt + 1 < nt) { kstep(I1{}, kt + 1); PH_STAMP(3 + (kt < 40 ? kt : 40)); }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // re-join: every wave has finished reading its operands
    PH_STAMP(50);
#undef PH_SYNC
#undef PH_END

    // ---- epilogue ----------------------------------------------------------------------------------------------
    // LDS image of the output tile: [256 rows][BN bf16], 16-byte chunk c of row r stored at chunk c ^ (r & 15)
    constexpr int ROWB = BN * 2;                  // bytes per row (512 / 256)
    constexpr int CPR = ROWB / 16;                // chunks per row (32 / 16)
    constexpr int ON = EPI == 2 ? BN / 2 : BN;    // output columns of the tile
    if constexpr (RESPRE) {
        const int b = (nt - 1) & 1;               // the last K-step's buffer: its A_m1 and B_n1 regions are still to come
        stage_res(b * STEP + OFF_A1);
        stage_res(b * STEP + OFF_B1);
        PH_WAIT(0);
        __builtin_amdgcn_s_barrier();
    } else if constexpr (EPI == 1) {
#pragma unroll
        for (int q = 0; q < (PBM * CPR) / 512; ++q) {
            const int idx = q * 512 + tid;
            const int row = idx / CPR, pc = idx % CPR;
            const int c = pc ^ (row & 15);
            const long m = m0 + min(row, vr - 1);
            const int n = min(n0 + c * 8, p.N - 8);
            dma16(Rr, (unsigned)((m * p.ldr + n) * 2), 0, lds + (q * 512 + wave * 64) * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
        PH_WAIT(0);
        __builtin_amdgcn_s_barrier();
    }
    PH_STAMP(51);
    int etid = tid;
    asm volatile("" : "+v"(etid));               // epilogue addresses are formed here, per tile, not hoisted out of the tile loop
    const int efr = etid & 15, ekq = (etid >> 4) & 3;
    auto bias4 = [&](const uint2 &q, float (&b)[4]) {
        b[0] = bf16_bits_to_f32(q.x & 0xffffu); b[1] = __uint_as_float(q.x & 0xffff0000u);
        b[2] = bf16_bits_to_f32(q.y & 0xffffu); b[3] = __uint_as_float(q.y & 0xffff0000u);
    };
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_barrier(0);    // one row group at a time: keeps the epilogue's registers few
            if (wr * 8 + mi * 4 + i >= nrt) continue;
            const int row = wr * 128 + mi * 64 + i * 16 + efr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (EPI == 2) {
                    const int oc = wc * (WN / 2) + j * 16 + 4 * ekq;          // output column inside the tile (BN / 2 wide)
                    float bv[4], bg[4], o[4];
                    bias4(bq[0][j], bv);
                    bias4(bq[1][j], bg);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float a = fmaf(acc[mi][0][i][j][g], p.alpha, bv[g]);
                        const float b = fmaf(acc[mi][1][i][j][g], p.alpha, bg[g]);
                        o[g] = a * __builtin_amdgcn_rcpf(1.f + __expf(-b));
                    }
                    uint2 w;
                    w.x = pack_bf16(o[0], o[1]);
                    w.y = pack_bf16(o[2], o[3]);
                    const int c = oc >> 3;
                    *reinterpret_cast<uint2 *>(lds + row * (ROWB / 2) + ((c ^ (row & 7)) * 16) + ((oc >> 2) & 1) * 8) = w;
                } else {
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) {
                        const int col = wc * WN + nj * (WN / 2) + j * 16 + 4 * ekq;
                        unsigned char *dst = lds + row * ROWB + (((col >> 3) ^ (row & 15)) * 16) + ((col >> 2) & 1) * 8;
                        float bv[4], o[4];
                        bias4(bq[nj][j], bv);
#pragma unroll
                        for (int g = 0; g < 4; ++g) o[g] = act_apply<ACT>(fmaf(acc[mi][nj][i][j][g], p.alpha, bv[g]));
                        if constexpr (EPI == 1) {
                            const uint2 rq = *reinterpret_cast<const uint2 *>(dst);
                            o[0] += bf16_bits_to_f32(rq.x & 0xffffu); o[1] += __uint_as_float(rq.x & 0xffff0000u);
                            o[2] += bf16_bits_to_f32(rq.y & 0xffffu); o[3] += __uint_as_float(rq.y & 0xffff0000u);
                        }
                        uint2 w;
                        w.x = pack_bf16(o[0], o[1]);
                        w.y = pack_bf16(o[2], o[3]);
                        *reinterpret_cast<uint2 *>(dst) = w;
                    }
                }
            }
        }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    PH_STAMP(52);
    // the image leaves LDS for registers (the accumulators' registers are free now); after the barrier the LDS belongs to the
    // next tile
#pragma unroll
    for (int q = 0; q < NST; ++q) {
        const int idx = q * 512 + etid;
        const int row = idx / CPO, pc = idx % CPO;
        img[q] = *reinterpret_cast<const uint4 *>(lds + row * (CPO * 16) + pc * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    pm0 = m0; pn0 = n0; pvr = vr; pz = z;
    st_pending = true;
    PH_STAMP(53);
    }   // tile loop
    if (st_pending) flush_stores();
}

template <int BN, int EPI, int ACT, bool CONV = false>
int launch_ph(const PhParams &p, int batch, hipStream_t s) {
    constexpr size_t step = 2 * 128 * 128 + 2 * (BN / 2) * 128;
    constexpr size_t lds = 2 * step > (size_t)PBM * BN * 2 ? 2 * step : (size_t)PBM * BN * 2;
    auto kern = gemm_ph_kernel<BN, EPI, ACT, CONV>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PAFC_ERR_LAUNCH;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return PAFC_ERR_LAUNCH;
    PhParams q = p;
    q.batch = batch;
    const long total = (long)p.mtiles * p.ntiles * batch;
    const long grid = total < cus ? total : cus;              // one 512-thread block per CU (128 KiB of LDS each)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, q);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // namespace
}  // namespace pafc

// Same contract as pafc_gemm_bf16 (which calls this for the shapes it suits); exported for A/B measurements.
extern "C" int pafc_gemm_bf16_ph(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W,
                                 long ldw, long strideW, const void *bias, long strideBias, const void *residual, long ldr,
                                 long strideR, void *out, long ldo, long strideO, float alpha, int act, int tile_n,
                                 int tile_m, pafc_stream_t stream) {
    if (!A || !W || !out) return PAFC_ERR_NULL_POINTER;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535) return PAFC_ERR_BAD_DIMS;
    if (act < 0 || act > 4) return PAFC_ERR_UNSUPPORTED;
    const bool glu = act == 4;
    if (tile_n != 256 && tile_n != 128) return PAFC_ERR_UNSUPPORTED;
    if (tile_m < 64 || tile_m > 256 || tile_m % 64) return PAFC_ERR_UNSUPPORTED;
    if (N % 8 || K % 64 || K < 128) return PAFC_ERR_UNSUPPORTED;
    if (glu && (N % tile_n || residual)) return PAFC_ERR_UNSUPPORTED;
    if (lda < K || ldw < K || ldo < (glu ? N / 2 : N) || (residual && ldr < N)) return PAFC_ERR_BAD_DIMS;
    if ((lda | ldw | ldo | strideA | strideW | strideO) % 8 || (residual && ((ldr | strideR) % 8))) return PAFC_ERR_ALIGNMENT;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)residual) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    // 31-bit byte extents inside one batch entry of A / W / residual (buffer descriptors)
    if ((double)M * lda * 2 >= 2.0e9 || (double)N * ldw * 2 >= 2.0e9 || (residual && (double)M * ldr * 2 >= 2.0e9))
        return PAFC_ERR_UNSUPPORTED;
    pafc::PhParams p{};
    p.A = (const pafc::bf16_t *)A; p.W = (const pafc::bf16_t *)W; p.bias = (const pafc::bf16_t *)bias;
    p.res = (const pafc::bf16_t *)residual; p.out = (pafc::bf16_t *)out;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr;
    p.sA = strideA; p.sW = strideW; p.sO = strideO; p.sB = strideBias; p.sR = strideR;
    p.alpha = alpha;
    p.tm = tile_m;
    p.mtiles = (int)((M + tile_m - 1) / tile_m);
    p.ntiles = (N + tile_n - 1) / tile_n;
    if ((long)p.mtiles * p.ntiles > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    if (residual && act != 0) return PAFC_ERR_UNSUPPORTED;      // the layer never pairs a residual with an activation
#define PH_DISPATCH(BN)                                                        \
    do {                                                                       \
        if (glu) return pafc::launch_ph<BN, 2, 0>(p, batch, s);                \
        if (residual) return pafc::launch_ph<BN, 1, 0>(p, batch, s);           \
        switch (act) {                                                         \
            case 1: return pafc::launch_ph<BN, 0, 1>(p, batch, s);             \
            case 2: return pafc::launch_ph<BN, 0, 2>(p, batch, s);             \
            case 3: return pafc::launch_ph<BN, 0, 3>(p, batch, s);             \
            default: return pafc::launch_ph<BN, 0, 0>(p, batch, s);            \
        }                                                                      \
    } while (0)
    if (tile_n == 256) PH_DISPATCH(256);
    PH_DISPATCH(128);
#undef PH_DISPATCH
}

// The second subsampling convolution, Conv2d(Ci, Co, 3, stride 2) + bias (+ ReLU) on NHWC bf16 (wenet/transformer/
// subsampling.py:187-192), as an implicit GEMM on the phase-pipelined kernel: same contract as pafc_conv3x3s2_nhwc_bf16,
// which dispatches here when the problem fills the chip with 256-wide tiles.  PAFC_ERR_UNSUPPORTED = take the other kernel.
extern "C" int pafc_conv3x3s2_nhwc_bf16_ph(int B, int T1, int F1, int Ci, int Co, const void *in, const void *w_tap_co_ci,
                                           const void *bias, void *out, int relu, int tile_m, pafc_stream_t stream) {
    if (!in || !w_tap_co_ci || !out) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || T1 < 3 || F1 < 3 || Ci <= 0 || Co <= 0) return PAFC_ERR_BAD_DIMS;
    if (Ci % 64 || (Ci / 64) & (Ci / 64 - 1) || Co % 8) return PAFC_ERR_UNSUPPORTED;       // 64-channel K-steps, a power of two per tap
    if (tile_m != 256 && tile_m != 192 && tile_m != 128) return PAFC_ERR_UNSUPPORTED;
    if ((((uintptr_t)in | (uintptr_t)w_tap_co_ci | (uintptr_t)out) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
    const long M = (long)B * T2 * F2;
    if ((double)9 * Co * Ci * 2 >= 2.0e9) return PAFC_ERR_UNSUPPORTED;
    pafc::PhParams p{};
    p.A = (const pafc::bf16_t *)in; p.W = (const pafc::bf16_t *)w_tap_co_ci; p.bias = (const pafc::bf16_t *)bias; p.out = (pafc::bf16_t *)out;
    p.M = M; p.N = Co; p.K = 9 * Ci; p.lda = Ci; p.ldw = Ci; p.ldo = Co; p.alpha = 1.f;
    p.T1 = T1; p.F1 = F1; p.T2 = T2; p.F2 = F2; p.Ci = Ci;
    p.in_bytes = (long)B * T1 * F1 * Ci * 2;
    int ks = 0;
    while ((64 << ks) < Ci) ++ks;
    p.kshift = ks;
    p.tm = tile_m;
    p.mtiles = (int)((M + tile_m - 1) / tile_m);
    p.ntiles = (Co + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    return relu ? pafc::launch_ph<256, 0, 3, true>(p, 1, s) : pafc::launch_ph<256, 0, 0, true>(p, 1, s);
}
