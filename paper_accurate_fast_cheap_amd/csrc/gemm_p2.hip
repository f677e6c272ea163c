// The short-K projections as a bf16 GEMM whose tiles overlap each other's epilogues (gfx950): 128 x 256 tiles, 4 waves,
// TWO blocks per CU, a three-slot LDS ring of 32-deep K-steps.  C ABI: include/pafc_encoder_ops.h (pafc_gemm_bf16 picks).
//
// Why.  gemm_ph.hip (one 512-thread block per CU, 256 x 256 tiles) keeps the matrix pipe at ~70 % inside its K loop, but at
// K = 512 -- most of the encoder layer's projections -- a tile spends 40 % of its time outside the loop (prologue latency,
// 128 SiLU / GLU evaluations per lane, the output's trip through LDS): its MfmaUtil is 30-36 %.  All eight waves of the block
// are in the epilogue at once, so nothing multiplies meanwhile.  Here a CU hosts two INDEPENDENT blocks of four waves (one
// wave of each per SIMD): while one block is between loops the other one is inside its loop, and the two loops fill each
// other's load segments.  Two blocks need <= 80 KiB of LDS each: a 128 x 256 tile with K-step 64 would take 96 KiB for two
// steps, so the K-step is 32 and the ring holds three of them (72 KiB), which also keeps three steps of prefetch in flight.
//
// One K-step = two phases of 16 MFMAs (all 128 rows x one half of the wave's 64 columns).  Units by read phase: A (128 rows
// x 32 k, 8 KiB) and B_n0 (the first column halves, 8 KiB) are read in phase 0, B_n1 in phase 1; a unit is re-filled two
// phases after its read (one barrier per phase: a wave passes the next phase's barrier only after the MFMAs that needed the
// reads), i.e. for the K-step three ahead -- phase 0 issues A and B_n0 of step t + 2, phase 1 issues B_n1 of step t + 2 --
// and is waited for one phase before its read with counted vmcnt (10 / 8 outstanding, never 0).  64-byte LDS rows: chunk c
// of unit row u sits at c ^ ((0 - (u >> 2)) & 3), applied on the DMA source side (conflict-free ds_read_b128, worked out in
// DESIGN.md section 4).  Everything else -- transposed product so that a lane owns four consecutive output columns, bias /
// activation / GLU / residual epilogue on the fp32 accumulator, output image in LDS, stores through a buffer descriptor
// issued behind the next tile's prologue DMA -- is gemm_ph.hip's.
#include <type_traits>

#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8q __attribute__((ext_vector_type(8)));
typedef float f32x2q __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2q __attribute__((ext_vector_type(2)));
typedef unsigned u32x4q __attribute__((ext_vector_type(4)));

struct P2Params {
    const bf16_t *A, *W, *bias, *res;
    bf16_t *out;
    long M;
    int N, K;
    long lda, ldw, ldo, ldr;          // row strides (elements)
    long sA, sW, sO, sB, sR;          // batch strides (elements); sB = 0 shares the bias
    float alpha;
    int mtiles, ntiles, batch;
};

__device__ __forceinline__ void dma16q(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uniform_off, void *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, lane_off,
                                             uniform_off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {       // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2q{lo, hi}, bf16x2q));
}
template <int ACT>
__device__ __forceinline__ float act2(float v) {
    if constexpr (ACT == 1) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    if constexpr (ACT == 2) return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
    if constexpr (ACT == 3) return fmaxf(v, 0.f);
    return v;
}
template <int N>
__device__ __forceinline__ void wait_vmq() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int QBM = 128, QBN = 256, QBK = 32;
constexpr int QUNIT = 128 * 64;                   // bytes of a unit: 128 rows x 32 k
constexpr int QSLOT = 3 * QUNIT;                  // A | B_n0 | B_n1
constexpr int QLDS = 3 * QSLOT;                   // 72 KiB

// EPI: 0 plain, 1 residual, 2 GLU (a wave's 64 columns = 32 values + the 32 gates of the same channels); ACT as gemm_ph.hip
template <int EPI, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_p2_kernel(const P2Params p) {
    constexpr int ON = EPI == 2 ? QBN / 2 : QBN;                 // output columns of a tile
    constexpr int CPO = ON * 2 / 16;                             // 16-byte chunks per output row (32 / 16)
    constexpr int NST = (QBM * CPO) / 256;                       // output stores per thread and tile (16 / 8)
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = its 64 output columns
    const long nblk = (long)p.mtiles * p.ntiles;
    const long total = nblk * p.batch;
    const long per = total / 8;
    const int nt = p.K / QBK;                                     // >= 4 (checked on the host)

    bool st_pending = false;
    uint4 img[NST];
    long pm0 = 0;
    int pn0 = 0, pvr = 0, pz = 0;
    auto flush_stores = [&]() {
        const int ncol = EPI == 2 ? p.N / 2 : p.N;
        const __amdgpu_buffer_rsrc_t Or = rsrc_of(p.out + pz * p.sO + pm0 * p.ldo, ((p.M - 1 - pm0) * p.ldo + ncol) * 2);
        int stid = tid;
        asm volatile("" : "+v"(stid));
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int idx = q * 256 + stid;
            const int row = idx / CPO, pc = idx % CPO;
            const int c = pc ^ (row & (EPI == 2 ? 7 : 15));
            const int col = (EPI == 2 ? pn0 / 2 : pn0) + c * 8;
            const bool ok = row < pvr && col < ncol;
            const unsigned off = ok ? (unsigned)(((long)row * p.ldo + col) * 2) : 0xC0000000u;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4q{img[q].x, img[q].y, img[q].z, img[q].w}, Or, off, 0, 0);
        }
    };

    for (long t = blockIdx.x; t < total; t += gridDim.x) {
        long tt = t;
        if (tt < per * 8) tt = (tt % 8) * per + tt / 8;          // the N-tiles of an M-tile on one XCD (blocks b, b + 8, ...)
        const int z = (int)(tt / nblk);
        const long bid = tt % nblk;
        const int mt0 = (int)(bid / p.ntiles), nt0 = (int)(bid % p.ntiles);
        const long m0 = (long)mt0 * QBM;
        const int n0 = nt0 * QBN;
        const int vr = (int)min((long)QBM, p.M - m0);
        const int nrt = (vr + 15) >> 4;

        // ---- LDS-DMA sources: a wave instruction fills 16 rows x 64 B; lane l -> row l >> 2, chunk position l & 3, which
        //      holds source chunk (l & 3) ^ f(row), f = (0 - (row >> 2)) & 3
        const int prow = lane >> 2, ppos = lane & 3;
        const int pchunk = ppos ^ ((0 - (prow >> 2)) & 3);
        const __amdgpu_buffer_rsrc_t Ar = rsrc_of(p.A + z * p.sA, ((p.M - 1) * p.lda + p.K) * 2);
        const __amdgpu_buffer_rsrc_t Wr = rsrc_of(p.W + z * p.sW, ((long)(p.N - 1) * p.ldw + p.K) * 2);
        unsigned a_off[2], b_off[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = (wc * 2 + j) * 16 + prow;                              // unit row 0..127
            a_off[j] = (unsigned)((m0 + min(u, vr - 1)) * p.lda * 2 + 16 * pchunk);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = min(n0 + (u >> 5) * 64 + h * 32 + (u & 31), p.N - 1);  // wave column u / 32, column half h
                b_off[h][j] = (unsigned)((long)n * p.ldw * 2 + 16 * pchunk);
            }
        }
        auto stage_a = [&](int slot, int kt) {
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16q(Ar, a_off[j], kt * 64, lds + slot * QSLOT + (wc * 2 + j) * 1024);
        };
        auto stage_b = [&](int h, int slot, int kt) {
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16q(Wr, b_off[h][j], kt * 64, lds + slot * QSLOT + (1 + h) * QUNIT + (wc * 2 + j) * 1024);
        };

        const bf16_t *bz = p.bias ? p.bias + z * p.sB : nullptr;
        uint2 bq[2][2];
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = min(n0 + wc * 64 + nj * 32 + j * 16 + 4 * (lane >> 4), p.N - 4);
                bq[nj][j] = bz ? *reinterpret_cast<const uint2 *>(bz + n) : uint2{0u, 0u};
            }

        // fragment reads: row fr of a 16-row tile, k chunk kq of 4
        const int fr = lane & 15, kq = lane >> 4;
        const unsigned frag = fr * 64 + ((kq ^ ((0 - (fr >> 2)) & 3)) * 16);
        const unsigned lb = wc * (32 * 64) + frag;

        f32x4q acc[2][8][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][j] = f32x4q{0.f, 0.f, 0.f, 0.f};
        bf16x8q af[8], bf0[2], bf1[2];

        auto read_a = [&](int slot) {
            const unsigned char *base = lds + slot * QSLOT;
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8q *>(base + frag + i * 1024);
        };
        auto read_b = [&](int h, int slot, bf16x8q (&dst)[2]) {
            const unsigned char *base = lds + slot * QSLOT + (1 + h) * QUNIT;
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[j] = *reinterpret_cast<const bf16x8q *>(base + lb + j * 1024);
        };
        auto mma = [&](int nj, const bf16x8q (&bw)[2]) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)        // transposed product: D[n][m], a lane owns 4 consecutive n of one m
                    acc[nj][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], af[i], acc[nj][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };

        // ---- prologue: K-steps 0 and 1 complete; the previous tile's stores go out behind them
        stage_a(0, 0); stage_b(0, 0, 0); stage_b(1, 0, 0); stage_a(1, 1); stage_b(0, 1, 1); stage_b(1, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (st_pending) {
            flush_stores();
            wait_vmq<8 + NST>();
        } else {
            wait_vmq<8>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();

        // counted wait (DMA instructions issued after the unit the NEXT phase reads) + the phase's barrier
#define P2_SYNC(W2, W1, W0)                                                               \
    do {                                                                                  \
        if (last >= 2) { if (pend) wait_vmq<W2 + NST>(); else wait_vmq<W2>(); }           \
        else if (last == 1) { if (pend) wait_vmq<W1 + NST>(); else wait_vmq<W1>(); }      \
        else { if (pend) wait_vmq<W0 + NST>(); else wait_vmq<W0>(); }                     \
        __builtin_amdgcn_s_barrier();                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)

        auto kstep = [&](auto slotc, int kt) {
            constexpr int S = decltype(slotc)::value;             // this K-step's slot; K-step kt + 2 goes to slot (S + 2) % 3
            constexpr int S2 = (S + 2) % 3;
            const int last = nt - 1 - kt;
            // phase 0: all rows x column half 0
            __builtin_amdgcn_sched_barrier(0);
            read_b(0, S, bf0);
            read_a(S);
            if (last >= 2) { stage_a(S2, kt + 2); stage_b(0, S2, kt + 2); }
            {
                const bool pend = st_pending && kt <= 1;          // next phase reads B_n1(kt): a prologue unit for kt <= 1
                P2_SYNC(10, 6, 0);
            }
            mma(0, bf0);
            // phase 1: all rows x column half 1
            __builtin_amdgcn_sched_barrier(0);
            read_b(1, S, bf1);
            if (last >= 2) stage_b(1, S2, kt + 2);
            {
                const bool pend = st_pending && kt == 0;          // next phase reads A, B_n0 of K-step 1: prologue units
                P2_SYNC(8, 2, 0);
            }
            mma(1, bf1);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        for (int kt = 0; kt < nt; kt += 3) {
            kstep(I0{}, kt);
            if (kt + 1 < nt) kstep(I1{}, kt + 1);
            if (kt + 2 < nt) kstep(I2{}, kt + 2);
        }
#undef P2_SYNC
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();              // every wave has finished reading its operands: the LDS is the epilogue's

        // ---- epilogue: output image [128 rows][ON bf16] in LDS, chunk c of row r at c ^ (r & 15) (GLU: & 7)
        constexpr int ROWB = ON * 2;
        if constexpr (EPI == 1) {
            const __amdgpu_buffer_rsrc_t Rr = rsrc_of(p.res + z * p.sR, ((p.M - 1) * p.ldr + p.N) * 2);
#pragma unroll
            for (int q = 0; q < NST; ++q) {
                const int idx = q * 256 + tid;
                const int row = idx / CPO, pc = idx % CPO;
                const int c = pc ^ (row & 15);
                const long m = m0 + min(row, vr - 1);
                const int n = min(n0 + c * 8, p.N - 8);
                dma16q(Rr, (unsigned)((m * p.ldr + n) * 2), 0, lds + (q * 256 + wc * 64) * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
            wait_vmq<0>();
            __builtin_amdgcn_s_barrier();
        }
        int etid = tid;
        asm volatile("" : "+v"(etid));             // epilogue addresses are formed here, per tile
        const int efr = etid & 15, ekq = (etid >> 4) & 3;
        auto bias4 = [&](const uint2 &q, float (&b)[4]) {
            b[0] = bf16_bits_to_f32(q.x & 0xffffu); b[1] = __uint_as_float(q.x & 0xffff0000u);
            b[2] = bf16_bits_to_f32(q.y & 0xffffu); b[3] = __uint_as_float(q.y & 0xffff0000u);
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if (i >= nrt) continue;
            const int row = i * 16 + efr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (EPI == 2) {
                    const int oc = wc * 32 + j * 16 + 4 * ekq;
                    float bv[4], bg[4], o[4];
                    bias4(bq[0][j], bv);
                    bias4(bq[1][j], bg);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float a = fmaf(acc[0][i][j][g], p.alpha, bv[g]);
                        const float b = fmaf(acc[1][i][j][g], p.alpha, bg[g]);
                        o[g] = a * __builtin_amdgcn_rcpf(1.f + __expf(-b));
                    }
                    uint2 w;
                    w.x = pack2(o[0], o[1]);
                    w.y = pack2(o[2], o[3]);
                    *reinterpret_cast<uint2 *>(lds + row * ROWB + (((oc >> 3) ^ (row & 7)) * 16) + ((oc >> 2) & 1) * 8) = w;
                } else {
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) {
                        const int col = wc * 64 + nj * 32 + j * 16 + 4 * ekq;
                        unsigned char *dst = lds + row * ROWB + (((col >> 3) ^ (row & 15)) * 16) + ((col >> 2) & 1) * 8;
                        float bv[4], o[4];
                        bias4(bq[nj][j], bv);
#pragma unroll
                        for (int g = 0; g < 4; ++g) o[g] = act2<ACT>(fmaf(acc[nj][i][j][g], p.alpha, bv[g]));
                        if constexpr (EPI == 1) {
                            const uint2 rq = *reinterpret_cast<const uint2 *>(dst);
                            o[0] += bf16_bits_to_f32(rq.x & 0xffffu); o[1] += __uint_as_float(rq.x & 0xffff0000u);
                            o[2] += bf16_bits_to_f32(rq.y & 0xffffu); o[3] += __uint_as_float(rq.y & 0xffff0000u);
                        }
                        uint2 w;
                        w.x = pack2(o[0], o[1]);
                        w.y = pack2(o[2], o[3]);
                        *reinterpret_cast<uint2 *>(dst) = w;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // the image leaves LDS for registers; after the barrier the LDS belongs to the next tile
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int idx = q * 256 + etid;
            const int row = idx / CPO, pc = idx % CPO;
            img[q] = *reinterpret_cast<const uint4 *>(lds + row * ROWB + pc * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        pm0 = m0; pn0 = n0; pvr = vr; pz = z;
        st_pending = true;
    }
    if (st_pending) flush_stores();
}

template <int EPI, int ACT>
int launch_p2(const P2Params &p, hipStream_t s) {
    auto kern = gemm_p2_kernel<EPI, ACT>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, QLDS) != hipSuccess) return PAFC_ERR_LAUNCH;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return PAFC_ERR_LAUNCH;
    const long total = (long)p.mtiles * p.ntiles * p.batch;
    const long grid = total < 2L * cus ? total : 2L * cus;       // two 256-thread blocks per CU (72 KiB of LDS each)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), QLDS, s, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // namespace
}  // namespace pafc

// Same contract as pafc_gemm_bf16 (which calls this for the shapes it suits); exported for A/B measurements.
extern "C" int pafc_gemm_bf16_p2(long M, int N, int K, int batch, const void *A, long lda, long strideA, const void *W, long ldw,
                                 long strideW, const void *bias, long strideBias, const void *residual, long ldr, long strideR,
                                 void *out, long ldo, long strideO, float alpha, int act, pafc_stream_t stream) {
    if (!A || !W || !out) return PAFC_ERR_NULL_POINTER;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535) return PAFC_ERR_BAD_DIMS;
    if (act < 0 || act > 4) return PAFC_ERR_UNSUPPORTED;
    const bool glu = act == 4;
    if (N % 8 || K % 32 || K < 128) return PAFC_ERR_UNSUPPORTED;
    if (glu && (N % 256 || residual)) return PAFC_ERR_UNSUPPORTED;
    if (residual && act != 0) return PAFC_ERR_UNSUPPORTED;
    if (lda < K || ldw < K || ldo < (glu ? N / 2 : N) || (residual && ldr < N)) return PAFC_ERR_BAD_DIMS;
    if ((lda | ldw | ldo | strideA | strideW | strideO) % 8 || (residual && ((ldr | strideR) % 8))) return PAFC_ERR_ALIGNMENT;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)residual) & 15) != 0) return PAFC_ERR_ALIGNMENT;
    if ((double)M * lda * 2 >= 2.0e9 || (double)N * ldw * 2 >= 2.0e9 || (residual && (double)M * ldr * 2 >= 2.0e9))
        return PAFC_ERR_UNSUPPORTED;
    pafc::P2Params p{};
    p.A = (const pafc::bf16_t *)A; p.W = (const pafc::bf16_t *)W; p.bias = (const pafc::bf16_t *)bias;
    p.res = (const pafc::bf16_t *)residual; p.out = (pafc::bf16_t *)out;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr;
    p.sA = strideA; p.sW = strideW; p.sO = strideO; p.sB = strideBias; p.sR = strideR;
    p.alpha = alpha;
    p.mtiles = (int)((M + 127) / 128);
    p.ntiles = (N + 255) / 256;
    p.batch = batch;
    if ((long)p.mtiles * p.ntiles * batch > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    hipStream_t s = (hipStream_t)stream;
    if (glu) return pafc::launch_p2<2, 0>(p, s);
    if (residual) return pafc::launch_p2<1, 0>(p, s);
    switch (act) {
        case 1: return pafc::launch_p2<0, 1>(p, s);
        case 2: return pafc::launch_p2<0, 2>(p, s);
        case 3: return pafc::launch_p2<0, 3>(p, s);
        default: return pafc::launch_p2<0, 0>(p, s);
    }
}
