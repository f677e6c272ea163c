// WKV-6 forward and backward for gfx950: T-parallel three-pass chunked scan, both directions in one grid.
//
// What it computes (per batch b, head h; j = key index, i = value index, d = exp(-exp(w))):
//     y_t[i]  = sum_j r_t[j] * (u[j] k_t[j] v_t[i] + S[j][i])
//     S[j][i] <- S[j][i] d_t[j] + k_t[j] v_t[i]
// i.e. kernel_forward of the reference (wenet/rwkv_v6/cuda/wkv6_cuda.cu:8-63), with the optional
// initial state of wkv6state_cuda.cu:6-65 and a final-state output the reference lacks.
//
// Schedule (see include/pafc_wkv6.h): one wave64 per (chunk, b*h, direction).  Lane i owns column i of
// the 64x64 state in 64 VGPRs (same ownership as the reference thread).  A tile of TT time steps is
// fetched with 16-byte lane accesses (a 64-channel row is 128 B in bf16, so one wave instruction brings
// 8 rows), converted once, staged in LDS as f32 and then broadcast-read as float4.  The next tile's
// global loads are issued before the current tile is consumed.  y goes back through LDS so that it is
// also stored 16 B per lane.  The bonus term sum_j r u k is a per-step wave reduction, which takes u out
// of the inner loop: 3 VALU ops per (j, i, t).
#include <stdlib.h>
#include "pafc_common.h"
#include "../../include/pafc_wkv6.h"

namespace pafc {
namespace {

constexpr int N = 64;   // head size
constexpr int TT = 8;   // time steps per LDS tile
constexpr int CHUNK_ALIGN = 16;  // chunk lengths are multiples of the MFMA kernel's block (and of TT)

struct DirArgs {
    const void *r, *k, *v, *w, *u;
    void *y;
    const float *s_in;
    float *s_out;
    int reverse;
    const void *wb;   // optional per-channel bias added to w (time_decay), (H * N) in the element type, or null
};

struct FwdParams {
    DirArgs d[2];
    int B, T, C, H;
    int L, NC;           // chunk length (multiple of TT unless == T) and number of chunks
    int nc_local;        // chunks whose local state pass A must produce (NC-1, or NC when a final state is wanted)
    int prio;            // wave priority scheme of pass C (PAFC_WKV6_PRIO): 3 = raised from the level operands on (default), 2 = tail only, 0 = none
    int order;           // grid order of the channel-lane passes A and C (PAFC_WKV6_ORDER): 0 = chunks of a head back to back, 1 = heads of a chunk
    int rev_c;           // pass C walks the grid in the REVERSE order of pass A (PAFC_WKV6_REVC): what pass A read last is read first
    float *ws_state;     // [ndir][B][H][NC][N(j)][N(i)]
    float *ws_decay;     // [ndir][B][H][NC][N(j)]
};

template <typename ET> struct TileGeom {
    static constexpr int EPL = Elem<ET>::kPerLane;   // elements per lane per 16-B access
    static constexpr int LPR = N / EPL;              // lanes per 64-channel row
    static constexpr int RPL = kWave / LPR;          // rows per wave-wide access
    static constexpr int NLD = TT / RPL;             // accesses per TT-row tile
    static_assert(TT % RPL == 0, "tile must be a whole number of wave accesses");
};

// One wave-wide 16-B/lane read of rows [q*RPL, (q+1)*RPL) of a TT x 64 tile.
template <typename ET>
__device__ __forceinline__ uint4 tile_load(const ET *base, int q, int lane, int s0, int s_end, int T, int C,
                                           int reverse) {
    using G = TileGeom<ET>;
    const int tt = q * G::RPL + lane / G::LPR;
    const int col = (lane % G::LPR) * G::EPL;
    const int s = s0 + tt;
    uint4 out = make_uint4(0, 0, 0, 0);
    if (s < s_end) {
        const int t = reverse ? (T - 1 - s) : s;
        out = *reinterpret_cast<const uint4 *>(base + (size_t)t * C + col);
    }
    return out;
}

template <typename ET, bool WRITE_Y>
__global__ __launch_bounds__(64, WRITE_Y ? 3 : 4) void wkv6_chunk_kernel(const FwdParams p) {
    using G = TileGeom<ET>;
    using E = Elem<ET>;
    const int c = blockIdx.x;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int dir = blockIdx.z;
    const int lane = threadIdx.x;
    const DirArgs &D = p.d[dir];
    const int T = p.T, C = p.C;

    __shared__ __attribute__((aligned(16))) float s_r[WRITE_Y ? TT : 1][N];
    __shared__ __attribute__((aligned(16))) float s_k[TT][N];
    __shared__ __attribute__((aligned(16))) float s_d[TT][N];
    __shared__ __attribute__((aligned(16))) float s_v[TT][N];
    __shared__ __attribute__((aligned(16))) ET s_y[WRITE_Y ? TT : 1][N];

    const size_t seq = ((size_t)dir * p.B + b) * p.H + h;
    const size_t head_off = (size_t)b * T * C + (size_t)h * N;
    const ET *gr = (const ET *)D.r + head_off;
    const ET *gk = (const ET *)D.k + head_off;
    const ET *gv = (const ET *)D.v + head_off;
    const ET *gw = (const ET *)D.w + head_off;
    ET *gy = (ET *)D.y + head_off;

    float st[N];
    if constexpr (WRITE_Y) { if (p.NC > 1) {
        const float *ws = p.ws_state + (seq * p.NC + c) * (size_t)(N * N) + lane;
#pragma unroll
        for (int j = 0; j < N; ++j) st[j] = ws[j * N];
        } else if (D.s_in != nullptr) {
            const float4 *s4 = reinterpret_cast<const float4 *>(D.s_in + (((size_t)b * p.H + h) * N + lane) * N);
#pragma unroll
            for (int j = 0; j < N; j += 4) {
                const float4 q = s4[j / 4];
                st[j] = q.x; st[j + 1] = q.y; st[j + 2] = q.z; st[j + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) st[j] = 0.f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) st[j] = 0.f;
    }
    const float u = WRITE_Y ? E::load((const ET *)D.u + h * N + lane) : 0.f;
    float dprod = 1.f;

    const int s_begin = c * p.L;
    const int s_end = min(T, s_begin + p.L);

    uint4 nr[G::NLD], nk[G::NLD], nv[G::NLD], nw[G::NLD];
#pragma unroll
    for (int q = 0; q < G::NLD; ++q) {
        if constexpr (WRITE_Y) nr[q] = tile_load<ET>(gr, q, lane, s_begin, s_end, T, C, D.reverse);
        nk[q] = tile_load<ET>(gk, q, lane, s_begin, s_end, T, C, D.reverse);
        nv[q] = tile_load<ET>(gv, q, lane, s_begin, s_end, T, C, D.reverse);
        nw[q] = tile_load<ET>(gw, q, lane, s_begin, s_end, T, C, D.reverse);
    }

    for (int s0 = s_begin; s0 < s_end; s0 += TT) {
        __syncthreads();  // the previous tile's LDS readers are done (one wave: just an ordering point)
#pragma unroll
        for (int q = 0; q < G::NLD; ++q) {
            const int tt = q * G::RPL + lane / G::LPR;
            const int col = (lane % G::LPR) * G::EPL;
            float f[G::EPL];
            if constexpr (WRITE_Y) {
                E::unpack(nr[q], f);
#pragma unroll
                for (int e = 0; e < G::EPL; e += 4)
                    *reinterpret_cast<float4 *>(&s_r[tt][col + e]) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
            }
            E::unpack(nk[q], f);
#pragma unroll
            for (int e = 0; e < G::EPL; e += 4)
                *reinterpret_cast<float4 *>(&s_k[tt][col + e]) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
            E::unpack(nv[q], f);
#pragma unroll
            for (int e = 0; e < G::EPL; e += 4)
                *reinterpret_cast<float4 *>(&s_v[tt][col + e]) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
            E::unpack(nw[q], f);
            if (D.wb != nullptr) {
#pragma unroll
                for (int e = 0; e < G::EPL; ++e) f[e] = E::round(f[e] + E::load((const ET *)D.wb + h * N + col + e));
            }
#pragma unroll
            for (int e = 0; e < G::EPL; ++e) f[e] = __expf(-__expf(f[e]));
#pragma unroll
            for (int e = 0; e < G::EPL; e += 4)
                *reinterpret_cast<float4 *>(&s_d[tt][col + e]) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
        }
        // prefetch the next tile while this one is consumed
        const int s1 = s0 + TT;
#pragma unroll
        for (int q = 0; q < G::NLD; ++q) {
            if constexpr (WRITE_Y) nr[q] = tile_load<ET>(gr, q, lane, s1, s_end, T, C, D.reverse);
            nk[q] = tile_load<ET>(gk, q, lane, s1, s_end, T, C, D.reverse);
            nv[q] = tile_load<ET>(gv, q, lane, s1, s_end, T, C, D.reverse);
            nw[q] = tile_load<ET>(gw, q, lane, s1, s_end, T, C, D.reverse);
        }
        __syncthreads();

        const int nt = min(TT, s_end - s0);
#pragma unroll 1
        for (int tt = 0; tt < nt; ++tt) {
            {
                const float v = s_v[tt][lane];
                float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
                float bonus = 0.f;
                if constexpr (WRITE_Y) bonus = wave_sum(s_r[tt][lane] * u * s_k[tt][lane]);
                else dprod *= s_d[tt][lane];
#pragma unroll
                for (int j = 0; j < N; j += 4) {
                    const float4 k4 = *reinterpret_cast<const float4 *>(&s_k[tt][j]);
                    const float4 d4 = *reinterpret_cast<const float4 *>(&s_d[tt][j]);
                    if constexpr (WRITE_Y) {
                        const float4 r4 = *reinterpret_cast<const float4 *>(&s_r[tt][j]);
                        y0 = fmaf(r4.x, st[j], y0);
                        y1 = fmaf(r4.y, st[j + 1], y1);
                        y2 = fmaf(r4.z, st[j + 2], y2);
                        y3 = fmaf(r4.w, st[j + 3], y3);
                    }
                    st[j] = fmaf(st[j], d4.x, k4.x * v);
                    st[j + 1] = fmaf(st[j + 1], d4.y, k4.y * v);
                    st[j + 2] = fmaf(st[j + 2], d4.z, k4.z * v);
                    st[j + 3] = fmaf(st[j + 3], d4.w, k4.w * v);
                    // keep hipcc from hoisting all 48 broadcast reads of a step to its top (192 VGPRs)
                    if ((j & 15) == 12) __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (WRITE_Y) E::store(&s_y[tt][lane], fmaf(bonus, v, (y0 + y1) + (y2 + y3)));
            }
        }
        if constexpr (WRITE_Y) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < G::NLD; ++q) {
                const int tt = q * G::RPL + lane / G::LPR;
                const int col = (lane % G::LPR) * G::EPL;
                const int s = s0 + tt;
                if (s < s_end) {
                    const int t = D.reverse ? (T - 1 - s) : s;
                    *reinterpret_cast<uint4 *>(gy + (size_t)t * C + col) = *reinterpret_cast<const uint4 *>(&s_y[tt][col]);
                }
            }
        }
    }

    if constexpr (!WRITE_Y) {
        float *ws = p.ws_state + (seq * p.NC + c) * (size_t)(N * N) + lane;
#pragma unroll
        for (int j = 0; j < N; ++j) ws[j * N] = st[j];
        p.ws_decay[(seq * p.NC + c) * N + lane] = dprod;
    } else if (p.NC == 1 && D.s_out != nullptr) {
        float4 *s4 = reinterpret_cast<float4 *>(D.s_out + (((size_t)b * p.H + h) * N + lane) * N);
#pragma unroll
        for (int j = 0; j < N; j += 4) s4[j / 4] = make_float4(st[j], st[j + 1], st[j + 2], st[j + 3]);
    }
}

// Pass B: exclusive scan over chunks of (decay, local state); in place: ws_state[c] becomes the state
// entering chunk c.  One thread per state element, 16 blocks of 256 per sequence.
__global__ __launch_bounds__(256) void wkv6_scan_kernel(const FwdParams p) {
    const int e = blockIdx.x * 256 + threadIdx.x;  // 0..4095
    const int j = e >> 6, i = e & 63;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int dir = blockIdx.z;
    const DirArgs &D = p.d[dir];
    const size_t seq = ((size_t)dir * p.B + b) * p.H + h;
    float *ws = p.ws_state + seq * p.NC * (size_t)(N * N) + e;
    const float *wd = p.ws_decay + seq * p.NC * (size_t)N + j;
    const size_t sidx = (((size_t)b * p.H + h) * N + i) * N + j;
    float run = D.s_in ? D.s_in[sidx] : 0.f;
    int c = 0;
    // The recurrence itself is one FMA per chunk; what the kernel waits for is the round trip of its loads (4 waves per CU,
    // the states of pass A still in L2 / the Infinity Cache).  PB chunks are fetched per round trip: with 4 (rounds 1-4) the
    // 128 chunks of the 30-minute shape were 32 dependent round trips = 17.5 us; 16 leaves 8.
    constexpr int PB = 16;
    for (; c + PB <= p.nc_local; c += PB) {
        float loc[PB], dec[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            loc[q] = ws[(size_t)(c + q) * (N * N)];
            dec[q] = wd[(size_t)(c + q) * N];
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            ws[(size_t)(c + q) * (N * N)] = run;
            run = fmaf(run, dec[q], loc[q]);
        }
    }
    for (; c + 4 <= p.nc_local; c += 4) {
        float loc[4], dec[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            loc[q] = ws[(size_t)(c + q) * (N * N)];
            dec[q] = wd[(size_t)(c + q) * N];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ws[(size_t)(c + q) * (N * N)] = run;
            run = fmaf(run, dec[q], loc[q]);
        }
    }
    for (; c < p.nc_local; ++c) {
        const float loc = ws[(size_t)c * (N * N)];
        const float dec = wd[(size_t)c * N];
        ws[(size_t)c * (N * N)] = run;
        run = fmaf(run, dec, loc);
    }
    if (p.nc_local < p.NC) ws[(size_t)(p.NC - 1) * (N * N)] = run;  // last chunk's incoming state
    if (D.s_out) D.s_out[sidx] = run;  // only reached with nc_local == NC
}

#include "wkv6_mfma.inc"

int pick_chunk_len(int B, int T, int H, int ndir) {
    const long seqs = (long)B * H * ndir;
    const long target_waves = 2048;  // 256 CUs x 4 SIMDs x 2 waves
    if (seqs >= target_waves || T <= 2 * TT) return T;
    long nc = (target_waves + seqs - 1) / seqs;
    long L = (T + nc - 1) / nc;
    if (L < 64) L = 64;                      // below this the 64 KB of state traffic per chunk dominates
    L = (L + CHUNK_ALIGN - 1) / CHUNK_ALIGN * CHUNK_ALIGN;
    return L >= T ? T : (int)L;
}

size_t ws_bytes(int B, int T, int H, int ndir, int L) {
    if (L >= T) return 0;
    const size_t NC = (T + L - 1) / L;
    return sizeof(float) * (size_t)ndir * B * H * NC * (N * N + N);
}

// PAFC_WKV6_IMPL=valu selects the register/LDS formulation (wkv6_chunk_kernel); default is the matrix-core one.
// Read per call (no global state); only meant for A/B measurements and tests.
bool use_mfma() {
    const char *e = getenv("PAFC_WKV6_IMPL");
    return !(e && e[0] == 'v');
}

// Pass A (chunk-local end states): bf16 I/O runs the channel-lane kernel; PAFC_WKV6_PASS_A=lt keeps the time-lane one
// (same results up to fp32 rounding; for A/B measurements).
template <typename ET>
void launch_pass_a(const FwdParams &p, dim3 grid, hipStream_t stream) {
    if constexpr (sizeof(ET) == 2) {
        const char *e = getenv("PAFC_WKV6_PASS_A");
        if (!(e && e[0] == 'l')) {
            // the variant without a decay bias issues neither the add nor its rounding (a tenth of the per-step arithmetic)
            FwdParams q = p;
            { const char *o = getenv("PAFC_WKV6_ORDER"); q.order = o ? atoi(o) : 0; }
            if (q.order) grid = dim3(grid.y, grid.x, grid.z);
            if (p.d[0].wb != nullptr || p.d[1].wb != nullptr) hipLaunchKernelGGL(wkv6_pass_a_cl_kernel<true>, grid, dim3(64), 0, stream, q);
            else hipLaunchKernelGGL(wkv6_pass_a_cl_kernel<false>, grid, dim3(64), 0, stream, q);
            return;
        }
    }
    hipLaunchKernelGGL((wkv6_mfma_kernel<ET, false>), grid, dim3(64), 0, stream, p);
}

// Pass C (outputs): bf16 I/O runs the channel-lane kernel; PAFC_WKV6_PASS_C=lt keeps the time-lane one (A/B measurements).
template <typename ET>
void launch_pass_c(const FwdParams &p, dim3 grid, hipStream_t stream) {
    if constexpr (sizeof(ET) == 2) {
        const char *e = getenv("PAFC_WKV6_PASS_C");
        if (!(e && e[0] == 'l')) {
            FwdParams q = p;
            { const char *o = getenv("PAFC_WKV6_ORDER"); q.order = o ? atoi(o) : 0; }
            if (q.order) grid = dim3(grid.y, grid.x, grid.z);
            if (p.d[0].wb != nullptr || p.d[1].wb != nullptr) hipLaunchKernelGGL(wkv6_pass_c_cl_kernel<true>, grid, dim3(64), 0, stream, q);
            else hipLaunchKernelGGL(wkv6_pass_c_cl_kernel<false>, grid, dim3(64), 0, stream, q);
            return;
        }
    }
    hipLaunchKernelGGL((wkv6_mfma_kernel<ET, true>), grid, dim3(64), 0, stream, p);
}

// A short bf16 sequence walked as one chunk (a streaming step: T = 64, 8 heads per stream) has 2-4 blocks of 16 steps and only
// B * H * ndir waves: wkv6_few_blocks_kernel runs its blocks side by side in one launch.  Measured (round 4,
// profiles/r04p_streaming_chunk_step_kernel_stats_few_blocks.csv): 19.3 us per launch against 17.3 us for the serial walk of
// the same chunk by one wave per head -- each wave now pays a cold operand fetch twice (its pass A body, then its pass C body),
// the 16 KiB carried state read per wave and a workgroup barrier, while the serial walk prefetches block n + 1 under block n;
// per-block arithmetic was never the limit at 8 waves on the chip.  Kept behind PAFC_WKV6_FEW=1 (tested), not the default.
template <int NW>
int launch_few_blocks(const FwdParams &p, int ndir, hipStream_t stream) {
    constexpr size_t lds = few_blocks_lds_bytes<NW>();
    auto kern = wkv6_few_blocks_kernel<NW>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PAFC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(1, p.B * p.H, ndir), dim3(64 * NW), lds, stream, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

template <typename ET>
int launch_fwd(FwdParams &p, int ndir, bool any_final, hipStream_t stream) {
    const bool mfma = use_mfma();
    if constexpr (sizeof(ET) == 2) {
        const int nb = (p.T + 15) / 16;
        if (mfma && p.NC == 1 && nb >= 2 && nb <= 4 && (long)p.B * p.H * ndir * nb <= 2048) {
            const char *e = getenv("PAFC_WKV6_FEW"), *c = getenv("PAFC_WKV6_PASS_C");
            if ((e && e[0] == '1') && !(c && c[0] == 'l'))
                return nb == 2 ? launch_few_blocks<2>(p, ndir, stream) : nb == 3 ? launch_few_blocks<3>(p, ndir, stream)
                                                                                  : launch_few_blocks<4>(p, ndir, stream);
        }
    }
    if (p.NC > 1) {
        p.nc_local = any_final ? p.NC : p.NC - 1;
        dim3 ga(p.nc_local, p.B * p.H, ndir);
        if (mfma) launch_pass_a<ET>(p, ga, stream);
        else hipLaunchKernelGGL((wkv6_chunk_kernel<ET, false>), ga, dim3(64), 0, stream, p);
        dim3 gb(16, p.B * p.H, ndir);
        hipLaunchKernelGGL(wkv6_scan_kernel, gb, dim3(256), 0, stream, p);
    }
    dim3 gc(p.NC, p.B * p.H, ndir);
    // pass C raises its wave priority for the second half of every block -- from the level operands through the inter-block term,
    // the intra-block term and the state update (scheme 3) -- and drops it for the loads and decay chains of the next one: the
    // partner wave of the SIMD, which is in the other phase, keeps issuing.  Round 4, first session: scheme 2 (the MFMA-dense tail
    // only) 228.3 -> 221.2 us per bidirectional launch; with the second session's block (fewer MFMAs at the tail, cheaper splits)
    // scheme 3 wins: stand-alone 207-209 (none / scheme 2) -> 197-202, in the model 204 -> 196 us on one box; raising it from the
    // block's start or for the front half only changed nothing.  PAFC_WKV6_PRIO=0 / 2 select none / the tail-only scheme (A/B).
    { const char *e = getenv("PAFC_WKV6_PRIO"); p.prio = e ? atoi(e) : 3; }
    { const char *e = getenv("PAFC_WKV6_REVC"); p.rev_c = e ? atoi(e) : 0; }
    if (mfma) launch_pass_c<ET>(p, gc, stream);
    else hipLaunchKernelGGL((wkv6_chunk_kernel<ET, true>), gc, dim3(64), 0, stream, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

int forward_impl(int dtype, int B, int T, int C, int H, int ndir, const DirArgs *dirs, int chunk_len,
                 void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    if (B <= 0 || T <= 0 || C <= 0 || H <= 0 || C % H != 0) return PAFC_ERR_BAD_DIMS;
    if (C / H != N) return PAFC_ERR_HEAD_SIZE;
    if (dtype != PAFC_F32 && dtype != PAFC_BF16) return PAFC_ERR_DTYPE;
    if ((long)B * H > 65535) return PAFC_ERR_BAD_DIMS;  // grid.y
    bool any_final = false;
    for (int d = 0; d < ndir; ++d) {
        const DirArgs &a = dirs[d];
        if (!a.r || !a.k || !a.v || !a.w || !a.u || !a.y) return PAFC_ERR_NULL_POINTER;
        if ((((uintptr_t)a.r | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.w | (uintptr_t)a.y) & 15) != 0)
            return PAFC_ERR_ALIGNMENT;
        any_final |= a.s_out != nullptr;
    }
    int L = chunk_len > 0 ? chunk_len : pick_chunk_len(B, T, H, ndir);
    if (workspace == nullptr) L = T;
    if (L < T) L = (L + CHUNK_ALIGN - 1) / CHUNK_ALIGN * CHUNK_ALIGN;
    if (L >= T) L = T;
    FwdParams p{};
    for (int d = 0; d < ndir; ++d) p.d[d] = dirs[d];
    p.B = B; p.T = T; p.C = C; p.H = H; p.L = L;
    p.NC = (T + L - 1) / L;
    p.nc_local = 0;
    if (p.NC > 1) {
        const size_t need = ws_bytes(B, T, H, ndir, L);
        if (workspace_bytes < need) return PAFC_ERR_WORKSPACE;
        p.ws_state = (float *)workspace;
        p.ws_decay = p.ws_state + (size_t)ndir * B * H * p.NC * (N * N);
    }
    hipStream_t s = (hipStream_t)stream;
    return dtype == PAFC_BF16 ? launch_fwd<bf16_t>(p, ndir, any_final, s) : launch_fwd<float>(p, ndir, any_final, s);
}


// =====================================================================================================
// Backward (replaces kernel_backward_101/102/103/201, wkv6_cuda.cu:65-263: five serial sweeps) as three
// chunk-parallel sweeps plus a light epilogue.
//
// With S_t the state before step t (S_{t+1} = d_t S_t + k_t v_t^T) and G_t its adjoint
// (G_t = d_t G_{t+1} + r_t gy_t^T, G_T = 0):
//   P_t[j] = sum_i S_t[j][i]     gy_t[i]        "row sweep" in forward time   (a = k, p = v,  q = gy)
//   Q_t[j] = sum_i G_{t+1}[j][i] v_t[i]         "row sweep" in reverse time   (a = r, p = gy, q = v)
//   gv_t[i] = sum_j k_t[j] G_{t+1}[j][i] + (sum_j u r k) gy_t[i]
//           = the FORWARD kernel with r<->k swapped, v := gy, run in reverse time  (pass C reused as is)
//   c_t = v_t . gy_t,  e_t = v_{t-1} . gy_t
//   gr_t = P_t + u k_t c_t        gk_t = Q_t + u r_t c_t        gu += r_t k_t c_t
//   gw_t = Z_t * (-exp(w_t)),  Z_t = Z_{t-1} + k_{t-1} (Q_{t-1} - r_t e_t) - r_t (P_t - k_{t-1} e_t),  Z_0 = 0
// (the last line is the reference's two-sweep sbbbb/sss recursion, wkv6_cuda.cu:211-261, rewritten so that it
// needs only P and Q; gw_0 = gw_{T-1} = 0 as there).  Chunk states for S and G come from the forward's own
// pass A / pass B (k,v,w in forward time; r,gy,w in reverse time).
struct RowArgs {
    const void *a, *p, *q, *w;
    float *out;      // (B, T, C) fp32
    int reverse;
    const float *s_in;   // (B, H, N, N) [value i][key j] initial state or null; only read when NC == 1 (else the scan
                         // has already put it into ws_state[0])
};

struct RowParams {
    RowArgs d[2];
    int B, T, C, H, L, NC;
    const float *ws_state;   // [2][B][H][NC][j][i]; entry c = state entering chunk c (after the scan)
};

// lane j owns ROW j of the state (64 VGPRs over i); a_j, d_j are per-lane, p_i and q_i are broadcast from LDS
template <typename ET>
__global__ __launch_bounds__(64, 3) void wkv6_row_kernel(const RowParams p) {
    using G = TileGeom<ET>;
    using E = Elem<ET>;
    using GO = TileGeom<float>;
    const int c = blockIdx.x;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int dir = blockIdx.z;
    const int lane = threadIdx.x;
    const RowArgs &D = p.d[dir];
    const int T = p.T, C = p.C;

    __shared__ __attribute__((aligned(16))) float s_a[TT][N];
    __shared__ __attribute__((aligned(16))) float s_d[TT][N];
    __shared__ __attribute__((aligned(16))) float s_p[TT][N];
    __shared__ __attribute__((aligned(16))) float s_q[TT][N];
    __shared__ __attribute__((aligned(16))) float s_o[TT][N];

    const size_t head_off = (size_t)b * T * C + (size_t)h * N;
    const ET *ga = (const ET *)D.a + head_off;
    const ET *gp = (const ET *)D.p + head_off;
    const ET *gq = (const ET *)D.q + head_off;
    const ET *gw = (const ET *)D.w + head_off;
    float *go = D.out + head_off;

    float st[N];
    if (p.NC > 1) {
        const size_t seq = ((size_t)dir * p.B + b) * p.H + h;
        const float4 *ws = reinterpret_cast<const float4 *>(p.ws_state + (seq * p.NC + c) * (size_t)(N * N) + lane * N);
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            const float4 v4 = ws[i / 4];
            st[i] = v4.x; st[i + 1] = v4.y; st[i + 2] = v4.z; st[i + 3] = v4.w;
        }
    } else if (D.s_in != nullptr) {
        const float *si = D.s_in + ((size_t)b * p.H + h) * (size_t)(N * N) + lane;   // lane j: column j of [i][j]
#pragma unroll
        for (int i = 0; i < N; ++i) st[i] = si[(size_t)i * N];
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) st[i] = 0.f;
    }
    const int s_begin = c * p.L;
    const int s_end = min(T, s_begin + p.L);

    uint4 na[G::NLD], np[G::NLD], nq[G::NLD], nw[G::NLD];
#pragma unroll
    for (int q = 0; q < G::NLD; ++q) {
        na[q] = tile_load<ET>(ga, q, lane, s_begin, s_end, T, C, D.reverse);
        np[q] = tile_load<ET>(gp, q, lane, s_begin, s_end, T, C, D.reverse);
        nq[q] = tile_load<ET>(gq, q, lane, s_begin, s_end, T, C, D.reverse);
        nw[q] = tile_load<ET>(gw, q, lane, s_begin, s_end, T, C, D.reverse);
    }
    for (int s0 = s_begin; s0 < s_end; s0 += TT) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < G::NLD; ++q) {
            const int tt = q * G::RPL + lane / G::LPR;
            const int col = (lane % G::LPR) * G::EPL;
            float f[G::EPL];
            E::unpack(na[q], f);
#pragma unroll
            for (int e = 0; e < G::EPL; e += 4)
                *reinterpret_cast<float4 *>(&s_a[tt][col + e]) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
            E::unpack(np[q], f);
#pragma unroll
            for (int e = 0; e < G::EPL; e += 4)
                *reinterpret_cast<float4 *>(&s_p[tt][col + e]) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
            E::unpack(nq[q], f);
#pragma unroll
            for (int e = 0; e < G::EPL; e += 4)
                *reinterpret_cast<float4 *>(&s_q[tt][col + e]) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
            E::unpack(nw[q], f);
#pragma unroll
            for (int e = 0; e < G::EPL; ++e) f[e] = __expf(-__expf(f[e]));
#pragma unroll
            for (int e = 0; e < G::EPL; e += 4)
                *reinterpret_cast<float4 *>(&s_d[tt][col + e]) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
        }
        const int s1 = s0 + TT;
#pragma unroll
        for (int q = 0; q < G::NLD; ++q) {
            na[q] = tile_load<ET>(ga, q, lane, s1, s_end, T, C, D.reverse);
            np[q] = tile_load<ET>(gp, q, lane, s1, s_end, T, C, D.reverse);
            nq[q] = tile_load<ET>(gq, q, lane, s1, s_end, T, C, D.reverse);
            nw[q] = tile_load<ET>(gw, q, lane, s1, s_end, T, C, D.reverse);
        }
        __syncthreads();
        const int nt = min(TT, s_end - s0);
#pragma unroll 1
        for (int tt = 0; tt < nt; ++tt) {
            const float aj = s_a[tt][lane], dj = s_d[tt][lane];
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
            for (int i = 0; i < N; i += 4) {
                const float4 p4 = *reinterpret_cast<const float4 *>(&s_p[tt][i]);
                const float4 q4 = *reinterpret_cast<const float4 *>(&s_q[tt][i]);
                o0 = fmaf(q4.x, st[i], o0);
                o1 = fmaf(q4.y, st[i + 1], o1);
                o2 = fmaf(q4.z, st[i + 2], o2);
                o3 = fmaf(q4.w, st[i + 3], o3);
                st[i] = fmaf(st[i], dj, aj * p4.x);
                st[i + 1] = fmaf(st[i + 1], dj, aj * p4.y);
                st[i + 2] = fmaf(st[i + 2], dj, aj * p4.z);
                st[i + 3] = fmaf(st[i + 3], dj, aj * p4.w);
                if ((i & 15) == 12) __builtin_amdgcn_sched_barrier(0);
            }
            s_o[tt][lane] = (o0 + o1) + (o2 + o3);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < GO::NLD; ++q) {
            const int tt = q * GO::RPL + lane / GO::LPR;
            const int col = (lane % GO::LPR) * GO::EPL;
            const int s = s0 + tt;
            if (s < s_end) {
                const int t = D.reverse ? (T - 1 - s) : s;
                *reinterpret_cast<float4 *>(go + (size_t)t * C + col) = *reinterpret_cast<const float4 *>(&s_o[tt][col]);
            }
        }
    }
}

struct EpiParams {
    const void *r, *k, *v, *w, *u, *gy;
    float *P;            // in: P_t; out (in place): the chunk-local running sum of the gw recursion's terms
    const float *Q;
    void *gr, *gk, *gw, *gu;
    float *tot, *gup;    // (B*H, NE, 64): per-chunk totals of the recursion terms / of the gu sums
    int B, T, C, H, reverse, TCH, NE;
    const float *s_in, *gs;   // initial state and its adjoint, (B, H, N, N) [value i][key j], or null (zero state)
};

constexpr int EPI_MAX_CHUNKS = 64;
__host__ __device__ inline int epi_chunk_steps(int T) {
    const int t = (T + EPI_MAX_CHUNKS - 1) / EPI_MAX_CHUNKS;
    return t < 16 ? 16 : (t + 7) / 8 * 8;
}

// Epilogue, pass 1: one wave per (time chunk, b, h), lane = channel.  Everything here is independent across steps
// except Z, a plain running sum over time of per-step terms -- so the time axis is cut into <= 64 chunks, a wave
// leaves its chunk-LOCAL running sums in P's slots and its total in `tot`; pass 2 adds the totals of the earlier
// chunks.  (The serial form -- one wave per (b, h) walking all T steps one load-latency at a time -- took 290 us for
// 32 x 500 steps; the loads of a group of steps are now in flight together.)
template <typename ET>
__global__ __launch_bounds__(64) void wkv6_bwd_epilogue_kernel(const EpiParams p) {
    using E = Elem<ET>;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int lane = threadIdx.x;
    const int T = p.T, C = p.C;
    const size_t base = (size_t)b * T * C + (size_t)h * N + lane;
    const ET *r = (const ET *)p.r + base, *k = (const ET *)p.k + base, *v = (const ET *)p.v + base;
    const ET *gy = (const ET *)p.gy + base;
    float *P = p.P + base;
    const float *Q = p.Q + base;
    ET *gr = (ET *)p.gr + base, *gk = (ET *)p.gk + base;
    const float u = E::load((const ET *)p.u + h * N + lane);
    auto at = [&](int s) { return (size_t)(p.reverse ? (T - 1 - s) : s) * C; };
    const int s0 = blockIdx.x * p.TCH, s1 = min(T, s0 + p.TCH);

    float gu = 0.f, Z = 0.f, k_prev = 0.f, v_prev = 0.f, q_prev = 0.f;
    float z0 = 0.f;   // with an initial state: Z_0 = <G_0, S_0>[j] - r_0 P_0  (G_0 = gs; zero state: 0 - 0)
    if (s0 == 0 && p.s_in != nullptr) {
        const size_t so = ((size_t)b * p.H + h) * (size_t)(N * N) + lane;
        for (int i = 0; i < N; ++i) z0 = fmaf(p.gs[so + (size_t)i * N], p.s_in[so + (size_t)i * N], z0);
    }
    if (s0 > 0 && s0 < T) {
        const size_t o = at(s0 - 1);
        k_prev = E::load(k + o); v_prev = E::load(v + o); q_prev = Q[o];
    }
    constexpr int U = 4;
    for (int sb = s0; sb < s1; sb += U) {
        float rr[U], kk[U], vv[U], gg[U], Ps[U], Qs[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const size_t o = at(min(sb + i, s1 - 1));
            rr[i] = E::load(r + o); kk[i] = E::load(k + o); vv[i] = E::load(v + o); gg[i] = E::load(gy + o);
            Ps[i] = P[o]; Qs[i] = Q[o];
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int s = sb + i;
            if (s < s1) {       // wave-uniform
                const size_t o = at(s);
                const float c = wave_sum(vv[i] * gg[i]);
                const float e = wave_sum(v_prev * gg[i]);
                E::store(gr + o, fmaf(u * kk[i], c, Ps[i]));
                E::store(gk + o, fmaf(u * rr[i], c, Qs[i]));
                gu = fmaf(rr[i] * kk[i], c, gu);
                if (s > 0) Z += k_prev * (q_prev - rr[i] * e) - rr[i] * (Ps[i] - k_prev * e);
                else if (p.s_in != nullptr) Z = z0 - rr[i] * Ps[i];
                P[o] = Z;
                k_prev = kk[i]; v_prev = vv[i]; q_prev = Qs[i];
            }
        }
    }
    const size_t slot = ((size_t)blockIdx.y * p.NE + blockIdx.x) * N + lane;
    p.tot[slot] = Z;
    p.gup[slot] = gu;
}

// Epilogue, pass 2: gw_t = (sum of the earlier chunks' totals + local running sum) * (-exp(w_t)); gu summed over chunks.
template <typename ET>
__global__ __launch_bounds__(64) void wkv6_bwd_gw_kernel(const EpiParams p) {
    using E = Elem<ET>;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int lane = threadIdx.x;
    const int T = p.T, C = p.C;
    const size_t base = (size_t)b * T * C + (size_t)h * N + lane;
    const ET *w = (const ET *)p.w + base;
    const float *Zl = p.P + base;
    ET *gw = (ET *)p.gw + base;
    auto at = [&](int s) { return (size_t)(p.reverse ? (T - 1 - s) : s) * C; };
    const int s0 = blockIdx.x * p.TCH, s1 = min(T, s0 + p.TCH);
    const float *tot = p.tot + (size_t)blockIdx.y * p.NE * N + lane;
    float off = 0.f;
    for (int c = 0; c < (int)blockIdx.x; ++c) off += tot[(size_t)c * N];
    if (blockIdx.x == 0) {
        const float *gp = p.gup + (size_t)blockIdx.y * p.NE * N + lane;
        float g = 0.f;
        for (int c = 0; c < p.NE; ++c) g += gp[(size_t)c * N];
        E::store((ET *)p.gu + (size_t)b * C + h * N + lane, g);
    }
    constexpr int U = 8;
    for (int sb = s0; sb < s1; sb += U) {
        float ww[U], zz[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const size_t o = at(min(sb + i, s1 - 1));
            ww[i] = E::load(w + o); zz[i] = Zl[o];
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int s = sb + i;
            // zero state: gw_0 = 0 (nothing to decay); always gw_{T-1} = 0 (nothing after it), as the reference stores them
            if (s < s1)
                E::store(gw + at(s), ((s == 0 && p.s_in == nullptr) || s == T - 1) ? 0.f : (off + zz[i]) * -__expf(ww[i]));
        }
    }
}

size_t bwd_ws_bytes(int B, int T, int C, int H, int L) {
    size_t n = 2 * (size_t)B * T * C;                      // P, Q
    n += 2 * (size_t)B * H * EPI_MAX_CHUNKS * N;           // epilogue: per-chunk totals (gw recursion, gu)
    n += (size_t)B * H * N * N;                            // adjoint of the initial state when the caller passes no gs
    if (L < T) {
        const size_t NC = (T + L - 1) / L;
        n += 2 * (size_t)B * H * NC * (N * N + N);          // chunk states + decays of S and G
    }
    return n * sizeof(float);
}

int bwd_chunk_len(int B, int T, int H, int chunk_len) {
    int L = chunk_len > 0 ? chunk_len : pick_chunk_len(B, T, H, 2);
    if (L < T) L = (L + CHUNK_ALIGN - 1) / CHUNK_ALIGN * CHUNK_ALIGN;
    return L >= T ? T : L;
}

template <typename ET>
int launch_bwd(int B, int T, int C, int H, const void *r, const void *k, const void *v, const void *w, const void *u,
               const void *gy, void *gr, void *gk, void *gv, void *gw, void *gu, const float *s_in, float *gs,
               int reverse, int L, float *ws, hipStream_t stream) {
    const int rev = reverse ? 1 : 0;
    const int NC = (T + L - 1) / L;
    float *P = ws, *Q = P + (size_t)B * T * C;
    float *tot = Q + (size_t)B * T * C, *gup = tot + (size_t)B * H * EPI_MAX_CHUNKS * N;
    float *gs_scratch = gup + (size_t)B * H * EPI_MAX_CHUNKS * N;
    float *ws_state = gs_scratch + (size_t)B * H * N * N;
    if (s_in != nullptr && gs == nullptr) gs = gs_scratch;      // the gw recursion starts from <gs, s_in>
    float *ws_decay = ws_state + 2 * (size_t)B * H * NC * (N * N);
    // the matrix-core forward kernels want 16-byte aligned operands (forward_impl checks the same)
    const bool mfma = use_mfma() &&
                      ((((uintptr_t)r | (uintptr_t)k | (uintptr_t)v | (uintptr_t)w | (uintptr_t)gy | (uintptr_t)gv) & 15) == 0);
    // dir 0: S from (k, v, w) in forward time; dir 1: G from (r, gy, w) in reverse time, whose pass C is gv
    FwdParams fp{};
    fp.d[0] = DirArgs{k, k, v, w, u, gv, s_in, nullptr, rev, nullptr};               // r, y unused by pass A
    fp.d[1] = DirArgs{k, r, gy, w, u, gv, nullptr, gs, 1 - rev, nullptr};            // forward kernel with r<->k, v := gy
    fp.B = B; fp.T = T; fp.C = C; fp.H = H; fp.L = L; fp.NC = NC;
    fp.nc_local = gs ? NC : NC - 1;     // the adjoint of the initial state is the FINAL state of the reverse-time sweep
    fp.ws_state = ws_state; fp.ws_decay = ws_decay;
    if (NC > 1) {
        if (mfma) launch_pass_a<ET>(fp, dim3(fp.nc_local, B * H, 2), stream);
        else hipLaunchKernelGGL((wkv6_chunk_kernel<ET, false>), dim3(fp.nc_local, B * H, 2), dim3(64), 0, stream, fp);
        hipLaunchKernelGGL(wkv6_scan_kernel, dim3(16, B * H, 2), dim3(256), 0, stream, fp);
    }
    FwdParams fg = fp;  // pass C for direction 1 only: present it as direction 0 of a one-direction launch
    fg.d[0] = fp.d[1];
    fg.ws_state = ws_state + (size_t)B * H * NC * (N * N);
    fg.ws_decay = ws_decay + (size_t)B * H * NC * N;
    if (mfma) launch_pass_c<ET>(fg, dim3(NC, B * H, 1), stream);
    else hipLaunchKernelGGL((wkv6_chunk_kernel<ET, true>), dim3(NC, B * H, 1), dim3(64), 0, stream, fg);

    RowParams rp{};
    rp.d[0] = RowArgs{k, v, gy, w, P, rev, s_in};
    rp.d[1] = RowArgs{r, gy, v, w, Q, 1 - rev, nullptr};
    rp.B = B; rp.T = T; rp.C = C; rp.H = H; rp.L = L; rp.NC = NC;
    rp.ws_state = ws_state;
    hipLaunchKernelGGL(wkv6_row_kernel<ET>, dim3(NC, B * H, 2), dim3(64), 0, stream, rp);

    EpiParams ep{r, k, v, w, u, gy, P, Q, gr, gk, gw, gu, tot, gup, B, T, C, H, rev, 0, 0, s_in, gs};
    ep.TCH = epi_chunk_steps(T);
    ep.NE = (T + ep.TCH - 1) / ep.TCH;
    hipLaunchKernelGGL(wkv6_bwd_epilogue_kernel<ET>, dim3(ep.NE, B * H), dim3(64), 0, stream, ep);
    hipLaunchKernelGGL(wkv6_bwd_gw_kernel<ET>, dim3(ep.NE, B * H), dim3(64), 0, stream, ep);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // namespace
}  // namespace pafc

using pafc::DirArgs;

extern "C" {

int pafc_abi_version(void) { return 1; }

int pafc_selftest_lane_ops(float *out_2x64x4, pafc_stream_t stream) {
    if (!out_2x64x4) return PAFC_ERR_NULL_POINTER;
    hipLaunchKernelGGL(pafc::lane_ops_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_2x64x4);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

int pafc_wkv6_pick_chunk_len(int B, int T, int C, int H, int ndir) {
    if (B <= 0 || T <= 0 || H <= 0 || ndir <= 0) return 0;
    (void)C;
    return pafc::pick_chunk_len(B, T, H, ndir);
}

size_t pafc_wkv6_fwd_workspace_bytes(int B, int T, int C, int H, int ndir, int chunk_len) {
    if (B <= 0 || T <= 0 || H <= 0 || ndir <= 0) return 0;
    (void)C;
    int L = chunk_len > 0 ? chunk_len : pafc::pick_chunk_len(B, T, H, ndir);
    if (L < T) L = (L + pafc::CHUNK_ALIGN - 1) / pafc::CHUNK_ALIGN * pafc::CHUNK_ALIGN;
    return pafc::ws_bytes(B, T, H, ndir, L);
}

int pafc_wkv6_forward_state(int dtype, int B, int T, int C, int H, const void *r, const void *k, const void *v,
                            const void *w, const void *u, void *y, const float *s_in, float *s_out, int reverse,
                            int chunk_len, void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    DirArgs d{r, k, v, w, u, y, s_in, s_out, reverse ? 1 : 0, nullptr};
    return pafc::forward_impl(dtype, B, T, C, H, 1, &d, chunk_len, workspace, workspace_bytes, stream);
}

int pafc_wkv6_forward_bf16(int B, int T, int C, int H, const void *r, const void *k, const void *v, const void *w,
                           const void *u, void *y, int chunk_len, void *workspace, size_t workspace_bytes,
                           pafc_stream_t stream) {
    return pafc_wkv6_forward_state(PAFC_BF16, B, T, C, H, r, k, v, w, u, y, nullptr, nullptr, 0, chunk_len, workspace,
                                   workspace_bytes, stream);
}

int pafc_wkv6_forward_f32(int B, int T, int C, int H, const void *r, const void *k, const void *v, const void *w,
                          const void *u, void *y, int chunk_len, void *workspace, size_t workspace_bytes,
                          pafc_stream_t stream) {
    return pafc_wkv6_forward_state(PAFC_F32, B, T, C, H, r, k, v, w, u, y, nullptr, nullptr, 0, chunk_len, workspace,
                                   workspace_bytes, stream);
}

int pafc_wkv6_forward_bidir_wbias(int dtype, int B, int T, int C, int H, const void *r_f, const void *k_f,
                                  const void *v_f, const void *w_f, const void *u_f, const void *wb_f, void *y_f,
                                  const void *r_b, const void *k_b, const void *v_b, const void *w_b, const void *u_b,
                                  const void *wb_b, void *y_b, int chunk_len, void *workspace, size_t workspace_bytes,
                                  pafc_stream_t stream);

int pafc_wkv6_forward_bidir(int dtype, int B, int T, int C, int H, const void *r_f, const void *k_f, const void *v_f,
                            const void *w_f, const void *u_f, void *y_f, const void *r_b, const void *k_b,
                            const void *v_b, const void *w_b, const void *u_b, void *y_b, int chunk_len,
                            void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    return pafc_wkv6_forward_bidir_wbias(dtype, B, T, C, H, r_f, k_f, v_f, w_f, u_f, nullptr, y_f, r_b, k_b, v_b, w_b,
                                         u_b, nullptr, y_b, chunk_len, workspace, workspace_bytes, stream);
}

int pafc_wkv6_forward_bidir_wbias(int dtype, int B, int T, int C, int H, const void *r_f, const void *k_f,
                                  const void *v_f, const void *w_f, const void *u_f, const void *wb_f, void *y_f,
                                  const void *r_b, const void *k_b, const void *v_b, const void *w_b, const void *u_b,
                                  const void *wb_b, void *y_b, int chunk_len, void *workspace, size_t workspace_bytes,
                                  pafc_stream_t stream) {
    DirArgs d[2] = {{r_f, k_f, v_f, w_f, u_f, y_f, nullptr, nullptr, 0, wb_f}, {r_b, k_b, v_b, w_b, u_b, y_b, nullptr, nullptr, 1, wb_b}};
    return pafc::forward_impl(dtype, B, T, C, H, 2, d, chunk_len, workspace, workspace_bytes, stream);
}

}  // extern "C"

extern "C" {

size_t pafc_wkv6_bwd_workspace_bytes(int B, int T, int C, int H, int chunk_len) {
    if (B <= 0 || T <= 0 || C <= 0 || H <= 0) return 0;
    return pafc::bwd_ws_bytes(B, T, C, H, pafc::bwd_chunk_len(B, T, H, chunk_len));
}

int pafc_wkv6_backward_state(int dtype, int B, int T, int C, int H, const void *r, const void *k, const void *v,
                             const void *w, const void *u, const float *s_in, const void *gy, void *gr, void *gk, void *gv,
                             void *gw, void *gu, float *gs, int reverse, int chunk_len, void *workspace,
                             size_t workspace_bytes, pafc_stream_t stream) {
    if (B <= 0 || T <= 0 || C <= 0 || H <= 0 || C % H != 0 || (long)B * H > 65535) return PAFC_ERR_BAD_DIMS;
    if (C / H != pafc::N) return PAFC_ERR_HEAD_SIZE;
    if (!r || !k || !v || !w || !u || !gy || !gr || !gk || !gv || !gw || !gu || !workspace) return PAFC_ERR_NULL_POINTER;
    const int L = pafc::bwd_chunk_len(B, T, H, chunk_len);
    if (workspace_bytes < pafc::bwd_ws_bytes(B, T, C, H, L)) return PAFC_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAFC_BF16)
        return pafc::launch_bwd<pafc::bf16_t>(B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu, s_in, gs, reverse, L,
                                              (float *)workspace, s);
    if (dtype == PAFC_F32)
        return pafc::launch_bwd<float>(B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu, s_in, gs, reverse, L,
                                       (float *)workspace, s);
    return PAFC_ERR_DTYPE;
}

int pafc_wkv6_backward(int dtype, int B, int T, int C, int H, const void *r, const void *k, const void *v,
                       const void *w, const void *u, const void *gy, void *gr, void *gk, void *gv, void *gw, void *gu,
                       int reverse, int chunk_len, void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    return pafc_wkv6_backward_state(dtype, B, T, C, H, r, k, v, w, u, nullptr, gy, gr, gk, gv, gw, gu, nullptr, reverse,
                                    chunk_len, workspace, workspace_bytes, stream);
}

}  // extern "C"
