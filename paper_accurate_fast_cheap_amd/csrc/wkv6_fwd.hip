// WKV-6 forward for gfx950: T-parallel three-pass chunked scan, both directions in one grid.
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
#include "pafc_common.h"
#include "../../include/pafc_wkv6.h"

namespace pafc {
namespace {

constexpr int N = 64;   // head size
constexpr int TT = 8;   // time steps per LDS tile

struct DirArgs {
    const void *r, *k, *v, *w, *u;
    void *y;
    const float *s_in;
    float *s_out;
    int reverse;
};

struct FwdParams {
    DirArgs d[2];
    int B, T, C, H;
    int L, NC;           // chunk length (multiple of TT unless == T) and number of chunks
    int nc_local;        // chunks whose local state pass A must produce (NC-1, or NC when a final state is wanted)
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

int pick_chunk_len(int B, int T, int H, int ndir) {
    const long seqs = (long)B * H * ndir;
    const long target_waves = 2048;  // 256 CUs x 4 SIMDs x 2 waves
    if (seqs >= target_waves || T <= 2 * TT) return T;
    long nc = (target_waves + seqs - 1) / seqs;
    long L = (T + nc - 1) / nc;
    if (L < 64) L = 64;                      // below this the 64 KB of state traffic per chunk dominates
    L = (L + TT - 1) / TT * TT;
    return L >= T ? T : (int)L;
}

size_t ws_bytes(int B, int T, int H, int ndir, int L) {
    if (L >= T) return 0;
    const size_t NC = (T + L - 1) / L;
    return sizeof(float) * (size_t)ndir * B * H * NC * (N * N + N);
}

template <typename ET>
int launch_fwd(FwdParams &p, int ndir, bool any_final, hipStream_t stream) {
    const dim3 grid_bh(1, p.B * p.H, ndir);
    if (p.NC > 1) {
        p.nc_local = any_final ? p.NC : p.NC - 1;
        dim3 ga(p.nc_local, p.B * p.H, ndir);
        hipLaunchKernelGGL((wkv6_chunk_kernel<ET, false>), ga, dim3(64), 0, stream, p);
        dim3 gb(16, p.B * p.H, ndir);
        hipLaunchKernelGGL(wkv6_scan_kernel, gb, dim3(256), 0, stream, p);
    }
    dim3 gc(p.NC, p.B * p.H, ndir);
    hipLaunchKernelGGL((wkv6_chunk_kernel<ET, true>), gc, dim3(64), 0, stream, p);
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
        any_final |= a.s_out != nullptr;
    }
    int L = chunk_len > 0 ? chunk_len : pick_chunk_len(B, T, H, ndir);
    if (workspace == nullptr) L = T;
    if (L < T) L = (L + TT - 1) / TT * TT;
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

}  // namespace
}  // namespace pafc

using pafc::DirArgs;

extern "C" {

int pafc_abi_version(void) { return 1; }

int pafc_wkv6_pick_chunk_len(int B, int T, int C, int H, int ndir) {
    if (B <= 0 || T <= 0 || H <= 0 || ndir <= 0) return 0;
    (void)C;
    return pafc::pick_chunk_len(B, T, H, ndir);
}

size_t pafc_wkv6_fwd_workspace_bytes(int B, int T, int C, int H, int ndir, int chunk_len) {
    if (B <= 0 || T <= 0 || H <= 0 || ndir <= 0) return 0;
    (void)C;
    int L = chunk_len > 0 ? chunk_len : pafc::pick_chunk_len(B, T, H, ndir);
    if (L < T) L = (L + pafc::TT - 1) / pafc::TT * pafc::TT;
    return pafc::ws_bytes(B, T, H, ndir, L);
}

int pafc_wkv6_forward_state(int dtype, int B, int T, int C, int H, const void *r, const void *k, const void *v,
                            const void *w, const void *u, void *y, const float *s_in, float *s_out, int reverse,
                            int chunk_len, void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    DirArgs d{r, k, v, w, u, y, s_in, s_out, reverse ? 1 : 0};
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

int pafc_wkv6_forward_bidir(int dtype, int B, int T, int C, int H, const void *r_f, const void *k_f, const void *v_f,
                            const void *w_f, const void *u_f, void *y_f, const void *r_b, const void *k_b,
                            const void *v_b, const void *w_b, const void *u_b, void *y_b, int chunk_len,
                            void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    DirArgs d[2] = {{r_f, k_f, v_f, w_f, u_f, y_f, nullptr, nullptr, 0}, {r_b, k_b, v_b, w_b, u_b, y_b, nullptr, nullptr, 1}};
    return pafc::forward_impl(dtype, B, T, C, H, 2, d, chunk_len, workspace, workspace_bytes, stream);
}

}  // extern "C"
