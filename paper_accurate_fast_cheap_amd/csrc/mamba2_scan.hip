// Mamba-2 selective scan (SSD) on the bf16 matrix cores, chunk-parallel (C ABI: include/pafc_encoder_ops.h: pafc_mamba2_scan).
//
// PARITY UNPINNED (third-party arithmetic, see mamba2.hip / transformer/mamba2.py).  Per head h (head dim P = 64, state
// dim N = 128, B_t and C_t shared by all heads, scalar decay a_t = exp(dt_t A_h)):
//     h_t = a_t h_{t-1} + dt_t B_t x_t^T          (N x P state)          y_t = C_t . h_t
// Running this on the WKV-6 kernel means broadcasting C, B, the decay over heads / channels into fp32 operand planes
// (1.1 GB per direction and layer at the 30-minute shape) and two fp32 scans of 1024 channels.  Here the structure of the
// SSD is used directly.  A chunk is walked in blocks of 16 steps; with H the state entering the block and
// cum_t = sum_{tau <= t} log a_tau inside the block:
//     y_t = e^{cum_t} C_t . H  +  sum_{s <= t} e^{cum_t - cum_s} dt_s (C_t . B_s) x_s
//     H'  = e^{cum_15} H + sum_s e^{cum_15 - cum_s} dt_s B_s x_s^T
// -- the decay is a scalar per step, so it factors OUT of the channel contractions: C B^T is one exact bf16 product per
// block (4 MFMAs) masked by a 16 x 16 scalar matrix, no hierarchical decay levels, almost no VALU per channel.
// All exponents are <= 0.  Same three-pass chunk schedule as the WKV-6 scan (pass A chunk-local end states, pass B scan
// over chunks, pass C outputs), one wave64 per (chunk, batch x head); operand layouts and the selection-matrix
// re-layouts are those of wkv6_mfma.inc (state tile [jm][in]: lane (i = l & 15, qq = l >> 4), reg g <-> H[16 jm + 4 qq + g]
// [16 in + i]; "Lt": lane (t = l & 15, q = l >> 4) holds channels 16 m + 4 q + g of row t).  bf16 inputs (x, B, C exact),
// fp32 state, split hi + lo operands wherever an fp32 quantity enters an MFMA.
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc {
namespace {

constexpr int SN = 128, SP = 64, SBL = 16;
typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));
typedef float sf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sbf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ sf32x4 smfma(su32x4 a, su32x4 b, sf32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sbf16x8, a), __builtin_bit_cast(sbf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned spack_exact(float lo, float hi) {   // two floats that ARE bf16 values
    return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}
__device__ __forceinline__ unsigned scvt_pk(float a, float b) {
    const sf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sbf16x2));
}
struct SHiLo { unsigned hi, lo; };
__device__ __forceinline__ SHiLo ssplit_pk(float a, float b) {          // (a, b) = hi + lo, packed bf16 pairs
    SHiLo r;
    r.hi = scvt_pk(a, b);
    r.lo = scvt_pk(a - __uint_as_float(r.hi << 16), b - __uint_as_float(r.hi & 0xffff0000u));
    return r;
}
template <int CTRL>
__device__ __forceinline__ float row_shr_zero(float x) {                 // lane t <- lane t - n of its 16-lane row, 0 outside
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

struct SsdParams {
    const bf16_t *xbc;     // (B, L, ldx): [x (d_inner) | B (128) | C (128)]
    long ldx;
    const float *dt, *la;  // (B, L, H): softplus(dt_raw + dt_bias), log a = dt * A
    float *y;              // (B, L, d_inner) fp32, or null when y16 is set
    bf16_t *y16;           // (B, L, d_inner) bf16 = round(scan + D[h] * x): what mamba_ssm's scan returns (skip term inside)
    const float *Dskip;    // (H), with y16
    int B, L, H, d_inner, Lc, NC, nc_local;
    int reverse;           // 1: step s of the recurrence is time index L - 1 - s (the right-to-left direction, un-flipped I/O)
    float *ws_state;       // [B][H][NC][128][64]
    float *ws_decay;       // [B][H][NC]
};

template <bool WRITE_Y, bool Y16 = false>
__global__ __launch_bounds__(64, 2) void mamba2_ssd_kernel(const SsdParams p) {
    const int c = blockIdx.x;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int lane = threadIdx.x, t16 = lane & 15, q = lane >> 4;
    __shared__ float s_cum[SBL], s_dt[SBL];
    __shared__ __attribute__((aligned(16))) float s_y[WRITE_Y ? SBL : 1][SP + 4];
    __shared__ __attribute__((aligned(16))) bf16_t s_x[(WRITE_Y && Y16) ? SBL : 1][SP + 8];   // x of the block, [step][channel], for the skip term

    // 0/1 selection operands of the layout-changing MFMAs (as in wkv6_mfma.inc)
    su32x4 selA = {0u, 0u, 0u, 0u}, selB = {0u, 0u, 0u, 0u};
    {
        const int e = t16 - 4 * q;
        const unsigned one_lo = 0x3f80u, one_hi = 0x3f800000u;
        if (e == 0) { selA[0] = one_lo; selB[2] = one_lo; }
        if (e == 1) { selA[0] = one_hi; selB[2] = one_hi; }
        if (e == 2) { selA[1] = one_lo; selB[3] = one_lo; }
        if (e == 3) { selA[1] = one_hi; selB[3] = one_hi; }
    }
    const sf32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    const size_t seq = (size_t)b * p.H + h;
    sf32x4 S[8][4];
    {
        const float *src = (WRITE_Y && p.NC > 1) ? p.ws_state + (seq * p.NC + c) * (size_t)(SN * SP) : nullptr;
#pragma unroll
        for (int jm = 0; jm < 8; ++jm)
#pragma unroll
            for (int in = 0; in < 4; ++in)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    S[jm][in][g] = src ? src[(16 * jm + 4 * q + g) * SP + 16 * in + t16] : 0.f;
    }
    float lsum = 0.f;   // log of the chunk's decay product (pass A output)
    float dk = 0.f;     // D[h] of the skip term, read once (a load inside the store loop would be re-issued every pass)
    if constexpr (WRITE_Y && Y16) dk = p.Dskip[h];

    const int s_begin = c * p.Lc, s_end = min(p.L, s_begin + p.Lc);
    const bf16_t *xb = p.xbc + (size_t)b * p.L * p.ldx;
    for (int s0 = s_begin; s0 < s_end; s0 += SBL) {
        const bool live = s0 + t16 < s_end;
        const int sstep = min(s0 + t16, s_end - 1);
        const int srow = p.reverse ? p.L - 1 - sstep : sstep;    // time index of this lane's step
        const bf16_t *row = xb + (size_t)srow * p.ldx;
        // ---- loads (Lt layout): x of this head, B and C of the step --------------------------------------------
        uint2 xr[4], Br[8], Cr[WRITE_Y ? 8 : 1];
#pragma unroll
        for (int in = 0; in < 4; ++in) {
            xr[in] = *reinterpret_cast<const uint2 *>(row + h * SP + 16 * in + 4 * q);
            if (!live) xr[in] = make_uint2(0u, 0u);              // padded step: x = 0, dt = 0, a = 1
        }
#pragma unroll
        for (int jm = 0; jm < 8; ++jm) {
            Br[jm] = *reinterpret_cast<const uint2 *>(row + p.d_inner + 16 * jm + 4 * q);
            if constexpr (WRITE_Y) Cr[jm] = *reinterpret_cast<const uint2 *>(row + p.d_inner + SN + 16 * jm + 4 * q);
        }
        const size_t sidx = ((size_t)b * p.L + srow) * p.H + h;
        const float dt_own = live ? p.dt[sidx] : 0.f;
        const float la_own = live ? p.la[sidx] : 0.f;
        // inclusive prefix sum of log a over the 16 steps (one DPP row = the 16 lanes that share q)
        float cum = la_own;
        cum += row_shr_zero<0x111>(cum);
        cum += row_shr_zero<0x112>(cum);
        cum += row_shr_zero<0x114>(cum);
        cum += row_shr_zero<0x118>(cum);
        __syncthreads();                                          // previous block's readers of the tables are done
        if (q == 0) { s_cum[t16] = cum; s_dt[t16] = dt_own; }
        if constexpr (WRITE_Y && Y16) {
#pragma unroll
            for (int in = 0; in < 4; ++in) *reinterpret_cast<uint2 *>(&s_x[t16][16 * in + 4 * q]) = xr[in];
        }
        __syncthreads();
        const float c15 = s_cum[15];
        const float4 cs4 = *reinterpret_cast<const float4 *>(&s_cum[4 * q]);
        const float4 ds4 = *reinterpret_cast<const float4 *>(&s_dt[4 * q]);
        const float cs[4] = {cs4.x, cs4.y, cs4.z, cs4.w}, ds[4] = {ds4.x, ds4.y, ds4.z, ds4.w};
        const float e15 = __expf(c15);

        // ---- x re-laid out with time as the contraction index (lane = channel, slot = step): selection MFMAs ------
        su32x4 xT[4];
#pragma unroll
        for (int ip = 0; ip < 4; ip += 2) {
            const su32x4 vh = {xr[ip].x, xr[ip].y, xr[ip + 1].x, xr[ip + 1].y};
            const sf32x4 t0 = smfma(vh, selA, zero4), t1 = smfma(vh, selB, zero4);
            xT[ip] = su32x4{spack_exact(t0[0], t0[1]), spack_exact(t0[2], t0[3]), 0u, 0u};
            xT[ip + 1] = su32x4{spack_exact(t1[0], t1[1]), spack_exact(t1[2], t1[3]), 0u, 0u};
        }

        if constexpr (WRITE_Y) {
            // ---- (C B^T)[s][t], exact: A rows = B_s, B columns = C_t, K = the 128 state dimensions -----------------
            sf32x4 G = zero4;
#pragma unroll
            for (int pr = 0; pr < 8; pr += 2) {
                const su32x4 a = {Br[pr].x, Br[pr].y, Br[pr + 1].x, Br[pr + 1].y};
                const su32x4 bq = {Cr[pr].x, Cr[pr].y, Cr[pr + 1].x, Cr[pr + 1].y};
                G = smfma(a, bq, G);
            }
            // M[s][t] = G e^{cum_t - cum_s} dt_s for s <= t (this lane: t = t16, s = 4 q + g), split hi + lo
            float Mv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) Mv[g] = (4 * q + g <= t16) ? G[g] * __expf(cum - cs[g]) * ds[g] : 0.f;
            su32x4 Mh = {0u, 0u, 0u, 0u}, Ml = {0u, 0u, 0u, 0u};
            { const SHiLo u = ssplit_pk(Mv[0], Mv[1]); Mh[0] = u.hi; Ml[0] = u.lo; }
            { const SHiLo u = ssplit_pk(Mv[2], Mv[3]); Mh[1] = u.hi; Ml[1] = u.lo; }
            // ---- y_t = e^{cum_t} C_t . H (split state) + sum_s M[s][t] x_s ----------------------------------------
#pragma unroll
            for (int in = 0; in < 4; ++in) {
                sf32x4 Y = zero4;
#pragma unroll
                for (int pr = 0; pr < 8; pr += 2) {
                    const su32x4 a = {Cr[pr].x, Cr[pr].y, Cr[pr + 1].x, Cr[pr + 1].y};
                    su32x4 sh, sl;
                    { const SHiLo u = ssplit_pk(S[pr][in][0], S[pr][in][1]); sh[0] = u.hi; sl[0] = u.lo; }
                    { const SHiLo u = ssplit_pk(S[pr][in][2], S[pr][in][3]); sh[1] = u.hi; sl[1] = u.lo; }
                    { const SHiLo u = ssplit_pk(S[pr + 1][in][0], S[pr + 1][in][1]); sh[2] = u.hi; sl[2] = u.lo; }
                    { const SHiLo u = ssplit_pk(S[pr + 1][in][2], S[pr + 1][in][3]); sh[3] = u.hi; sl[3] = u.lo; }
                    Y = smfma(a, sh, Y);
                    Y = smfma(a, sl, Y);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) Y[g] *= __expf(cs[g]);       // rows of the C/D layout are t = 4 q + g
                Y = smfma(Mh, xT[in], Y);
                Y = smfma(Ml, xT[in], Y);
#pragma unroll
                for (int g = 0; g < 4; ++g) s_y[4 * q + g][16 * in + t16] = Y[g];
            }
        } else {
            lsum += c15;
        }

        // ---- state: H <- e^{cum_15} H + sum_s (e^{cum_15 - cum_s} dt_s B_s) x_s^T ---------------------------------
        const float csc = __expf(c15 - cum) * dt_own;            // this lane's row s = t16
#pragma unroll
        for (int pr = 0; pr < 8; pr += 2) {
            su32x4 kh, kl;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint2 r = Br[pr + e];
                const float b0 = bf16_bits_to_f32(r.x & 0xffffu) * csc, b1 = __uint_as_float(r.x & 0xffff0000u) * csc;
                const float b2 = bf16_bits_to_f32(r.y & 0xffffu) * csc, b3 = __uint_as_float(r.y & 0xffff0000u) * csc;
                const SHiLo u0 = ssplit_pk(b0, b1), u1 = ssplit_pk(b2, b3);
                kh[2 * e] = u0.hi; kh[2 * e + 1] = u1.hi; kl[2 * e] = u0.lo; kl[2 * e + 1] = u1.lo;
            }
            const sf32x4 h0 = smfma(kh, selA, zero4), h1 = smfma(kh, selB, zero4);
            const sf32x4 l0 = smfma(kl, selA, zero4), l1 = smfma(kl, selB, zero4);
            const su32x4 ah0 = {spack_exact(h0[0], h0[1]), spack_exact(h0[2], h0[3]), 0u, 0u};
            const su32x4 ah1 = {spack_exact(h1[0], h1[1]), spack_exact(h1[2], h1[3]), 0u, 0u};
            const su32x4 al0 = {spack_exact(l0[0], l0[1]), spack_exact(l0[2], l0[3]), 0u, 0u};
            const su32x4 al1 = {spack_exact(l1[0], l1[1]), spack_exact(l1[2], l1[3]), 0u, 0u};
#pragma unroll
            for (int in = 0; in < 4; ++in) {
                S[pr][in] *= e15;
                S[pr + 1][in] *= e15;
                S[pr][in] = smfma(ah0, xT[in], S[pr][in]);
                S[pr + 1][in] = smfma(ah1, xT[in], S[pr + 1][in]);
                S[pr][in] = smfma(al0, xT[in], S[pr][in]);
                S[pr + 1][in] = smfma(al1, xT[in], S[pr + 1][in]);
            }
        }

        if constexpr (WRITE_Y) {
            __syncthreads();
            const int nvalid = min(SBL, s_end - s0);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int tt = pass * 4 + (lane >> 4), col = (lane & 15) * 4;
                const int trow = p.reverse ? p.L - 1 - (s0 + tt) : s0 + tt;
                if (tt < nvalid) {
                    const float4 v = *reinterpret_cast<const float4 *>(&s_y[tt][col]);
                    if constexpr (Y16) {     // + D x, one rounding to bf16 (x staged in LDS by the lanes that loaded it)
                        const uint2 xq = *reinterpret_cast<const uint2 *>(&s_x[tt][col]);
                        const float o0 = fmaf(dk, bf16_bits_to_f32(xq.x & 0xffffu), v.x);
                        const float o1 = fmaf(dk, __uint_as_float(xq.x & 0xffff0000u), v.y);
                        const float o2 = fmaf(dk, bf16_bits_to_f32(xq.y & 0xffffu), v.z);
                        const float o3 = fmaf(dk, __uint_as_float(xq.y & 0xffff0000u), v.w);
                        *reinterpret_cast<uint2 *>(p.y16 + ((size_t)b * p.L + trow) * p.d_inner + h * SP + col) =
                            make_uint2(scvt_pk(o0, o1), scvt_pk(o2, o3));
                    } else {
                        *reinterpret_cast<float4 *>(p.y + ((size_t)b * p.L + trow) * p.d_inner + h * SP + col) = v;
                    }
                }
            }
        }
    }

    if constexpr (!WRITE_Y) {
        float *ws = p.ws_state + (seq * p.NC + c) * (size_t)(SN * SP);
#pragma unroll
        for (int jm = 0; jm < 8; ++jm)
#pragma unroll
            for (int in = 0; in < 4; ++in)
#pragma unroll
                for (int g = 0; g < 4; ++g) ws[(16 * jm + 4 * q + g) * SP + 16 * in + t16] = S[jm][in][g];
        if (lane == 0) p.ws_decay[seq * p.NC + c] = __expf(lsum);
    }
}

// pass B: exclusive scan of (decay, state) over the chunks of one (batch, head), in place
__global__ __launch_bounds__(256) void mamba2_ssd_scan_kernel(const SsdParams p) {
    const size_t seq = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;     // 0 .. 8191
    float *ws = p.ws_state + seq * p.NC * (size_t)(SN * SP) + e;
    const float *wd = p.ws_decay + seq * p.NC;
    float run = 0.f;
    for (int c = 0; c < p.nc_local; ++c) {
        const float loc = ws[(size_t)c * (SN * SP)];
        ws[(size_t)c * (SN * SP)] = run;
        run = fmaf(run, wd[c], loc);
    }
    if (p.nc_local < p.NC) ws[(size_t)(p.NC - 1) * (SN * SP)] = run;
}

}  // namespace
}  // namespace pafc

extern "C" int pafc_mamba2_scan_chunk_len(int B, int L, int H) {
    const long seqs = (long)B * H;
    if (seqs >= 2048 || L <= 64) return L;
    long nc = (2048 + seqs - 1) / seqs;
    long Lc = (L + nc - 1) / nc;
    if (Lc < 64) Lc = 64;
    Lc = (Lc + 15) / 16 * 16;
    return Lc >= L ? L : (int)Lc;
}

extern "C" size_t pafc_mamba2_scan_workspace_bytes(int B, int L, int H, int chunk_len) {
    if (B <= 0 || L <= 0 || H <= 0) return 0;
    const int Lc = chunk_len > 0 ? chunk_len : pafc_mamba2_scan_chunk_len(B, L, H);
    if (Lc >= L) return 0;
    const size_t NC = (L + Lc - 1) / Lc;
    return sizeof(float) * (size_t)B * H * NC * (pafc::SN * pafc::SP + 1);
}

extern "C" int pafc_mamba2_scan(int B, int L, int H, const void *xbc, long ldx, const float *dt, const float *log_a, float *y,
                                int chunk_len, void *workspace, size_t workspace_bytes, pafc_stream_t stream) {
    return pafc_mamba2_scan_dir(B, L, H, xbc, ldx, dt, log_a, y, 0, chunk_len, workspace, workspace_bytes, stream);
}

namespace pafc {
namespace {
int ssd_launch(int B, int L, int H, const void *xbc, long ldx, const float *dt, const float *log_a, float *y, bf16_t *y16,
               const float *Dskip, int reverse, int chunk_len, void *workspace, size_t workspace_bytes, hipStream_t s) {
    if (!xbc || !dt || !log_a || (!y && !y16) || (y16 && !Dskip)) return PAFC_ERR_NULL_POINTER;
    if (B <= 0 || L <= 0 || H <= 0 || (long)B * H > 65535 || ldx < (long)H * 64 + 256 || (ldx % 4)) return PAFC_ERR_BAD_DIMS;
    if (((uintptr_t)xbc & 7) || ((uintptr_t)y & 15) || ((uintptr_t)y16 & 7)) return PAFC_ERR_ALIGNMENT;
    int Lc = chunk_len > 0 ? chunk_len : pafc_mamba2_scan_chunk_len(B, L, H);
    if (!workspace) Lc = L;
    if (Lc < L) Lc = (Lc + 15) / 16 * 16;
    if (Lc >= L) Lc = L;
    SsdParams p{};
    p.xbc = (const bf16_t *)xbc; p.ldx = ldx; p.dt = dt; p.la = log_a; p.y = y; p.y16 = y16; p.Dskip = Dskip;
    p.B = B; p.L = L; p.H = H; p.d_inner = H * 64; p.Lc = Lc; p.reverse = reverse ? 1 : 0;
    p.NC = (L + Lc - 1) / Lc;
    p.nc_local = p.NC - 1;
    if (p.NC > 1) {
        if (workspace_bytes < pafc_mamba2_scan_workspace_bytes(B, L, H, Lc)) return PAFC_ERR_WORKSPACE;
        p.ws_state = (float *)workspace;
        p.ws_decay = p.ws_state + (size_t)B * H * p.NC * (SN * SP);
        hipLaunchKernelGGL((mamba2_ssd_kernel<false, false>), dim3(p.nc_local, B * H), dim3(64), 0, s, p);
        hipLaunchKernelGGL(mamba2_ssd_scan_kernel, dim3(SN * SP / 256, B * H), dim3(256), 0, s, p);
    }
    if (y16) hipLaunchKernelGGL((mamba2_ssd_kernel<true, true>), dim3(p.NC, B * H), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((mamba2_ssd_kernel<true, false>), dim3(p.NC, B * H), dim3(64), 0, s, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}
}  // namespace
}  // namespace pafc

extern "C" int pafc_mamba2_scan_dir(int B, int L, int H, const void *xbc, long ldx, const float *dt, const float *log_a,
                                    float *y, int reverse, int chunk_len, void *workspace, size_t workspace_bytes,
                                    pafc_stream_t stream) {
    return pafc::ssd_launch(B, L, H, xbc, ldx, dt, log_a, y, nullptr, nullptr, reverse, chunk_len, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

extern "C" int pafc_mamba2_scan_skip_bf16(int B, int L, int H, const void *xbc, long ldx, const float *dt, const float *log_a,
                                          const float *D, void *y_bf16, int reverse, int chunk_len, void *workspace,
                                          size_t workspace_bytes, pafc_stream_t stream) {
    return pafc::ssd_launch(B, L, H, xbc, ldx, dt, log_a, nullptr, (pafc::bf16_t *)y_bf16, D, reverse, chunk_len, workspace,
                            workspace_bytes, (hipStream_t)stream);
}
