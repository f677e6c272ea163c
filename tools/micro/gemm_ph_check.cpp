// Stand-alone check + race of csrc/gemm_ph.hip (no Python, no torch: seconds per run on the GPU box):
//   * every epilogue / output form against a naive fp32 GEMM on the same operands, element-wise, at the 30-minute shapes
//     (44 998 rows, ragged last tile, N = 5000 column tail, 192-row tiles, batched problems) and at small odd shapes;
//   * the round-2 kernel (tools/micro/gemm_ph_r02.inc, a verbatim copy of the file this round replaced) raced against
//     the new one in ONE process, variants interleaved round by round, medians reported.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I paper_accurate_fast_cheap_amd/csrc tools/micro/gemm_ph_check.cpp -o tools/micro/bin/gemm_ph_check
//   tools/micro/bin/gemm_ph_check [check|race|all]
#include "pafc_common.h"
#include "../../include/pafc_encoder_ops.h"

namespace pafc_r02 { using namespace pafc; }
#define pafc pafc_r02
#define pafc_gemm_bf16_ph r02_gemm_bf16_ph
#define pafc_conv3x3s2_nhwc_bf16_ph r02_conv3x3s2_nhwc_bf16_ph
#include "gemm_ph_r02.inc"
#undef pafc
#undef pafc_gemm_bf16_ph
#undef pafc_conv3x3s2_nhwc_bf16_ph
#undef PH_STAMP
#undef PH_WAIT
#undef PH_BAR1
#undef PH_BAR2
#include "../../paper_accurate_fast_cheap_amd/csrc/gemm_ph.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned short u16;

__device__ __host__ inline float bf2f(u16 h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
__device__ inline u16 f2bf(float f) { return (u16)pafc::f32_to_bf16_bits(f); }

__global__ void fill_bf16(u16 *p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; h *= 0x27d4eb2du; h ^= h >> 16;
        p[i] = f2bf(((int)(h & 0xffff) - 32768) / 32768.0f * scale);
    }
}
__global__ void fill_f32(float *p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; h *= 0x27d4eb2du; h ^= h >> 16;
        p[i] = ((int)(h & 0xffffff) - 8388608) / 8388608.0f * scale;
    }
}
// x (rows x cols fp32) -> planes [hi | lo] (rows x 2 cols bf16)
__global__ void split_planes(const float *x, u16 *out, long rows, int cols) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t n = (size_t)rows * cols;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long r = i / cols; const int c = (int)(i % cols);
        const u16 hi = f2bf(x[i]);
        out[r * 2 * cols + c] = hi;
        out[r * 2 * cols + cols + c] = f2bf(x[i] - bf2f(hi));
    }
}
// w (N x K fp32) -> [hi | hi | lo] (N x 3K bf16)
__global__ void split_weight3(const float *w, u16 *out, int N, int K) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t n = (size_t)N * K;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long r = i / K; const int c = (int)(i % K);
        const u16 hi = f2bf(w[i]);
        out[r * 3 * K + c] = hi;
        out[r * 3 * K + K + c] = hi;
        out[r * 3 * K + 2 * K + c] = f2bf(w[i] - bf2f(hi));
    }
}

// naive reference: lin[z][m][n] = sum_k A[z][m][k] W[z][n][k] in fp32 (operands bf16 or fp32)
template <typename TA>
__global__ void ref_gemm(const TA *A, const TA *W, float *lin, long M, int N, int K) {
    const long m = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = blockIdx.z;
    if (n >= N) return;
    const TA *a = A + ((size_t)z * M + m) * K, *w = W + ((size_t)z * N + n) * K;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k = 0; k < K; k += 4) {
        if constexpr (sizeof(TA) == 2) {
            s0 = fmaf(bf2f(a[k]), bf2f(w[k]), s0); s1 = fmaf(bf2f(a[k + 1]), bf2f(w[k + 1]), s1);
            s2 = fmaf(bf2f(a[k + 2]), bf2f(w[k + 2]), s2); s3 = fmaf(bf2f(a[k + 3]), bf2f(w[k + 3]), s3);
        } else {
            s0 = fmaf(a[k], w[k], s0); s1 = fmaf(a[k + 1], w[k + 1], s1);
            s2 = fmaf(a[k + 2], w[k + 2], s2); s3 = fmaf(a[k + 3], w[k + 3], s3);
        }
    }
    lin[((size_t)z * M + m) * N + n] = (s0 + s1) + (s2 + s3);
}

struct Cmp { double max_err = 0, max_over_tol = 0; size_t bad = 0, n = 0; };
// want[m][c] from lin (+bias, act, residual) vs got; act: 0 none 1 silu 2 tanh 3 relu 4 glu(h = 32)
// out_kind 0 bf16, 1 fp32, 2 planes; res_kind 0 none 1 bf16 2 fp32; bias fp32 when out_kind != 0
__global__ void compare(const float *lin, const void *bias, const void *res, const void *got, long M, int N, int act, int out_kind,
                        int res_kind, float alpha, long ldo, long lo_off, float rtol, float atol, double *stats /* max_err, max_over_tol, bad */) {
    const long m = blockIdx.y;
    const int No = act == 4 ? N / 2 : N;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = blockIdx.z;
    if (c >= No) return;
    auto B = [&](int n) -> float {
        if (!bias) return 0.f;
        return out_kind == 0 ? bf2f(((const u16 *)bias)[(size_t)z * N + n]) : ((const float *)bias)[(size_t)z * N + n];
    };
    const float *l = lin + ((size_t)z * M + m) * N;
    float want;
    if (act == 4) {
        const int t = c / 32, cc = c % 32;
        const float a = alpha * l[64 * t + cc] + B(64 * t + cc), b = alpha * l[64 * t + 32 + cc] + B(64 * t + 32 + cc);
        want = a / (1.f + expf(-b));
    } else {
        float v = alpha * l[c] + B(c);
        if (res_kind == 1) v += bf2f(((const u16 *)res)[((size_t)z * M + m) * N + c]);
        if (res_kind == 2) v += ((const float *)res)[((size_t)z * M + m) * N + c];
        want = act == 1 ? v / (1.f + expf(-v)) : act == 2 ? tanhf(v) : act == 3 ? fmaxf(v, 0.f) : v;
    }
    float g;
    const size_t o = ((size_t)z * M + m) * ldo + c;
    if (out_kind == 0) g = bf2f(((const u16 *)got)[o]);
    else if (out_kind == 1) g = ((const float *)got)[o];
    else g = bf2f(((const u16 *)got)[o]) + bf2f(((const u16 *)got)[o + lo_off]);
    const double err = fabs((double)g - want), tol = (double)rtol * fabs(want) + atol;
    if (!(err <= tol)) atomicAdd((unsigned long long *)&stats[2], 1ull);
    // (races on the two maxima lose nothing that matters: they are monotone under the CAS loops below)
    unsigned long long *pm = (unsigned long long *)&stats[0];
    unsigned long long old = *pm, assumed;
    do { assumed = old; if (__longlong_as_double(assumed) >= err) break; old = atomicCAS(pm, assumed, __double_as_longlong(err)); } while (assumed != old);
    const double r = err / tol;
    pm = (unsigned long long *)&stats[1];
    old = *pm;
    do { assumed = old; if (__longlong_as_double(assumed) >= r) break; old = atomicCAS(pm, assumed, __double_as_longlong(r)); } while (assumed != old);
}

static int g_fail = 0;

struct Case {
    const char *name;
    long M; int N, K, Z, act; int res_kind, out_kind, a_split; int tile_m; float alpha;
};

static float time_us(std::function<void()> fn, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

static void run_case(const Case &c, bool race) {
    const long M = c.M; const int N = c.N, K = c.K, Z = c.Z;
    const bool glu = c.act == 4;
    const int No = glu ? N / 2 : N;
    const int Ka = c.a_split ? 2 * K : K, Kw = c.a_split ? 3 * K : K;
    u16 *A, *W; float *Af = nullptr, *Wf = nullptr, *lin;
    void *bias, *res = nullptr, *out, *out_old = nullptr;
    CK(hipMalloc(&A, (size_t)Z * M * Ka * 2)); CK(hipMalloc(&W, (size_t)Z * N * Kw * 2));
    CK(hipMalloc(&lin, (size_t)Z * M * N * 4));
    const float wscale = 1.7f / sqrtf((float)K);
    if (c.a_split) {
        CK(hipMalloc(&Af, (size_t)Z * M * K * 4)); CK(hipMalloc(&Wf, (size_t)Z * N * K * 4));
        fill_f32<<<2048, 256>>>(Af, (size_t)Z * M * K, 11, 1.7f);
        fill_f32<<<512, 256>>>(Wf, (size_t)Z * N * K, 12, wscale);
        split_planes<<<2048, 256>>>(Af, A, Z * M, K);
        split_weight3<<<512, 256>>>(Wf, W, Z * N, K);
        ref_gemm<float><<<dim3((N + 255) / 256, (unsigned)M, Z), 256>>>(Af, Wf, lin, M, N, K);
    } else {
        fill_bf16<<<2048, 256>>>(A, (size_t)Z * M * K, 11, 1.7f);
        fill_bf16<<<512, 256>>>(W, (size_t)Z * N * K, 12, wscale);
        ref_gemm<u16><<<dim3((N + 255) / 256, (unsigned)M, Z), 256>>>(A, W, lin, M, N, K);
    }
    const bool bias_f32 = c.out_kind != 0;
    const bool nobias = strstr(c.name, "no bias") != nullptr;
    CK(hipMalloc(&bias, (size_t)Z * N * 4));
    if (bias_f32) fill_f32<<<8, 256>>>((float *)bias, (size_t)Z * N, 13, 0.5f);
    else fill_bf16<<<8, 256>>>((u16 *)bias, (size_t)Z * N, 13, 0.5f);
    if (c.res_kind) {
        CK(hipMalloc(&res, (size_t)Z * M * N * 4));
        if (c.res_kind == 2) fill_f32<<<2048, 256>>>((float *)res, (size_t)Z * M * N, 14, 1.3f);
        else fill_bf16<<<2048, 256>>>((u16 *)res, (size_t)Z * M * N, 14, 1.3f);
    }
    const long ldo = c.out_kind == 2 ? 2 * No : No, lo_off = c.out_kind == 2 ? No : 0;
    const size_t osz = c.out_kind == 1 ? 4 : 2;
    const size_t obytes = (size_t)Z * M * ldo * osz, guard = 4096;
    CK(hipMalloc(&out, obytes + guard));
    CK(hipMemset(out, 0x5a, obytes + guard));
    auto run_new = [&]() {
        const int rc = pafc_gemm_ph_ex(M, N, K, Z, A, Ka, M * Ka, c.a_split, W, Kw, (long)N * Kw, nobias ? nullptr : bias, N, res, c.res_kind, N, M * (long)N,
                                       out, c.out_kind, ldo, lo_off, M * ldo, c.alpha, c.act, c.tile_m, 0);
        if (rc != PAFC_OK) { printf("  %s: pafc_gemm_ph_ex returned %d\n", c.name, rc); g_fail++; }
    };
    run_new();
    CK(hipDeviceSynchronize());
    double *stats;
    CK(hipMalloc(&stats, 24));
    const float rtol = c.a_split ? 2e-4f : c.out_kind == 0 ? 1.f / 128 : 2e-3f;       // split: 3 bf16 products ~ 2^-16; bf16 out: one rounding
    const float atol = c.a_split ? 2e-4f : c.out_kind == 0 ? 2e-2f : 2e-3f;
    auto check = [&](const void *got, const char *who) {
        CK(hipMemset(stats, 0, 24));
        compare<<<dim3((No + 255) / 256, (unsigned)M, Z), 256>>>(lin, nobias ? nullptr : bias, res, got, M, N, c.act, c.out_kind, c.res_kind, c.alpha, ldo, lo_off,
                                                                  rtol, atol, stats);
        double h[3];
        CK(hipMemcpy(h, stats, 24, hipMemcpyDeviceToHost));
        unsigned long long bad; memcpy(&bad, &h[2], 8);
        printf("  %-44s %-4s max|err| %.3e  worst err/tol %.3f  elements out of tolerance %llu of %zu\n", c.name, who, h[0], h[1], bad,
               (size_t)Z * M * No);
        if (bad) g_fail++;
    };
    check(out, "new");
    // nothing written beyond the output
    std::vector<unsigned char> g(guard);
    CK(hipMemcpy(g.data(), (unsigned char *)out + obytes, guard, hipMemcpyDeviceToHost));
    for (unsigned char b : g) if (b != 0x5a) { printf("  %s: wrote beyond the output\n", c.name); g_fail++; break; }
    const bool has_old = !c.a_split && c.out_kind == 0 && c.res_kind != 2;
    if (has_old) {
        CK(hipMalloc(&out_old, obytes));
        auto run_old = [&]() {
            r02_gemm_bf16_ph(M, N, K, Z, A, K, M * K, W, K, (long)N * K, nobias ? nullptr : bias, N, res, N, M * (long)N, out_old, No, M * (long)No, c.alpha,
                             c.act, 256, c.tile_m, 0);
        };
        run_old();
        CK(hipDeviceSynchronize());
        if (!glu) check(out_old, "r02");       // (the r02 kernel's GLU wants the same h = 32 interleave: also comparable)
        else check(out_old, "r02");
        if (race) {
            std::vector<float> tn, to;
            for (int r = 0; r < 9; ++r) {
                to.push_back(time_us(run_old, 6));
                tn.push_back(time_us(run_new, 6));
            }
            std::sort(tn.begin(), tn.end()); std::sort(to.begin(), to.end());
            const double fl = 2.0 * Z * M * N * K;
            printf("  %-44s race: r02 %.1f us (min %.1f) = %.0f TF/s | new %.1f us (min %.1f) = %.0f TF/s | %.3fx\n", c.name, to[4], to[0],
                   fl / to[4] / 1e6, tn[4], tn[0], fl / tn[4] / 1e6, to[4] / tn[4]);
        }
    } else if (c.a_split && c.out_kind != 0) {
        // round 5: the shared-fragment walk (default, checked above) against the round-3 walk hi, lo, hi: same operands, same
        // process, interleaved rounds
        pafc::g_split_walk = 0;
        CK(hipMemset(out, 0x5a, obytes + guard));
        run_new();
        CK(hipDeviceSynchronize());
        check(out, "r03w");
        if (race) {
            std::vector<float> tn, to;
            for (int r = 0; r < 9; ++r) {
                pafc::g_split_walk = 0;
                to.push_back(time_us(run_new, 6));
                pafc::g_split_walk = 1;
                tn.push_back(time_us(run_new, 6));
            }
            std::sort(tn.begin(), tn.end()); std::sort(to.begin(), to.end());
            const double fl = 2.0 * Z * M * N * Kw;
            printf("  %-44s race: hi,lo,hi walk %.1f us (min %.1f) = %.0f TF/s | shared fragments %.1f us (min %.1f) = %.0f TF/s of bf16 products | %.3fx\n",
                   c.name, to[4], to[0], fl / to[4] / 1e6, tn[4], tn[0], fl / tn[4] / 1e6, to[4] / tn[4]);
        }
        pafc::g_split_walk = 1;
    } else if (race) {
        std::vector<float> tn;
        for (int r = 0; r < 9; ++r) tn.push_back(time_us(run_new, 6));
        std::sort(tn.begin(), tn.end());
        const double fl = 2.0 * Z * M * N * Kw;
        printf("  %-44s time: new %.1f us (min %.1f) = %.0f TF/s of bf16 products\n", c.name, tn[4], tn[0], fl / tn[4] / 1e6);
    }
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(lin)); CK(hipFree(bias)); CK(hipFree(out)); CK(hipFree(stats));
    if (res) CK(hipFree(res));
    if (out_old) CK(hipFree(out_old));
    if (Af) { CK(hipFree(Af)); CK(hipFree(Wf)); }
}

// `pmc <case>`: the new kernel alone, eight launches of one 30-minute shape on random operands -- the command profiled by
// rocprofv3 --pmc for profiles/r03*_gemm_ph_pmc_*.txt (0 SiLU 512->2048, 1 w_2 + residual, 2 GLU, 3 pointwise_conv2 +
// residual, 5 r,k,v stack, 6 CTC head)
static int run_pmc(const Case &c) {
    const long M = c.M; const int N = c.N, K = c.K, Z = c.Z;
    const int No = c.act == 4 ? N / 2 : N;
    u16 *A, *W, *bias, *res = nullptr, *out;
    CK(hipMalloc(&A, (size_t)Z * M * K * 2)); CK(hipMalloc(&W, (size_t)Z * N * K * 2)); CK(hipMalloc(&bias, (size_t)Z * N * 2));
    CK(hipMalloc(&out, (size_t)Z * M * No * 2));
    fill_bf16<<<2048, 256>>>(A, (size_t)Z * M * K, 11, 1.7f);
    fill_bf16<<<512, 256>>>(W, (size_t)Z * N * K, 12, 1.7f / sqrtf((float)K));
    fill_bf16<<<8, 256>>>(bias, (size_t)Z * N, 13, 0.5f);
    if (c.res_kind) { CK(hipMalloc(&res, (size_t)Z * M * N * 2)); fill_bf16<<<2048, 256>>>(res, (size_t)Z * M * N, 14, 1.3f); }
    for (int i = 0; i < 8; ++i) {
        const int rc = pafc_gemm_ph_ex(M, N, K, Z, A, K, M * K, 0, W, K, (long)N * K, bias, N, res, c.res_kind, N, M * (long)N, out, 0, No, 0,
                                       M * (long)No, c.alpha, c.act, c.tile_m, 0);
        if (rc != PAFC_OK) { printf("pafc_gemm_ph_ex returned %d\n", rc); return 1; }
    }
    CK(hipDeviceSynchronize());
    printf("%s: 8 launches done\n", c.name);
    return 0;
}

// Experiment (round 5): both operands with their planes INTERLEAVED in blocks of 32 columns, [hi 32 | lo 32] [hi 32 | lo 32] ...
// (a K-step's 128-byte LDS row is then ONE contiguous 128-byte line of the source row instead of two 64-byte halves 2 K bytes
// apart), against the shipped plane layout on the same fp32 operands.  `il <M> <N> <K>`.
__global__ void interleave32(const float *x, u16 *out, long rows, int cols) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t n = (size_t)rows * cols;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long r = i / cols; const int c = (int)(i % cols);
        const u16 hi = f2bf(x[i]);
        const size_t o = (size_t)r * 2 * cols + (size_t)(c >> 5) * 64 + (c & 31);
        out[o] = hi;
        out[o + 32] = f2bf(x[i] - bf2f(hi));
    }
}

static int run_interleaved(long M, int N, int K) {
    float *Af, *Wf, *bias, *out0, *out1;
    u16 *A0, *W0, *A1, *W1;
    CK(hipMalloc(&Af, (size_t)M * K * 4)); CK(hipMalloc(&Wf, (size_t)N * K * 4)); CK(hipMalloc(&bias, (size_t)N * 4));
    CK(hipMalloc(&A0, (size_t)M * 2 * K * 2)); CK(hipMalloc(&W0, (size_t)N * 3 * K * 2));
    CK(hipMalloc(&A1, (size_t)M * 2 * K * 2)); CK(hipMalloc(&W1, (size_t)N * 2 * K * 2));
    CK(hipMalloc(&out0, (size_t)M * N * 4)); CK(hipMalloc(&out1, (size_t)M * N * 4));
    fill_f32<<<2048, 256>>>(Af, (size_t)M * K, 11, 1.7f);
    fill_f32<<<512, 256>>>(Wf, (size_t)N * K, 12, 1.7f / sqrtf((float)K));
    fill_f32<<<8, 256>>>(bias, (size_t)N, 13, 0.5f);
    split_planes<<<2048, 256>>>(Af, A0, M, K);
    split_weight3<<<512, 256>>>(Wf, W0, N, K);
    interleave32<<<2048, 256>>>(Af, A1, M, K);
    interleave32<<<512, 256>>>(Wf, W1, N, K);
    const int tm = N <= 512 ? 192 : 256;
    auto run_planes = [&]() {
        const int rc = pafc_gemm_ph_ex(M, N, K, 1, A0, 2 * K, 0, 1, W0, 3 * K, 0, bias, 0, nullptr, 0, 0, 0, out0, 1, N, 0, 0, 1.f, 0, tm, 0);
        if (rc != PAFC_OK) { printf("planes: rc %d\n", rc); exit(1); }
    };
    auto run_il = [&]() {
        pafc::PhParams p{};
        p.A = (const pafc::bf16_t *)A1; p.W = (const pafc::bf16_t *)W1; p.bias = bias; p.out = out1;
        p.M = M; p.N = N; p.K = 2 * K; p.lda = 2 * K; p.ldw = 2 * K; p.ldo = N; p.alpha = 1.f;
        p.nk1 = K / 64; p.pb_shift = 0; p.pb_bytes = 64; p.nsteps = K / 32; p.a_lo = 64; p.w_lo = 64; p.w_step = 128;
        p.tm = tm; p.mtiles = (int)((M + tm - 1) / tm); p.ntiles = (N + 255) / 256;
        const int rc = pafc::launch_ph<false, 0, 0, 1, false, 0, true>(p, 1, 0);
        if (rc != PAFC_OK) { printf("interleaved: rc %d\n", rc); exit(1); }
    };
    run_planes(); run_il();
    CK(hipDeviceSynchronize());
    std::vector<float> h0((size_t)1 << 20), h1((size_t)1 << 20);
    const size_t n = std::min((size_t)M * N, h0.size());
    CK(hipMemcpy(h0.data(), out0, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), out1, n * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)fabsf(h0[i] - h1[i])); mx = std::max(mx, (double)fabsf(h0[i])); }
    std::vector<float> t0, t1;
    for (int r = 0; r < 9; ++r) { t0.push_back(time_us(run_planes, 6)); t1.push_back(time_us(run_il, 6)); }
    std::sort(t0.begin(), t0.end()); std::sort(t1.begin(), t1.end());
    printf("M %ld N %d K %d: planes [hi K | lo K] x [hi | hi | lo] %.1f us (min %.1f) | interleaved blocks of 32 %.1f us (min %.1f) | %.3fx | max |diff| %.2e of %.2e\n",
           M, N, K, t0[4], t0[0], t1[4], t1[0], t0[4] / t1[4], md, mx);
    return 0;
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "all";
    const bool race = mode != "check";
    const long M = 44998;
    const Case big[] = {
        {"ffn w_1 + SiLU 512->2048", M, 2048, 512, 1, 1, 0, 0, 0, 256, 1.f},
        {"ffn w_2 + residual 2048->512 (tm 192)", M, 512, 2048, 1, 0, 1, 0, 0, 192, 0.5f},
        {"pointwise_conv1 + GLU 512->1024", M, 1024, 512, 1, 4, 0, 0, 0, 256, 1.f},
        {"pointwise_conv2 + residual 512->512 (tm 192)", M, 512, 512, 1, 0, 1, 0, 0, 192, 1.f},
        {"slot output + residual 1024->512 (tm 192)", M, 512, 1024, 1, 0, 1, 0, 0, 192, 1.f},
        {"r,k,v stack 6 x 512->512", M, 512, 512, 6, 0, 0, 0, 0, 256, 1.f},
        {"CTC head 512->5000", M, 5000, 512, 1, 0, 0, 0, 0, 256, 1.f},
        {"plain 512->2048 tanh", M, 2048, 512, 1, 2, 0, 0, 0, 256, 1.f},
        // fp32 models: split operands, fp32 / plane outputs, fp32 residual
        {"f32: w_1 + SiLU -> planes (split A)", M, 2048, 512, 1, 1, 0, 2, 1, 256, 1.f},
        {"f32: w_2 + f32 residual (split A)", M, 512, 2048, 1, 0, 2, 1, 1, 192, 0.5f},
        {"f32: pointwise_conv1 + GLU -> f32 (split A)", M, 1024, 512, 1, 4, 0, 1, 1, 256, 1.f},
        {"f32: slot output bf16 A + f32 residual", M, 512, 1024, 1, 0, 2, 1, 0, 192, 1.f},
        {"f32: CTC head -> f32 (split A)", M, 5000, 512, 1, 0, 0, 1, 1, 256, 1.f},
        {"f32: pointwise_conv2 + f32 residual (split A)", M, 512, 512, 1, 0, 2, 1, 1, 192, 1.f},
        {"f32: Linear 9728->512 -> f32 (split A)", M, 512, 9728, 1, 0, 0, 1, 1, 192, 1.f},
    };
    const Case small[] = {
        {"small 256x256x128", 256, 256, 128, 1, 0, 0, 0, 0, 256, 1.f},
        {"small 1000x512x512 silu", 1000, 512, 512, 1, 1, 0, 0, 0, 256, 1.f},
        {"small 513x264x384 tanh", 513, 264, 384, 1, 2, 0, 0, 0, 256, 1.f},
        {"small 300x1024x256 relu x2", 300, 1024, 256, 2, 3, 0, 0, 0, 256, 1.f},
        {"small 2049x512x1024 res tm128", 2049, 512, 1024, 1, 0, 1, 0, 0, 128, 0.5f},
        {"small 777x2048x512 silu tm64", 777, 2048, 512, 1, 1, 0, 0, 0, 64, 1.f},
        {"small 260x512x2048 res x3", 260, 512, 2048, 3, 0, 1, 0, 0, 256, 1.f},
        {"small 1x256x128", 1, 256, 128, 1, 0, 0, 0, 0, 256, 1.f},
        {"small 700x1024x512 glu", 700, 1024, 512, 1, 4, 0, 0, 0, 256, 1.f},
        {"small f32 517x520x256 split res", 517, 520, 256, 1, 0, 2, 1, 1, 256, 1.f},
        {"small 300x512x128 no bias check", 300, 512, 128, 1, 0, 1, 0, 0, 256, 1.f},
        {"small f32 300x512x128 planes", 300, 512, 128, 2, 1, 0, 2, 1, 192, 1.f},
        {"small f32 1x256x128 split", 1, 256, 128, 1, 0, 0, 1, 1, 256, 1.f},
        {"small f32 777x1024x384 glu split tm64", 777, 1024, 384, 1, 4, 0, 1, 1, 64, 1.f},
        {"small f32 333x264x640 split res x2", 333, 264, 640, 2, 0, 2, 1, 1, 128, 0.5f},
    };
    if (mode == "pmc") return run_pmc(big[argc > 2 ? atoi(argv[2]) : 0]);
    if (mode == "il") return run_interleaved(argc > 2 ? atol(argv[2]) : M, argc > 3 ? atoi(argv[3]) : 2048, argc > 4 ? atoi(argv[4]) : 512);
    if (mode != "race") for (const Case &c : small) run_case(c, false);
    for (const Case &c : big) run_case(c, race);
    printf(g_fail ? "FAILED: %d problem(s)\n" : "all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
