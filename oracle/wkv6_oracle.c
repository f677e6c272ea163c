/*
 * wkv6_oracle.c -- CPU restatement of the reference's WKV-6 recurrence.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under paper_accurate_fast_cheap_amd/ may
 * include, link or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the reported
 * CPU baseline -- never as the thing shipped.
 *
 * What it restates (paths relative to /root/reference):
 *   forward            wenet/rwkv_v6/cuda/wkv6_cuda.cu:8-63    (kernel_forward)
 *   forward + state    wenet/rwkv_v6/cuda/wkv6state_cuda.cu:6-65 (initial state
 *                      layout s[b][h][i][j], i = value index, j = key index;
 *                      the reference never writes the final state back, this
 *                      restatement can, because the carry tests need it)
 *   backward gr, gu    wenet/rwkv_v6/cuda/wkv6_cuda.cu:65-109  (kernel_backward_101)
 *   backward gk        wenet/rwkv_v6/cuda/wkv6_cuda.cu:111-151 (kernel_backward_102)
 *   backward gv        wenet/rwkv_v6/cuda/wkv6_cuda.cu:153-195 (kernel_backward_103)
 *   backward gw        wenet/rwkv_v6/cuda/wkv6_cuda.cu:197-263 (kernel_backward_201)
 *
 * Differences from the CUDA source, all deliberate:
 *   - exact expf() instead of __expf() (the CUDA build uses --use_fast_math,
 *     wenet/rwkv_v6/src/model.py:106); no flush-to-zero;
 *   - a*b+c is written as fmaf(a,b,c) where nvcc's default -fmad=true would
 *     contract it, so the rounding points are those of the GPU build;
 *   - kernel_backward_201 keeps a per-thread array sbbbb[_T_-2] and therefore
 *     needs T <= _T_ (2048); here the scratch is heap-allocated, no limit;
 *   - the reference walks a flat pointer t = b*T*C + h*N + i (+C per step);
 *     here the indices are explicit.
 *
 * Pinning: the reference has NO executable implementation of this arithmetic
 * outside CUDA (model.py:296-299 always calls the CUDA op) and no tests, so
 * this file is pinned by (i) line-by-line restatement of the .cu source and
 * (ii) an independent float64 closed-form check in tests/test_oracle_goldens.py.
 * Everything *around* the op is pinned by goldens captured from the reference
 * Python (tests/golden/make_goldens.py), which runs with this file as the op.
 *
 * Element types: "f32" = float in/out; "bf16" = uint16_t holding the upper 16
 * bits of an IEEE float, converted like at::BFloat16 (round-to-nearest-even,
 * NaN kept quiet).  Accumulation is always float, as in the CUDA kernels
 * (float state[_N_] even in the fp64 instantiation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)0x7fc0; /* NaN, as c10::BFloat16 */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

/* One generic body per kernel, instantiated for the two element types through
 * LOAD/STORE macros (C has no templates). */

#define DEFINE_FORWARD(NAME, ET, LD, ST)                                                     \
void NAME(int B, int T, int C, int H, const ET *r, const ET *k, const ET *v, const ET *w,    \
          const ET *u, ET *y, const float *s_in, float *s_out, int reverse)                  \
{                                                                                            \
    const int N = C / H;                                                                     \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                 \
    for (int b = 0; b < B; ++b) {                                                            \
        for (int h = 0; h < H; ++h) {                                                        \
            /* state[i][j]: thread i of the CUDA block owns state[j], j = key index */       \
            float *state = (float *)calloc((size_t)N * N, sizeof(float));                    \
            float *rr = (float *)malloc(sizeof(float) * 4 * N);                              \
            float *kk = rr + N, *ww = kk + N, *uu = ww + N;                                  \
            if (s_in) memcpy(state, s_in + ((size_t)b * H + h) * N * N, sizeof(float) * N * N); \
            for (int j = 0; j < N; ++j) uu[j] = LD(u[h * N + j]);                            \
            for (int step = 0; step < T; ++step) {                                           \
                const int t = reverse ? (T - 1 - step) : step;                               \
                const size_t base = ((size_t)b * T + t) * C + (size_t)h * N;                 \
                for (int j = 0; j < N; ++j) {                                                \
                    ww[j] = expf(-expf(LD(w[base + j])));                                    \
                    rr[j] = LD(r[base + j]);                                                 \
                    kk[j] = LD(k[base + j]);                                                 \
                }                                                                            \
                for (int i = 0; i < N; ++i) {                                                \
                    const float vv = LD(v[base + i]);                                        \
                    float *s = state + (size_t)i * N;                                        \
                    float acc = 0.f;                                                         \
                    for (int j = 0; j < N; ++j) {                                            \
                        const float x = kk[j] * vv;                                          \
                        acc = fmaf(rr[j], fmaf(uu[j], x, s[j]), acc);                        \
                        s[j] = fmaf(s[j], ww[j], x);                                         \
                    }                                                                        \
                    y[base + i] = ST(acc);                                                   \
                }                                                                            \
            }                                                                                \
            if (s_out) memcpy(s_out + ((size_t)b * H + h) * N * N, state, sizeof(float) * N * N); \
            free(state);                                                                     \
            free(rr);                                                                        \
        }                                                                                    \
    }                                                                                        \
}

#define LD_F32(x) (x)
#define ST_F32(x) (x)
#define LD_BF16(x) bf16_to_f32(x)
#define ST_BF16(x) f32_to_bf16(x)

DEFINE_FORWARD(wkv6_oracle_forward_f32, float, LD_F32, ST_F32)
DEFINE_FORWARD(wkv6_oracle_forward_bf16, uint16_t, LD_BF16, ST_BF16)

/* Backward.  gu is the per-batch partial (B, C) exactly as the kernel writes it;
 * the sum over B happens on the host in the reference (model.py:151) and in
 * the caller here. */
#define DEFINE_BACKWARD(NAME, ET, LD, ST)                                                    \
void NAME(int B, int T, int C, int H, const ET *r, const ET *k, const ET *v, const ET *w,    \
          const ET *u, const ET *gy, ET *gr, ET *gk, ET *gv, ET *gw, ET *gu)                 \
{                                                                                            \
    const int N = C / H;                                                                     \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                 \
    for (int b = 0; b < B; ++b) {                                                            \
        for (int h = 0; h < H; ++h) {                                                        \
            float *st = (float *)malloc(sizeof(float) * ((size_t)N * N + 2 * N));            \
            float *va = st + (size_t)N * N, *ga = va + N;                                    \
            float *sb = (float *)calloc((size_t)(T > 2 ? T : 2) * N, sizeof(float));         \
            const size_t b0 = (size_t)b * T * C + (size_t)h * N;                             \
            /* ---- kernel_backward_101: gr, gu (forward-time sweep) ---- */                 \
            memset(st, 0, sizeof(float) * N * N);                                            \
            float *gu_acc = (float *)calloc(N, sizeof(float));                               \
            for (int t = 0; t < T; ++t) {                                                    \
                const size_t base = b0 + (size_t)t * C;                                      \
                for (int j = 0; j < N; ++j) { va[j] = LD(v[base + j]); ga[j] = LD(gy[base + j]); } \
                for (int i = 0; i < N; ++i) {                                                \
                    const float uu = LD(u[h * N + i]);                                       \
                    const float kk = LD(k[base + i]);                                        \
                    const float ww = expf(-expf(LD(w[base + i])));                           \
                    float *s = st + (size_t)i * N;                                           \
                    float g = 0.f, gu_ = 0.f;                                                \
                    for (int j = 0; j < N; ++j) {                                            \
                        const float x = kk * va[j];                                          \
                        g = fmaf(fmaf(uu, x, s[j]), ga[j], g);                               \
                        gu_ = fmaf(x, ga[j], gu_);                                           \
                        s[j] = fmaf(s[j], ww, x);                                            \
                    }                                                                        \
                    gr[base + i] = ST(g);                                                    \
                    gu_acc[i] = fmaf(LD(r[base + i]), gu_, gu_acc[i]);                       \
                }                                                                            \
            }                                                                                \
            for (int i = 0; i < N; ++i) gu[(size_t)b * C + h * N + i] = ST(gu_acc[i]);       \
            free(gu_acc);                                                                    \
            /* ---- kernel_backward_102: gk (reverse-time sweep) ---- */                     \
            memset(st, 0, sizeof(float) * N * N);                                            \
            for (int t = T - 1; t >= 0; --t) {                                               \
                const size_t base = b0 + (size_t)t * C;                                      \
                for (int j = 0; j < N; ++j) { va[j] = LD(v[base + j]); ga[j] = LD(gy[base + j]); } \
                for (int i = 0; i < N; ++i) {                                                \
                    const float uu = LD(u[h * N + i]);                                       \
                    const float rr = LD(r[base + i]);                                        \
                    const float ww = expf(-expf(LD(w[base + i])));                           \
                    float *s = st + (size_t)i * N;                                           \
                    float g = 0.f;                                                           \
                    for (int j = 0; j < N; ++j) {                                            \
                        const float x = rr * ga[j];                                          \
                        g = fmaf(fmaf(uu, x, s[j]), va[j], g);                               \
                        s[j] = fmaf(s[j], ww, x);                                            \
                    }                                                                        \
                    gk[base + i] = ST(g);                                                    \
                }                                                                            \
            }                                                                                \
            /* ---- kernel_backward_103: gv (reverse-time sweep) ---- */                     \
            memset(st, 0, sizeof(float) * N * N);                                            \
            {                                                                                \
                float *ra = (float *)malloc(sizeof(float) * 4 * N);                          \
                float *ka = ra + N, *wa = ka + N, *ua = wa + N;                              \
                for (int j = 0; j < N; ++j) ua[j] = LD(u[h * N + j]);                        \
                for (int t = T - 1; t >= 0; --t) {                                           \
                    const size_t base = b0 + (size_t)t * C;                                  \
                    for (int j = 0; j < N; ++j) {                                            \
                        ra[j] = LD(r[base + j]);                                             \
                        ka[j] = LD(k[base + j]);                                             \
                        wa[j] = expf(-expf(LD(w[base + j])));                                \
                    }                                                                        \
                    for (int i = 0; i < N; ++i) {                                            \
                        const float gyy = LD(gy[base + i]);                                  \
                        float *s = st + (size_t)i * N;                                       \
                        float g = 0.f;                                                       \
                        for (int j = 0; j < N; ++j) {                                        \
                            const float x = gyy * ra[j];                                     \
                            g = fmaf(fmaf(ua[j], x, s[j]), ka[j], g);                        \
                            s[j] = fmaf(s[j], wa[j], x);                                     \
                        }                                                                    \
                        gv[base + i] = ST(g);                                                \
                    }                                                                        \
                }                                                                            \
                free(ra);                                                                    \
            }                                                                                \
            /* ---- kernel_backward_201: gw (two sweeps) ---- */                             \
            if (T == 1) {                                                                    \
                for (int i = 0; i < N; ++i) gw[b0 + i] = ST(0.f);                            \
            } else {                                                                         \
                /* sweep 1: t = T-1 .. 2 (reverse), saaaa */                                 \
                memset(st, 0, sizeof(float) * N * N);                                        \
                for (int t = T - 1; t > 1; --t) {                                            \
                    const size_t base = b0 + (size_t)t * C;                                  \
                    for (int j = 0; j < N; ++j) {                                            \
                        ga[j] = LD(gy[base + j]);                                            \
                        va[j] = LD(v[base - 2 * (size_t)C + j]);                             \
                    }                                                                        \
                    for (int i = 0; i < N; ++i) {                                            \
                        const float rr = LD(r[base + i]);                                    \
                        const float ww = expf(-expf(LD(w[base - C + i])));                   \
                        float *s = st + (size_t)i * N;                                       \
                        float sum = 0.f;                                                     \
                        for (int j = 0; j < N; ++j) {                                        \
                            const float x = rr * ga[j];                                      \
                            s[j] = (s[j] + x) * ww;                                          \
                            sum = fmaf(s[j], va[j], sum);                                    \
                        }                                                                    \
                        sb[(size_t)(t - 2) * N + i] = sum * LD(k[base - 2 * (size_t)C + i]); \
                    }                                                                        \
                }                                                                            \
                float *sss = (float *)malloc(sizeof(float) * N);                             \
                for (int i = 0; i < N; ++i) {                                                \
                    sss[i] = sb[i]; /* sbbbb[0]; zero when T == 2 (array is zero-init) */    \
                    gw[b0 + i] = ST(0.f);                                                    \
                    gw[b0 + C + i] = ST(sss[i] * -expf(LD(w[b0 + C + i])));                  \
                }                                                                            \
                /* sweep 2: t = 2 .. T-2 (forward), scccc */                                 \
                memset(st, 0, sizeof(float) * N * N);                                        \
                for (int t = 2; t < T - 1; ++t) {                                            \
                    const size_t base = b0 + (size_t)t * C;                                  \
                    for (int j = 0; j < N; ++j) {                                            \
                        ga[j] = LD(gy[base + j]);                                            \
                        va[j] = LD(v[base - 2 * (size_t)C + j]);                             \
                    }                                                                        \
                    for (int i = 0; i < N; ++i) {                                            \
                        const float ww = expf(-expf(LD(w[base - C + i])));                   \
                        const float kk = LD(k[base - 2 * (size_t)C + i]);                    \
                        float *s = st + (size_t)i * N;                                       \
                        float sum = 0.f;                                                     \
                        for (int j = 0; j < N; ++j) {                                        \
                            const float x = kk * va[j];                                      \
                            s[j] = (s[j] + x) * ww;                                          \
                            sum = fmaf(s[j], ga[j], sum);                                    \
                        }                                                                    \
                        sss[i] += fmaf(-sum, LD(r[base + i]), sb[(size_t)(t - 1) * N + i]);  \
                        gw[base + i] = ST(sss[i] * -expf(LD(w[base + i])));                  \
                    }                                                                        \
                }                                                                            \
                /* the kernel stores _gw[t_1] first and _gw[t_T_1] = 0 last: for T == 2   */ \
                /* both name the same element, so the final value is 0                    */ \
                for (int i = 0; i < N; ++i) gw[b0 + (size_t)(T - 1) * C + i] = ST(0.f);      \
                free(sss);                                                                   \
            }                                                                                \
            free(sb);                                                                        \
            free(st);                                                                        \
        }                                                                                    \
    }                                                                                        \
}

DEFINE_BACKWARD(wkv6_oracle_backward_f32, float, LD_F32, ST_F32)
DEFINE_BACKWARD(wkv6_oracle_backward_bf16, uint16_t, LD_BF16, ST_BF16)

/* Independent float64 closed form of the forward, used only to pin the float
 * restatement above (tests/test_oracle_goldens.py):
 *   y_t[i] = sum_j r_t[j] * ( u[j] k_t[j] v_t[i]
 *                             + sum_{s<t} (prod_{s<q<t} d_q[j]) k_s[j] v_s[i] )
 * evaluated directly (O(T^2)), no running state. */
void wkv6_oracle_forward_closed_form_f64(int B, int T, int C, int H, const double *r,
                                         const double *k, const double *v, const double *w,
                                         const double *u, double *y)
{
    const int N = C / H;
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int t = 0; t < T; ++t) {
                const size_t bt = ((size_t)b * T + t) * C + (size_t)h * N;
                for (int i = 0; i < N; ++i) {
                    double acc = 0.0;
                    for (int j = 0; j < N; ++j) {
                        double inner = u[h * N + j] * k[bt + j] * v[bt + i];
                        double decay = 1.0; /* prod over s<q<t, built walking s downward */
                        for (int s = t - 1; s >= 0; --s) {
                            const size_t bs = ((size_t)b * T + s) * C + (size_t)h * N;
                            inner += decay * k[bs + j] * v[bs + i];
                            decay *= exp(-exp(w[bs + j]));
                        }
                        acc += r[bt + j] * inner;
                    }
                    y[bt + i] = acc;
                }
            }
}
