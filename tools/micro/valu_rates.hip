// Micro-benchmark: issue rate of the VALU instructions the scan's hi + lo operand split can be built from, and an exactness
// check of the bf16 dot product as the "lo = a - hi" step (v_dot2c_f32_bf16 with a (-1, 0) / (0, -1) selector).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o tools/micro/bin/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
// one wave per SIMD (256 threads per block, one block per CU): cycles per instruction = elapsed / (iters * 64)
template <int WHICH>
__global__ void rate_kernel(float *out, int iters, float a, float b, unsigned long long mask) {
    float v0 = a + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    float w0 = b, w1 = b + 1, w2 = b + 2, w3 = b + 3, w4 = b + 4, w5 = b + 5, w6 = b + 6, w7 = b + 7;
    for (int it = 0; it < iters; ++it) {
        if constexpr (WHICH == 0) {          // v_sub_f32
            REP8(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                              "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0));)
        } else if constexpr (WHICH == 1) {   // v_dot2c_f32_bf16
            REP8(asm volatile("v_dot2c_f32_bf16 %0, %8, %9\n v_dot2c_f32_bf16 %1, %8, %9\n v_dot2c_f32_bf16 %2, %8, %9\n v_dot2c_f32_bf16 %3, %8, %9\n"
                              "v_dot2c_f32_bf16 %4, %8, %9\n v_dot2c_f32_bf16 %5, %8, %9\n v_dot2c_f32_bf16 %6, %8, %9\n v_dot2c_f32_bf16 %7, %8, %9\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0), "v"(w1));)
        } else if constexpr (WHICH == 2) {   // v_pk_add_f32 (two subtractions per instruction)
            REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              : "+v"(*(f32x2 *)&v0), "+v"(*(f32x2 *)&v2), "+v"(*(f32x2 *)&v4), "+v"(*(f32x2 *)&v6) : "v"(*(f32x2 *)&w0));)
        } else if constexpr (WHICH == 3) {   // v_cvt_pk_bf16_f32
            REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %8, %9\n v_cvt_pk_bf16_f32 %1, %8, %9\n v_cvt_pk_bf16_f32 %2, %8, %9\n v_cvt_pk_bf16_f32 %3, %8, %9\n"
                              "v_cvt_pk_bf16_f32 %4, %8, %9\n v_cvt_pk_bf16_f32 %5, %8, %9\n v_cvt_pk_bf16_f32 %6, %8, %9\n v_cvt_pk_bf16_f32 %7, %8, %9\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0), "v"(w1));)
        } else if constexpr (WHICH == 4) {   // v_cndmask_b32 (VOP3 form with an SGPR-pair mask)
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0) : "vcc");)
        } else if constexpr (WHICH == 5) {   // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if constexpr (WHICH == 6) {   // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              : "+v"(*(f32x2 *)&v0), "+v"(*(f32x2 *)&v2), "+v"(*(f32x2 *)&v4), "+v"(*(f32x2 *)&v6) : "v"(*(f32x2 *)&w0));)
        } else if constexpr (WHICH == 7) {   // v_perm_b32
            REP8(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                              "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0), "v"(w1));)
        } else if constexpr (WHICH == 9) {   // v_cndmask_b32, VOP3 form, mask in an SGPR pair, distinct destination chains
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0), "s"(mask));)
        } else if constexpr (WHICH == 10) {  // v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0));)
        } else if constexpr (WHICH == 11) {  // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0), "v"(w1));)
        } else if constexpr (WHICH == 12) {  // v_and_b32
            REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                              "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0));)
        } else if constexpr (WHICH == 13) {  // v_dot2_f32_bf16 (VOP3P, separate destination)
            REP8(asm volatile("v_dot2_f32_bf16 %0, %8, %9, %0\n v_dot2_f32_bf16 %1, %8, %9, %1\n v_dot2_f32_bf16 %2, %8, %9, %2\n v_dot2_f32_bf16 %3, %8, %9, %3\n"
                              "v_dot2_f32_bf16 %4, %8, %9, %4\n v_dot2_f32_bf16 %5, %8, %9, %5\n v_dot2_f32_bf16 %6, %8, %9, %6\n v_dot2_f32_bf16 %7, %8, %9, %7\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0), "s"((unsigned)mask));)
        } else if constexpr (WHICH == 14) {  // v_cndmask_b32 VOP2 with vcc, sources distinct from the destination chain
            REP8(asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                              "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0), "v"(w1) : "vcc");)
        } else if constexpr (WHICH == 15) {  // v_lshlrev_b32
            REP8(asm volatile("v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n"
                              "v_lshlrev_b32 %4, 16, %4\n v_lshlrev_b32 %5, 16, %5\n v_lshlrev_b32 %6, 16, %6\n v_lshlrev_b32 %7, 16, %7\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if constexpr (WHICH == 16) {  // v_exp_f16
            REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n"
                              "v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if constexpr (WHICH == 17) {  // v_rcp_f16
            REP8(asm volatile("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3\n"
                              "v_rcp_f16 %4, %4\n v_rcp_f16 %5, %5\n v_rcp_f16 %6, %6\n v_rcp_f16 %7, %7\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if constexpr (WHICH == 18) {  // v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if constexpr (WHICH == 19) {  // v_permlane32_swap
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if constexpr (WHICH == 8) {   // v_mul_f32 with a DPP row permutation
            REP8(asm volatile("v_mul_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mul_f32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mul_f32_dpp %4, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mul_f32_dpp %6, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + w2 + w3 + w4 + w5 + w6 + w7;
}

template <int WHICH>
void rate(const char *name, int waves_per_simd, float *out, double ghz) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<WHICH>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, 10, 1.0f, 0.5f, 0x5555aaaa3333ccccull);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<WHICH>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, iters, 1.0f, 0.5f, 0x5555aaaa3333ccccull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * 64 * waves_per_simd);   // ns per instruction per SIMD
    printf("%-22s %d wave(s) per SIMD: %8.3f ms  %.3f ns per instruction per SIMD = %.2f cycles at %.2f GHz\n", name, waves_per_simd,
           ms, per, per * ghz, ghz);
}

// exactness: lo(a) = a - float(bf16(a)) by subtraction vs by v_dot2c_f32_bf16 with a (-1, 0) / (0, -1) selector
__global__ void split_check_kernel(const float *x, float *lo_sub, float *lo_dot, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const f32x2 v = {a, b};
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    lo_sub[2 * i] = a - __uint_as_float(hi << 16);
    lo_sub[2 * i + 1] = b - __uint_as_float(hi & 0xffff0000u);
    // (the builtin is not used: the compiler folds the (-1, 0) selector into the inline constant -1.0, which the hardware reads as
    // the fp32 pattern 0xbf800000 = (0, -1): the first remainder came out as a - hi.y in the first version of this check)
    float l0, l1;
    const unsigned m0 = 0x0000bf80u, m1 = 0xbf800000u;
    asm volatile("v_dot2_f32_bf16 %0, %2, %3, %5\n\tv_dot2_f32_bf16 %1, %2, %4, %6\n\ts_nop 2"
                 : "=&v"(l0), "=&v"(l1) : "v"(hi), "s"(m0), "s"(m1), "v"(a), "v"(b));
    lo_dot[2 * i] = l0;
    lo_dot[2 * i + 1] = l1;
}

int main() {
    float *out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    for (int w = 1; w <= 2; ++w) {
        rate<0>("v_sub_f32", w, out, ghz);
        rate<1>("v_dot2c_f32_bf16", w, out, ghz);
        rate<2>("v_pk_add_f32", w, out, ghz);
        rate<3>("v_cvt_pk_bf16_f32", w, out, ghz);
        rate<4>("v_cndmask_b32", w, out, ghz);
        rate<5>("v_exp_f32", w, out, ghz);
        rate<6>("v_pk_mul_f32", w, out, ghz);
        rate<7>("v_perm_b32", w, out, ghz);
        rate<8>("v_mul_f32_dpp", w, out, ghz);
        rate<9>("v_cndmask_b32_e64 sgpr", w, out, ghz);
        rate<14>("v_cndmask_b32 vcc, free src", w, out, ghz);
        rate<10>("v_mul_f32", w, out, ghz);
        rate<11>("v_fma_f32", w, out, ghz);
        rate<12>("v_and_b32", w, out, ghz);
        rate<15>("v_lshlrev_b32", w, out, ghz);
        rate<13>("v_dot2_f32_bf16 (VOP3P)", w, out, ghz);
        rate<18>("v_rcp_f32", w, out, ghz);
        rate<16>("v_exp_f16", w, out, ghz);
        rate<17>("v_rcp_f16", w, out, ghz);
        rate<19>("v_permlane32_swap_b32", w, out, ghz);
    }
    // split check: random magnitudes over the whole exponent range, plus denormals, zeros, exact bf16 values, ties
    const int n = 1 << 22;
    float *h = (float *)malloc(n * sizeof(float)), *a = (float *)malloc(n * sizeof(float)), *b = (float *)malloc(n * sizeof(float));
    srand(7);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
        if ((u & 0x7f800000u) == 0x7f800000u) u &= 0xbfffffffu;   // no inf / nan
        if (i % 97 == 0) u &= 0x807fffffu;                          // denormals
        if (i % 101 == 0) u &= 0xffff0000u;                         // exact bf16
        if (i % 103 == 0) u = (u & 0xffff0000u) | 0x8000u;          // ties
        memcpy(&h[i], &u, 4);
    }
    float *dx, *da, *db; hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_check_kernel, dim3(n / 2 / 256), dim3(256), 0, 0, dx, da, db, n);
    hipMemcpy(a, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b, db, n * 4, hipMemcpyDeviceToHost);
    long diff = 0, diff_norm = 0; double worst = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&a[i], &b[i], 4) != 0 && !(a[i] == 0.f && b[i] == 0.f)) {
            ++diff;
            if (fabsf(h[i]) >= 1.2e-38f * 65536.f) {
                ++diff_norm;
                const double rel = fabs((double)a[i] - (double)b[i]) / fabs((double)h[i]);
                if (rel > worst) worst = rel;
                if (diff_norm <= 5) printf("  x %.9g (0x%08x): sub %.9g dot %.9g\n", h[i], *(unsigned *)&h[i], a[i], b[i]);
            }
        }
    }
    printf("split check: %d values, lo differs on %ld (of which %ld with |x| >= 2^-110: worst |diff| / |x| = %.3g)\n", n, diff, diff_norm, worst);
    return 0;
}
